"""The generated loop of the 512-key attention backward (csrc/attn_bwd512_asm.inc <- tools/gen_attn_bwd512.py), checked without a GPU:
the committed text is what the generator emits today (its own checks ran: the list-scheduled body computes, register by register and
store by store, what the hand-ordered body computes, and the loop text is a fixed point over the back edge); an independent pass over the
TEXT finds no MFMA hazard and counts the instructions per (head, 32-query block) pair (round-3 review: <= 600 besides the 96 MFMAs)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_committed_loop_is_the_generators_output(tmp_path):
    out = tmp_path / "loop.inc"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_bwd512.py"), "--stats", "--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_text() == (ROOT / "osufusion_amd" / "csrc" / "attn_bwd512_asm.inc").read_text()
    assert "non-MFMA instructions per pair" in r.stdout


def test_hand_ordered_body_still_generates(tmp_path):
    """--manual: the body the list-scheduled one is compared against (pinned on the GPU: dK / dV bit-identical to the compiled sweep)"""
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_bwd512.py"), "--manual", "--out", str(tmp_path / "m.inc")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_loop_text_has_no_mfma_hazard_and_stays_on_its_diet():
    sys.path.insert(0, str(ROOT / "tools"))
    import check_mfma_hazards as H
    assert H.check_inc(str(ROOT / "osufusion_amd" / "csrc" / "attn_bwd512_asm.inc"), max_non_mfma=600)


def test_a_broken_schedule_is_caught(tmp_path):
    """the dataflow check is not vacuous: without the rule that a producer is due no later than its consumers, dS = p dP' of a tile lands behind
    the next tile's S chain (the round's first list-scheduled body did exactly that) and the generator refuses"""
    src = (ROOT / "tools" / "gen_attn_bwd512.py").read_text()
    assert "for _ in range(6):" in src
    bad = tmp_path / "gen_bad.py"
    bad.write_text(src.replace("for _ in range(6):", "for _ in range(0):", 1).replace("ROOT = Path(__file__).resolve().parent.parent", f"ROOT = Path({str(ROOT)!r})"))
    r = subprocess.run([sys.executable, str(bad), "--out", str(tmp_path / "x.inc")], capture_output=True, text=True)
    assert r.returncode != 0 and "dataflow differs" in r.stderr
