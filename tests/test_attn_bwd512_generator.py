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


def test_committed_prescaled_query_loop_is_the_generators_output(tmp_path):
    """--qs: the variant for queries pre-scaled by scale * log2 e (osuf_mqa_bwd_fused_qs): same generator, same checks, 64 fewer vector
    instructions per pair, its own committed text"""
    out = tmp_path / "loop_qs.inc"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_bwd512.py"), "--qs", "--stats", "--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_text() == (ROOT / "osufusion_amd" / "csrc" / "attn_bwd512qs_asm.inc").read_text()
    n = int(r.stdout.split("non-MFMA instructions per pair: ")[1].split(";")[0])
    assert n <= 440, n
    assert "v_mul_f32_e32 v128, s74" not in out.read_text()          # no score multiply left


def test_hand_ordered_body_still_generates(tmp_path):
    """--manual: the body the list-scheduled one is compared against (pinned on the GPU: dK / dV bit-identical to the compiled sweep)"""
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_bwd512.py"), "--manual", "--out", str(tmp_path / "m.inc")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_loop_text_has_no_mfma_hazard_and_stays_on_its_diet():
    sys.path.insert(0, str(ROOT / "tools"))
    import check_mfma_hazards as H
    assert H.check_inc(str(ROOT / "osufusion_amd" / "csrc" / "attn_bwd512_asm.inc"), max_non_mfma=600)
    assert H.check_inc(str(ROOT / "osufusion_amd" / "csrc" / "attn_bwd512qs_asm.inc"), max_non_mfma=600)


def test_a_broken_schedule_is_caught(tmp_path):
    """the dataflow check is not vacuous: without the rule that a producer is due no later than its consumers, dS = p dP' of a tile lands behind
    the next tile's S chain (the round's first list-scheduled body did exactly that) and the generator refuses"""
    src = (ROOT / "tools" / "gen_attn_bwd512.py").read_text()
    assert "for _ in range(6):" in src
    bad = tmp_path / "gen_bad.py"
    bad.write_text(src.replace("for _ in range(6):", "for _ in range(0):", 1).replace("ROOT = Path(__file__).resolve().parent.parent", f"ROOT = Path({str(ROOT)!r})"))
    r = subprocess.run([sys.executable, str(bad), "--out", str(tmp_path / "x.inc")], capture_output=True, text=True)
    assert r.returncode != 0 and "dataflow differs" in r.stderr


def test_every_global_address_of_the_loop_is_the_element_it_should_be():
    """tools/check_bwd512a_addresses.py: the emitted text's scalar state machine and 32-bit running offsets, executed for all lanes from the
    register values the kernel's prologue passes in, give exactly the wanted Q / dO / lse / delta / dQ element at every vector-memory
    instruction -- for every shape family the launcher admits (H == 1, minimum trip count, odd H, query splits, strides at the 2^31 guard) --
    and the statement clobbers everything it writes (m0).  Then three deliberately broken streams must be refused: a wrong dQ
    head step, a request wrap delta taken from the wrong register, and a block wrap of the requests one head late."""
    sys.path.insert(0, str(ROOT / "tools"))
    import check_bwd512a_addresses as C
    lines, clob = C.inc_lines(C.INC)
    errs, checked = C.check_text(lines, clob, quick=True, verbose=False)
    assert errs == 0 and checked > 100_000
    # the pre-scaled-query variant forms the same addresses: two short shape families + its clobber list
    qlines, qclob = C.inc_lines(C.INC.with_name("attn_bwd512qs_asm.inc"))
    assert not C.check_clobbers(qlines, qclob)
    for shp in ((2, 1024, 1, 2), (2, 1024, 16, 2), (1, 512, 2, 16)):
        sh = C.Shape(*shp, b=shp[0] - 1, part=shp[3] - 1)
        assert sum(len(C.run(qlines, sh, w)[0]) for w in range(4)) == 0

    def findings(mutate):
        bad = mutate(list(lines))
        assert bad != lines
        sh = C.Shape(2, 1024, 4, 2, b=1, part=1)
        return sum(len(C.run(bad, sh, w)[0]) for w in range(4))
    assert findings(lambda L: [l.replace("s_movk_i32 s90, 256", "s_movk_i32 s90, 252") for l in L]) > 0
    assert findings(lambda L: [l.replace("v_add_u32_e32 v252, s66, v252", "v_add_u32_e32 v252, s68, v252") for l in L]) > 0
    assert findings(lambda L: [l.replace("s_sub_u32 s62, s61, 2", "s_sub_u32 s62, s61, 1") for l in L]) > 0
    assert C.check_clobbers(lines, clob - {"m0"})                     # without the m0 clobber the check complains
