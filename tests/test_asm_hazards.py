"""hipcc pads no hazards around inline-asm MFMAs (cdna_hip_programming.md 5.7): the emitted code of the asm-MFMA kernel is linted instead.
CPU-only (hipcc cross-compiles gfx950 without a GPU): compiles csrc/attn.hip with -save-temps and runs tools/check_mfma_hazards.py on
both instantiations of mqa_bwd_fused512_kernel (the generated loop of mqa_bwd_fused512a_kernel has its own checks: tests/test_attn_bwd512_generator.py)
-- no VALU write of an MFMA source within two instructions, no compiler-generated
v_accvgpr_* and no scratch access inside the main loop (a reload there would wait for vmcnt(0), i.e. for the float atomics in flight)."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.timeout(600)
@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_asm_mfma_kernels_have_no_unpadded_hazards(tmp_path):
    src = ROOT / "osufusion_amd" / "csrc"
    for f in ("attn.hip", "attn_generic.hpp", "common.hpp", "attn_bwd512_asm.inc", "attn_bwd512qs_asm.inc"):
        shutil.copy(src / f, tmp_path / f)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-save-temps", "-c", "attn.hip",
                        "-o", "attn.o"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = next(tmp_path.glob("attn-hip-amdgcn-amd-amdhsa-gfx950.s"))
    for inst in ("mqa_bwd_fused512_kernelILb1", "mqa_bwd_fused512_kernelILb0"):
        lint = subprocess.run([sys.executable, str(ROOT / "tools" / "check_mfma_hazards.py"), inst, str(asm)], capture_output=True, text=True)
        assert lint.returncode == 0, lint.stdout[-3000:]
        assert "128 MFMAs checked, 0 finding(s)" in lint.stdout, lint.stdout[-500:]
    # round 5: the loop kernel hands c and -1/c to its asm statement through inline-asm v_readfirstlane statements.  A vector write of the source
    # register directly in front of such a statement is a hazard the compiler pads for its own instructions only: with the kernel argument left
    # in an SGPR hipcc put `v_mov_b32 v, s` right there and the loop ran with a wrong c (bisected on the GPU: s_nop pads inside the statement
    # fix it, so does pinning the value into a VGPR earlier; csrc/attn.hip does both).  Every such statement must carry its pads.
    lines = [l.strip() for l in asm.read_text().split("\n")]
    for sym in ("_Z24mqa_bwd_fused512a_kernelILb0EEv8AttnArgsPf", "_Z24mqa_bwd_fused512a_kernelILb1EEv8AttnArgsPf"):
        start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        body = lines[start:end]
        n_rfl = 0
        for i, l in enumerate(body):
            if l.startswith("v_readfirstlane_b32") and any(x.startswith(";;#ASMSTART") for x in body[max(0, i - 2):i]):
                n_rfl += 1
                assert body[i - 1].startswith("s_nop") and body[i + 1].startswith("s_nop"), (sym, body[i - 1], l, body[i + 1])
        assert n_rfl >= 2, (sym, n_rfl)
    # the forward kernel keeps its four waves per SIMD (<= 128 VGPRs) WITHOUT scratch since round 4 (one unmasked copy of the tile body);
    # the generated backward loop runs at one wave per SIMD on the whole register file, also without scratch traffic in its loop
    import re
    text = asm.read_text()
    found = re.findall(r"^(_Z14mqa_fwd_kernelILi8E\w+):.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+)", text, re.S | re.M)
    assert len(found) >= 6, [f[0] for f in found]            # QS x WHOLE, + the two ROPE forms
    for sym, nv, sc in found:
        m = re.match(r"(\d+) (\d+)", f"{nv} {sc}")
        max_scratch = 0
        assert int(m.group(1)) <= 128 and int(m.group(2)) <= max_scratch, (sym, m.group(1), m.group(2))
