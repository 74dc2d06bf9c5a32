"""Sustained shader clock under full-chip MFMA / VALU load (osuf_clock_probe)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops
for mode, name in ((1, "MFMA 32x32x16 bf16"), (2, "MFMA 16x16x32 bf16 (16 per iteration)"), (0, "v_fma_f32")):
    for blocks in (256, 1024, 2048):
        out = torch.zeros(2 * blocks, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        iters = 20000 if mode else 40000
        for _ in range(2):
            ops.call("osuf_clock_probe", blocks, iters, mode, out.data_ptr(), st)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.call("osuf_clock_probe", blocks, iters, mode, out.data_ptr(), st); e.record(); torch.cuda.synchronize()
        o = out.view(-1, 2).double().cpu()
        ghz = (o[:, 0] / (o[:, 1] * 10.0)).mean().item()          # cycles per 10-ns tick
        ms = s.elapsed_time(e)
        if mode:
            tf = blocks * 4 * iters * 8 * 2 * 32 * 32 * 16 / ms / 1e9
            cyc_per_mfma = (o[:, 0] / (iters * 8)).mean().item()
            print(f"{name}: {blocks} blocks x 4 waves: {ms:.2f} ms, {tf:.0f} TFLOP/s, shader clock {ghz:.2f} GHz, {cyc_per_mfma:.1f} cycles per MFMA per wave")
        else:
            print(f"{name}: {blocks} blocks: {ms:.2f} ms, shader clock {ghz:.2f} GHz, {(o[:,0]/(iters*64)).mean().item():.2f} cycles per FMA per wave")
