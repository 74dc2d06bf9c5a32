"""GroupNorm(1, C) + FiLM + SiLU forward apply and backward (reduce, apply with the per-sample finalize folded in), gate_residual, wcolsum at the UNet's four level shapes
(B = 32): time per call; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.   python tools/bench_gn.py"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B = 32
for L, C in ((4096, 256), (2048, 512), (1024, 768), (512, 1024)):
    M = B * L
    y = torch.randn(M, C, device="cuda").bfloat16(); dh = torch.randn(M, C, device="cuda").bfloat16()
    g = torch.randn(C, device="cuda"); bt = torch.randn(C, device="cuda"); ss = torch.randn(B, 2 * C, device="cuda") * 0.1
    mr = ops.gn_stats(y, L)
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); dbias = torch.zeros(C, device="cuda")
    gate = torch.rand(B, C, device="cuda")
    t_f = timeit(lambda: ops.gn_apply(y, mr, g, bt, ss, L))
    t_b = timeit(lambda: ops.gn_bwd(dh, y, mr, g, bt, ss, L, dg, db, dbias))
    t_g = timeit(lambda: ops.gate_residual(y, gate, dh, L))
    by = M * C * 2
    print(f"B*L={M:6d} C={C:4d}  gn_apply {t_f*1e3:6.1f} us {2*by/t_f/1e6:5.0f} GB/s | gn_bwd (reduce + apply) {t_b*1e3:6.1f} us {5*by/t_b/1e6:5.0f} GB/s | "
          f"gate_residual {t_g*1e3:6.1f} us {3*by/t_g/1e6:5.0f} GB/s", flush=True)
