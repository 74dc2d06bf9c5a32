"""256-tile GEMM epilogue variants: plain / +residual / +dact at the transformer shapes (B=32)."""
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
M = 131072
for name, N, K in (("to_out 1024->256", 256, 1024), ("ff2 512->256", 256, 512), ("ff-dpre 256->512", 512, 256), ("dx 512->256", 256, 512)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(1, N, K, device="cuda") * 0.05).bfloat16()
    r = torch.randn(M, N, device="cuda").bfloat16(); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.gemm_nt(x, w, None, out=out))
    t1 = timeit(lambda: ops.gemm_nt(x, w, None, residual=r, out=out))
    t2 = timeit(lambda: ops.gemm_nt(x, w, None, dact=r, out=out))
    print(f"{name:20s} plain {t0*1e3:7.1f} us | +residual {t1*1e3:7.1f} us | +dact {t2*1e3:7.1f} us", flush=True)
