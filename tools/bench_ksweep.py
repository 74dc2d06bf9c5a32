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
for N in (256, 1152):
    for K in (64, 128, 256, 512, 1024):
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(1, N, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm_nt(x, w, bias, out=out))
        t0 = timeit(lambda: ops.gemm_nt(x, w, None, out=out))
        print(f"N={N:5d} K={K:5d}  {t*1e3:8.1f} us (no bias {t0*1e3:8.1f})  {2.0*M*N*K/t/1e9:7.0f} TF/s   out bytes {M*N*2/1e6:.0f} MB -> {M*N*2/t/1e6:6.0f} GB/s")
