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
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1024, 1024), (1024, 256)]      # N,K pairs
for N, K in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(1, N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm_nt(x, w, None, out=out))
    print(f"N={N} K={K}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:7.0f} TF/s-equivalent")
