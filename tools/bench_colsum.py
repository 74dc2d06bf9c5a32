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
for M, N in ((131072, 256), (131072, 512), (32768, 1024), (16384, 2048)):
    y = torch.randn(M, N, device="cuda").bfloat16(); out = torch.zeros(N, device="cuda")
    t = timeit(lambda: ops.colsum(y, N, out=out))
    print(f"colsum {M}x{N}: {t*1e3:7.1f} us  {M*N*2/t/1e6:7.0f} GB/s")
