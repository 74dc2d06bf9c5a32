import sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
M = 32
for N, K in ((512, 2048), (2048, 2048), (1024, 1024), (512, 1024), (256, 128)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    t1 = timeit(lambda: ops.skinny_fwd(x, w, b, torch.bfloat16, 1, 0))
    t2 = timeit(lambda: ops.skinny_bwd(dy, None, x, w, torch.bfloat16, 1, 0, True, None, None, False))
    t3 = timeit(lambda: ops.skinny_bwd(dy, None, x, w, torch.bfloat16, 1, 0, False, dw, db, True))
    print(f"N={N:5d} K={K:5d}: fwd {t1*1e3:6.1f} us  dx {t2*1e3:6.1f} us  dw {t3*1e3:6.1f} us   (W = {N*K*4/1e6:.1f} MB)")
