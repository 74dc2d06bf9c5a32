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
for L, C in ((4096, 256), (1024, 512), (512, 1024)):
    y = torch.randn(B, L, C, device="cuda").bfloat16(); dh = torch.randn(B, L, C, device="cuda").bfloat16()
    mr = torch.stack([torch.zeros(B, device="cuda"), torch.ones(B, device="cuda")], 1).contiguous()
    g = torch.ones(C, device="cuda"); bt = torch.zeros(C, device="cuda"); ss = torch.randn(B, 2 * C, device="cuda") * 0.1
    t1 = timeit(lambda: ops.gn_apply(y, mr, g, bt, ss, L))
    t2 = timeit(lambda: ops.gn_bwd(dh, y, mr, g, bt, ss, L))
    by = B * L * C * 2
    print(f"L={L} C={C}: gn_apply {t1*1e3:6.1f} us ({2*by/t1/1e6:5.0f} GB/s)   gn_bwd (3 kernels) {t2*1e3:6.1f} us ({5*by/t2/1e6:5.0f} GB/s)")
