"""GlobalContext pooling: the one-pass kernel pair (osuf_gca_pool) against rowdot + softmax_rows + wcolsum at the UNet's level shapes (B = 32), for a few
rows-per-workgroup settings (OSUF_GCA_RPB).   python tools/bench_gca.py"""
import os, sys
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
    return s.elapsed_time(e) / iters * 1e3
B = 32
for L, C in ((4096, 256), (2048, 512), (1024, 768), (512, 1024), (8192, 256)):
    h = torch.randn(B * L, C, device="cuda").bfloat16(); wk = torch.randn(C, device="cuda") * 0.1; bk = torch.randn(1, device="cuda")
    def old():
        p = ops.rowdot(h, wk, bk, L); ops.softmax_rows_(p, B, L); return ops.wcolsum(h, None, p, B, L)
    def old_repro():
        with ops.reproducible_mode(True):
            return old()
    line = f"B*L={B * L:6d} C={C:4d}: three kernels {timeit(old):6.1f} us (reproducible form {timeit(old_repro):6.1f})  one pass:"
    for rpb in (32, 64, 128, 256, 512):
        os.environ["OSUF_GCA_RPB"] = str(rpb)
        line += f"  [{rpb}] {timeit(lambda: ops.gca_pool(h, wk, bk, L)):6.1f}"
    os.environ.pop("OSUF_GCA_RPB", None)
    print(line, flush=True)
