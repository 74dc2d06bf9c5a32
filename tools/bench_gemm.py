"""Per-shape timing of the tap-GEMM kernels on the UNet's real shapes (B=32, L=4096)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops

dev = "cuda"
B = 32
dt = torch.bfloat16 if "--f32" not in sys.argv else torch.float32

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

shapes = [  # (name, L, Cin, Cout, taps)
    ("conv3 256->256 L4096", 4096, 256, 256, 3), ("conv3 256->256 L2048", 2048, 256, 256, 3), ("conv3 512->512 L1024", 1024, 512, 512, 3),
    ("conv3 768->768 L512", 512, 768, 768, 3), ("conv3 1024->1024 L512", 512, 1024, 1024, 3), ("conv3 2048->1024 L512", 512, 2048, 1024, 3),
    ("conv3 512->256 L4096", 4096, 512, 256, 3),
    ("qkv 256->1152 L4096", 4096, 256, 1152, 1), ("to_out 1024->256 L4096", 4096, 1024, 256, 1), ("ff1 256->512 L4096", 4096, 256, 512, 1),
    ("ff2 512->256 L4096", 4096, 512, 256, 1), ("qkv 1024->1152 L512", 512, 1024, 1152, 1), ("ff1 1024->2048 L512", 512, 1024, 2048, 1),
]
print(f"{'shape':28s} {'nt ms':>8s} {'TF/s':>7s} | {'nt+stats':>8s} | {'tn ms':>8s} {'TF/s':>7s}")
for name, L, ci, co, taps in shapes:
    x = torch.randn(B, L, ci, device=dev).to(dt)
    w = (torch.randn(taps, co, ci, device=dev) * 0.05).to(dt)
    bias = torch.randn(co, device=dev)
    dy = torch.randn(B, L, co, device=dev).to(dt)
    out = torch.empty(B, L, co, device=dev, dtype=dt)
    stats = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    fl = 2.0 * B * L * ci * co * taps
    t1 = timeit(lambda: ops.gemm_nt(x, w, bias, taps=taps, lin=L, lout=L, pad=taps // 2, out=out))
    t2 = timeit(lambda: ops.gemm_nt(x, w, bias, taps=taps, lin=L, lout=L, pad=taps // 2, out=out, stats=stats))
    gw = torch.zeros(taps, co, ci, device=dev)
    t3 = timeit(lambda: ops.gemm_tn(dy, x, taps=taps, lin=L, lout=L, pad=taps // 2, out=gw))
    print(f"{name:28s} {t1:8.3f} {fl / t1 / 1e9:7.0f} | {t2:8.3f} | {t3:8.3f} {fl / t3 / 1e9:7.0f}", flush=True)
