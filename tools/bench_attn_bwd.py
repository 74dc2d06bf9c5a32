"""Fused attention backward, per-variant timing at the UNet's shapes (B=32, H=16, D=64): the 256-key sweep, the 512-key sweep and
the 512-key sweep without its atomics (timing only).   python tools/bench_attn_bwd.py [N ...]"""
import os
import sys
os.environ["OSUF_ALLOW_TIMING_BUILDS"] = "1"        # the no-atomics timing build is refused without it
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops
B, H, D = 32, 16, 64
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for N in ([int(a) for a in sys.argv[1:]] or [4096, 2048, 1024, 512]):
    qkv = torch.randn(B, N, (H + 2) * D, device="cuda").to(torch.bfloat16)
    o, lse = ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5)
    do = torch.randn(B, N, H * D, device="cuda").to(torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    delta = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
    ops.call("osuf_attn_delta", do.data_ptr(), H * D, o.data_ptr(), H * D, 1, delta.data_ptr(), B, H, N, D, st)
    f = 8.0 * B * H * N * N * D
    row = []
    for name, var in (("fused256", ops.ATTN_FUSED256), ("fused512", ops.ATTN_FUSED512), ("fused512a", ops.ATTN_FUSED512A), ("fused512-noatomics", ops.ATTN_FUSED512_TIMING)):
        t = timeit(lambda: ops.mqa_bwd(qkv, o, do, lse, B, N, H, D, D ** -0.5, torch.bfloat16, None, None, variant=var, delta=delta))
        row.append(f"{name} {t:7.3f} ms ({f / t / 1e9:5.0f} alg TF/s)")
    print(f"N={N:5d}  " + " | ".join(row), flush=True)
