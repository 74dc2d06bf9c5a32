"""Per-kernel timing of the MQA attention kernels at the UNet's shapes (B=32, H=16, D=64)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops
B, H, D = 32, 16, 64
dev = "cuda"
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
print(f"{'N':>6s} {'fwd ms':>8s} {'TF/s':>6s} | {'dq ms':>8s} {'TF/s':>6s} | {'dkv ms':>8s} {'TF/s':>6s} | {'fused ms':>8s} {'alg TF/s':>8s} (whole backward = 2 x fwd FLOPs)")
for N in ([int(a) for a in sys.argv[1:]] or [4096, 2048, 1024, 512]):
    qkv = torch.randn(B, N, (H + 2) * D, device=dev).to(torch.bfloat16)
    o, lse = ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5)
    do = torch.randn(B, N, H * D, device=dev).to(torch.bfloat16)
    dqkv = torch.empty(B, N, (H + 2) * D, dtype=torch.float32, device=dev)
    delta = torch.empty(B, H, N, dtype=torch.float32, device=dev)
    base, gbase, ld, W = qkv.data_ptr(), dqkv.data_ptr(), (H + 2) * D, (H + 2) * D
    kp, vp = base + 2 * H * D, base + 2 * (H + 1) * D
    st = torch.cuda.current_stream().cuda_stream
    ops.call("osuf_attn_delta", do.data_ptr(), H * D, o.data_ptr(), H * D, 1, delta.data_ptr(), B, H, N, D, st)
    need = ops._lib.load().osuf_mqa_bwd_dkv_workspace_bytes(B, N, 0)
    ws = torch.empty(max(need, 16) // 4, dtype=torch.float32, device=dev)
    wsp = ws.data_ptr() if need else None
    f = 4.0 * B * H * N * N * D
    t1 = timeit(lambda: ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5))
    t2 = timeit(lambda: ops.call("osuf_mqa_bwd_dq", base, ld, kp, ld, vp, ld, do.data_ptr(), H * D, lse.data_ptr(), delta.data_ptr(), gbase, W, B, H, N, D, D ** -0.5, 0, None, None, 0, st))
    t3 = timeit(lambda: ops.call("osuf_mqa_bwd_dkv", base, ld, kp, ld, vp, ld, do.data_ptr(), H * D, lse.data_ptr(), delta.data_ptr(), gbase + 4 * H * D, gbase + 4 * (H + 1) * D, W, B, H, N, D, D ** -0.5, 0, None, None, wsp, need, 0, 0, st))
    cos = sin = None
    t4 = timeit(lambda: ops.mqa_bwd(qkv, o, do, lse, B, N, H, D, D ** -0.5, torch.bfloat16, cos, sin, variant=ops.ATTN_FUSED))
    t5 = timeit(lambda: ops.mqa_bwd(qkv, o, do, lse, B, N, H, D, D ** -0.5, torch.bfloat16, cos, sin, variant=ops.ATTN_AUTO))
    t6 = timeit(lambda: ops.mqa_bwd(qkv, o, do, lse, B, N, H, D, D ** -0.5, torch.bfloat16, cos, sin, variant=ops.ATTN_FUSED_SLABS))
    print(f"{N:6d} {t1:8.3f} {f / t1 / 1e9:6.0f} | {t2:8.3f} {1.5 * f / t2 / 1e9:6.0f} | {t3:8.3f} {2 * f / t3 / 1e9:6.0f} | {t4:8.3f} {2 * f / t4 / 1e9:8.0f}"
          f"  [ops.mqa_bwd incl. delta + finish: fused(atomics) {t4:.3f} ms, fused(slabs) {t6:.3f} ms, pair {t5:.3f} ms]", flush=True)
