"""Forward attention kernels side by side on the same operands (one process, interleaved): osuf_mqa_fwd_qs on pre-rotated queries, osuf_mqa_fwd_rope
without / with the stored queries, and with the dQ zero fill.   python tools/time_fwd_rope.py"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import functional as Fn
from osufusion_amd import ops
D = 64
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for B, N, H in ((32, 4096, 16), (32, 2048, 16), (32, 1024, 16), (32, 8192, 16)):
    scale = D ** -0.5
    raw = torch.randn(B, N, (H + 2) * D, device="cuda").to(torch.bfloat16)
    cos, sin = Fn.rope_tables(N, D, 2 * N, "cuda")
    ref = ops.rope_cast(raw, cos, sin, N, H + 1, H + 2, D, q_mul=scale * ops.LOG2E, n_q_heads=H)
    o = torch.empty(B, N, H * D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, N, device="cuda")
    qout = torch.empty_like(ref); ws = torch.empty(B * N * H * D, device="cuda")
    W = (H + 2) * D
    kp, vp = ref.data_ptr() + 2 * H * D, ref.data_ptr() + 2 * (H + 1) * D
    common = (kp, W, vp, W, o.data_ptr(), H * D, 2, lse.data_ptr(), B, H, N, D, scale)
    DT = ops._DT[torch.bfloat16]
    common = (kp, W, vp, W, o.data_ptr(), H * D, DT, lse.data_ptr(), B, H, N, D, scale)
    st = ops._stream
    fns = {
        "qs": lambda: ops.call("osuf_mqa_fwd_qs", ref.data_ptr(), W, *common, st()),
        "qs+zdq": lambda: ops.call("osuf_mqa_fwd_zdq", ref.data_ptr(), W, *common, 1, ws.data_ptr(), st()),
        "rope": lambda: ops.call("osuf_mqa_fwd_rope", raw.data_ptr(), W, *common, cos.data_ptr(), sin.data_ptr(), scale * ops.LOG2E, None, W, None, st()),
        "rope+q": lambda: ops.call("osuf_mqa_fwd_rope", raw.data_ptr(), W, *common, cos.data_ptr(), sin.data_ptr(), scale * ops.LOG2E, qout.data_ptr(), W, None, st()),
        "rope+q+zdq": lambda: ops.call("osuf_mqa_fwd_rope", raw.data_ptr(), W, *common, cos.data_ptr(), sin.data_ptr(), scale * ops.LOG2E, qout.data_ptr(), W, ws.data_ptr(), st()),
        "rope_cast(all)": lambda: ops.rope_cast(raw, cos, sin, N, H + 1, H + 2, D, q_mul=scale * ops.LOG2E, n_q_heads=H),
        "rope_cast(k|v)": lambda: ops.call("osuf_rope_cast", DT, raw.data_ptr() + 2 * H * D, W, qout.data_ptr() + 2 * H * D, W, cos.data_ptr(), sin.data_ptr(), B * N, N, 1, 2, D, st()),
    }
    ts = {k: [] for k in fns}
    for rnd in range(3):
        for k, f in fns.items():
            ts[k].append(timeit(f))
    print(f"B={B} N={N}: " + "  ".join(f"{k} {min(v) * 1e3:7.1f} us" for k, v in ts.items()), flush=True)
