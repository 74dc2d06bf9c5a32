"""Forward attention with pre-scaled queries: the QS kernel (running maximum as the S chain's C operand) against the plain kernel fed the same
queries (OSUF_ATTN_FWD_NOQSK=1), interleaved in one process.   python tools/time_fwd_qs.py"""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops
D = 64
for B, N, H in ((32, 4096, 16), (32, 2048, 16), (32, 1024, 16), (32, 8192, 16)):
    qkv = torch.randn(B, N, (H + 2) * D, device="cuda").to(torch.bfloat16)
    qkv[..., : H * D] = (qkv[..., : H * D].float() * (D ** -0.5 * ops.LOG2E)).to(torch.bfloat16)
    outs, ts = {}, {False: [], True: []}
    for rnd in range(3):
        for old in (False, True):
            if old: os.environ["OSUF_ATTN_FWD_NOQSK"] = "1"
            else: os.environ.pop("OSUF_ATTN_FWD_NOQSK", None)
            fn = lambda: ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5, qs=True)
            outs[old] = fn()
            for _ in range(2): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8): fn()
            e.record(); torch.cuda.synchronize()
            ts[old].append(s.elapsed_time(e) / 8)
    os.environ.pop("OSUF_ATTN_FWD_NOQSK", None)
    f = 4.0 * B * H * N * N * D
    eo = ((outs[False][0].float() - outs[True][0].float()).norm() / outs[True][0].float().norm()).item()
    el = (outs[False][1] - outs[True][1]).abs().max().item()
    print(f"B={B} N={N}: plain kernel {min(ts[True]):.3f} ms ({f / min(ts[True]) / 1e9:5.0f} TF/s)   QS kernel {min(ts[False]):.3f} ms ({f / min(ts[False]) / 1e9:5.0f} TF/s)   "
          f"o rel diff {eo:.2e}  lse max diff {el:.2e}", flush=True)
