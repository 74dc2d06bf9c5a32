"""Forward attention A/B on one per-call environment switch of the library, interleaved in one process (default: OSUF_ATTN_FWD_NOWHOLE, the
kernel without / with the bounds checks of its K / V loads).   python tools/time_fwd_env.py [ENV_VAR]"""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops
VAR = sys.argv[1] if len(sys.argv) > 1 else "OSUF_ATTN_FWD_NOWHOLE"
D = 64
for B, N, H in ((32, 4096, 16), (32, 2048, 16), (32, 1024, 16), (32, 512, 16), (32, 8192, 16)):
    qkv = torch.randn(B, N, (H + 2) * D, device="cuda").to(torch.bfloat16)
    qkv[..., : H * D] = (qkv[..., : H * D].float() * (D ** -0.5 * ops.LOG2E)).to(torch.bfloat16)
    outs, ts = {}, {False: [], True: []}
    for rnd in range(3):
        for old in (False, True):
            if old: os.environ[VAR] = "1"
            else: os.environ.pop(VAR, None)
            fn = lambda: ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5, qs=True)
            outs[old] = fn()
            for _ in range(2): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8): fn()
            e.record(); torch.cuda.synchronize()
            ts[old].append(s.elapsed_time(e) / 8)
    os.environ.pop(VAR, None)
    f = 4.0 * B * H * N * N * D
    same = torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1])
    print(f"B={B} N={N}: {VAR}=1 {min(ts[True]):.3f} ms ({f / min(ts[True]) / 1e9:5.0f} TF/s)   default {min(ts[False]):.3f} ms ({f / min(ts[False]) / 1e9:5.0f} TF/s)   "
          f"bit-identical {same}", flush=True)
