"""LayerNorm forward / backward at the UNet's shapes: time and HBM-side GB/s (algorithmic bytes); the backward at several grid caps
(OSUF_LN_BWD_BLOCKS, one dgamma / dbeta atomic per channel and block).  python tools/bench_ln.py"""
import os, sys
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
for M, C in ((131072, 256), (65536, 512), (32768, 768), (16384, 1024)):
    x = torch.randn(M, C, device="cuda").bfloat16(); dy = torch.randn(M, C, device="cuda").bfloat16()
    g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    out, mr = ops.ln_fwd(x, g, b)
    t1 = timeit(lambda: ops.ln_fwd(x, g, b))
    line = f"M={M:6d} C={C:4d}  fwd {t1*1e3:6.1f} us {M*C*4/t1/1e6:6.0f} GB/s | bwd"
    ref = None
    for cap in (None, 128, 256, 512, 1024, 2048):
        if cap is None: os.environ.pop("OSUF_LN_BWD_BLOCKS", None)
        else: os.environ["OSUF_LN_BWD_BLOCKS"] = str(cap)
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        dx = ops.ln_bwd(dy, x, mr, g, dg, db)
        dx = dx[0] if isinstance(dx, (tuple, list)) else dx
        if ref is None: ref = (dx.clone(), dg.clone(), db.clone())
        ok = torch.equal(dx, ref[0]) and (dg - ref[1]).abs().max().item() <= 1e-3 * ref[1].abs().max().item() and (db - ref[2]).abs().max().item() <= 1e-3 * ref[2].abs().max().item()
        t2 = timeit(lambda: ops.ln_bwd(dy, x, mr, g, dg, db))
        line += f"  [{cap or 'default'}] {t2*1e3:6.1f} us {M*C*6/t2/1e6:5.0f} GB/s{'' if ok else ' MISMATCH'}"
    os.environ.pop("OSUF_LN_BWD_BLOCKS", None)
    print(line, flush=True)
