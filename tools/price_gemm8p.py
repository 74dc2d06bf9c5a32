"""Timing-only triage of the 8-phase GEMM loop (OSUF_GEMM_DBG builds; results are garbage unless the row says 'valid'): per K-step and tile,
what the full loop, the loop without MFMAs / fragment reads / DMA, the DMA alone and the experiment variants cost.  K-sweep at M = 131072 so that
slope = cost per K-step, intercept = per tile.      python tools/price_gemm8p.py [N ...]"""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops

M = 131072
Ns = [int(v) for v in sys.argv[1:]] or [1024, 256]
VARIANTS = ((0, "full (valid)"), (1, "no MFMA"), (2, "no frag reads"), (3, "no DMA"), (4, "DMA only"), (5, "DMA only, B from one hot KiB"),
            (6, "DMA only, A from one hot KiB"))
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for N in Ns:
    rounds = -(-(M // 256) * -(-N // 256) // 256)
    res = {}
    for K in (512, 2048):
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(1, N, K, device="cuda") * 0.05).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for rep in range(2):
            for dbg, name in VARIANTS:
                if dbg: os.environ["OSUF_GEMM_DBG"] = str(dbg)
                else: os.environ.pop("OSUF_GEMM_DBG", None)
                t = timeit(lambda: ops.gemm_nt(x, w, None, out=out))
                res[(K, name)] = min(t, res.get((K, name), 1e9))
        os.environ.pop("OSUF_GEMM_DBG", None)
    for _, name in VARIANTS:
        a, b = res[(512, name)], res[(2048, name)]
        slope = (b - a) / rounds / 24                    # 24 more K-steps
        print(f"N={N:5d} {name:48s} K=512 {a:8.1f} us  K=2048 {b:8.1f} us   per K-step {slope:5.2f} us   per tile (K->0) {a / rounds - 8 * slope:5.2f} us", flush=True)
