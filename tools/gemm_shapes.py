"""Per-shape time of every GEMM launch of one real training step (B=32, L=4096, dim_h=256, bf16): which shapes the 256^2
kernels spend their time on, at what rate, with their 256x256 tile count and the idle share of the last round of 256 CUs.
HIP events around each C-ABI call on the launch stream.
    python tools/gemm_shapes.py [--top 40]"""
import argparse, collections, sys
sys.path.insert(0, "/root/repo")
import torch
import bench
from osufusion_amd import ops
from osufusion_amd.train import Trainer

ap = argparse.ArgumentParser(); ap.add_argument("--top", type=int, default=45); args = ap.parse_args()
model = bench.build_model("cuda", bench.DIM_H)
trainer = Trainer(model, lr=1e-4, clip_grad_norm=1.0, compute_dtype=torch.bfloat16)
x, a, c, noise, t = bench.synth_batch(0, "cuda", bench.BATCH, bench.LENGTH)
for _ in range(2):
    trainer.step(x, a, c, noise, t)
torch.cuda.synchronize()
rec = []
orig = ops.call
POS = {"osuf_gemm_nt": (17, 18, 19, 20, 25, 26), "osuf_gemm_tn": (8, 9, 10, 11, 16, 19), "osuf_gemm_nt_rowdot": (10, 11, 12, None, None, None)}
def timed(name, *a, meta=None):
    if name not in POS:
        return orig(name, *a, meta=meta)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(name, *a, meta=meta); e.record()
    p = POS[name]
    has = lambda i: a[i] is not None
    extra = ""
    if name == "osuf_gemm_nt":
        extra = ("+C2" if has(8) else "") + ("+R" if has(10) else "") + ("+U" if has(12) else "") + ("+st" if has(16) else "")
    rec.append((name[5:], tuple(a[i] if i is not None else 1 for i in p), extra, s, e))
ops.call = timed
trainer.step(x, a, c, noise, t)
torch.cuda.synchronize()
ops.call = orig
agg = collections.OrderedDict()
for name, (M, N, K, taps, mode, act), extra, s, e in rec:
    k = (name, M, N, K, taps, mode, act, extra)
    n, ms = agg.get(k, (0, 0.0))
    agg[k] = (n + 1, ms + s.elapsed_time(e))
tot = collections.Counter()
print(f"{'kernel':14s} {'M':>7s} {'N':>5s} {'K':>5s} t md act {'extra':8s} {'calls':>5s} {'ms':>8s} {'us/call':>8s} {'TF/s':>6s} {'tiles':>6s} {'idle':>5s}")
for (name, M, N, K, taps, mode, act, extra), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
    fl = 2.0 * M * N * K * taps * n
    tiles = -(-M // 256) * -(-N // 256)
    idle = 1 - (tiles / 256) / -(-tiles // 256)
    tl = f"{tiles:6d} {idle:5.2f}" if name != "gemm_tn" else ""
    print(f"{name:14s} {M:7d} {N:5d} {K:5d} {taps} {mode!s:>2s} {act!s:>3s} {extra:8s} {n:5d} {ms:8.3f} {1e3 * ms / n:8.1f} {fl / ms / 1e9:6.0f} {tl}")
for (name, *_), (n, ms) in agg.items():
    tot[name] += ms
print({k: round(v, 2) for k, v in tot.items()}, "launches", len(rec))
