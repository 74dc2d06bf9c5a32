"""Shapes and times of the embedding-sized ("skinny", M = batch) linear launches of one real training step."""
import collections, sys
sys.path.insert(0, "/root/repo")
import torch
import bench
from osufusion_amd import ops
from osufusion_amd.train import Trainer

model = bench.build_model("cuda", bench.DIM_H)
trainer = Trainer(model, lr=1e-4, clip_grad_norm=1.0, compute_dtype=torch.bfloat16)
x, a, c, noise, t = bench.synth_batch(0, "cuda", bench.BATCH, bench.LENGTH)
for _ in range(2):
    trainer.step(x, a, c, noise, t)
torch.cuda.synchronize()
rec = []
orig = ops.call
POS = {"osuf_skinny_fwd": (7, 8, 9, 10, 11), "osuf_skinny_bwd": (12, 13, 14, 15, 16)}
def timed(name, *a, meta=None):
    if name not in POS:
        return orig(name, *a, meta=meta)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(name, *a, meta=meta); e.record()
    extra = ""
    if name == "osuf_skinny_bwd":
        extra = ("dx " if a[8] is not None else "") + ("dW " if a[10] is not None else "") + ("db" if a[11] is not None else "")
    rec.append((name[5:], tuple(a[i] for i in POS[name]), extra, s, e))
ops.call = timed
trainer.step(x, a, c, noise, t)
torch.cuda.synchronize()
ops.call = orig
agg = collections.OrderedDict()
for name, shp, extra, s, e in rec:
    k = (name, shp, extra)
    n, ms = agg.get(k, (0, 0.0))
    agg[k] = (n + 1, ms + s.elapsed_time(e))
print(f"{'kernel':12s} {'(M, N, K, in_act, out_act)':30s} {'outputs':10s} {'calls':>5s} {'ms':>7s} {'us/call':>8s} {'GB/s (W)':>9s}")
for (name, shp, extra), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K = shp[:3]
    print(f"{name:12s} {str(shp):30s} {extra:10s} {n:5d} {ms:7.3f} {1e3 * ms / n:8.1f} {4.0 * N * K * n / ms / 1e6:9.0f}")
print("total ms", round(sum(ms for _, ms in agg.values()), 2), "launch sites", len(rec))
