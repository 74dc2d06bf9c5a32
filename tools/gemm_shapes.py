"""Histogram of the tap-GEMM shapes of one training step at the headline size, with their 256x256 tile counts and the
share of the last round of 256 CUs that is idle.  usage: python tools/gemm_shapes.py"""
import collections
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from osufusion_amd import ops  # noqa: E402
from osufusion_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev, 256)
trainer = Trainer(model, lr=1e-4, weight_decay=1e-2, clip_grad_norm=1.0, compute_dtype=torch.bfloat16)
x, a, c, noise, t = bench.synth_batch(0, dev, 32, 4096)
trainer.step(x, a, c, noise, t)
torch.cuda.synchronize()
seen = collections.Counter()
orig = ops.call


def spy(name, *args, **kw):
    if name == "osuf_gemm_nt":
        M, N, K, taps = args[17], args[18], args[19], args[20]
        seen[(M, N, K, taps)] += 1
    return orig(name, *args, **kw)


ops.call = spy
trainer.step(x, a, c, noise, t)
torch.cuda.synchronize()
ops.call = orig
rows = []
for (M, N, K, taps), n in seen.items():
    tiles = -(-M // 256) * -(-N // 256)
    rounds = tiles / 256
    waste = 1 - rounds / -(-tiles // 256) if tiles >= 1 else 0
    gf = 2.0 * M * N * K * taps * n / 1e9
    rows.append((gf * waste / max(1e-9, 1 - waste), gf, M, N, K, taps, n, tiles, waste))
rows.sort(reverse=True)
print(f"{'M':>7s} {'N':>5s} {'K':>5s} taps  calls  tiles  idle-in-last-round  GFLOP/step")
for _, gf, M, N, K, taps, n, tiles, waste in rows[:40]:
    print(f"{M:7d} {N:5d} {K:5d} {taps:4d} {n:6d} {tiles:6d} {100 * waste:17.0f}% {gf:11.0f}")
tot = sum(r[1] for r in rows)
lost = sum(r[1] / (1 - r[8]) - r[1] for r in rows)
print(f"total {tot / 1e3:.1f} TFLOP/step in osuf_gemm_nt; tile quantisation stretches it by {100 * lost / tot:.1f} %")
