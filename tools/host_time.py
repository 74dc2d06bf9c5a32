"""Host-side enqueue time of one training step (the CPU must stay ahead of the GPU): wall time of Trainer.step calls WITHOUT a device
synchronisation in between, against the GPU time of the same steps.    python tools/host_time.py"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from osufusion_amd.train import Trainer

model = bench.build_model("cuda", bench.DIM_H)
trainer = Trainer(model, lr=1e-4, clip_grad_norm=1.0, compute_dtype=torch.bfloat16)
x, a, c, noise, t = bench.synth_batch(0, "cuda", bench.BATCH, bench.LENGTH)
for _ in range(3):
    trainer.step(x, a, c, noise, t)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(6):
    t0 = time.perf_counter()
    trainer.step(x, a, c, noise, t)
    host.append(time.perf_counter() - t0)          # returns when everything is enqueued (unless the launch queue back-pressures)
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / 6
print(f"enqueue time per step (ms): {[round(1e3 * h, 1) for h in host]}   GPU-paced step: {1e3 * total:.1f} ms")
