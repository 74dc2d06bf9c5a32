"""Which aten ops issue the small device-to-device memcpys of a training step (torch profiler, with Python stacks)."""
import collections
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from osufusion_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev, 256)
trainer = Trainer(model, lr=1e-4, weight_decay=1e-2, clip_grad_norm=1.0, compute_dtype=torch.bfloat16)
x, a, c, noise, t = bench.synth_batch(0, dev, 8, 1024)
for _ in range(2):
    trainer.step(x, a, c, noise, t)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.step(x, a, c, noise, t)
    torch.cuda.synchronize()
ev = prof.events()
by = collections.Counter()
for e in ev:
    n = e.name.lower()
    if "memcpy" in n or "copybuffer" in n:
        by[("GPU", e.name)] += 1
print(by.most_common(10))
# CPU-side ops that are aten::copy_ / clone with their stacks
stacks = collections.Counter()
for e in ev:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy") and e.device_type.name == "CPU":
        st = [s for s in (e.stack or []) if "osufusion_amd" in s or "torch/autograd" in s][:3]
        stacks[(e.name, tuple(st))] += 1
for (name, st), n in stacks.most_common(25):
    print(n, name, " <- ".join(s.split("/")[-1] for s in st))
