"""Bit-reproducibility of the sampler at config 4's size, many times over: 3-step DDIM samples (B=16, L=8192, cond_scale 2, bf16) of the
full-size model, each compared bit for bit with the first; optionally after an fp32-mode sample (the order of tests/test_full_size.py).
    python tools/check_sampler_repro.py [rounds]"""
import sys
sys.path.insert(0, "/root/repo")
import torch
import osufusion_amd as oa
from osufusion_amd.models.diffusion import OsuFusion

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
model = OsuFusion(256)
with torch.no_grad():
    model.unet.final_conv.weight.normal_(0.0, 0.02)
model = model.cuda().eval()
g = torch.Generator().manual_seed(404)
a = (torch.randn(16, 96, 8192, generator=g) * 3 - 10).cuda()
c = (torch.rand(16, 5, generator=g) * 2 - 1).cuda()
x0 = torch.randn(16, 6, 8192, generator=g).cuda()
model.sampling_timesteps = 50
model.stop_after = 1
with oa.forced_compute_dtype(torch.float32):
    model.sample(a, c, x0, cond_scale=2.0)
model.stop_after = 3
bad = 0
with oa.forced_compute_dtype(torch.bfloat16):
    ref = model.sample(a, c, x0, cond_scale=2.0)
    for i in range(rounds):
        y = model.sample(a, c, x0, cond_scale=2.0)
        if not torch.equal(y, ref):
            bad += 1
            print(f"round {i}: differs, rel-L2 {((y - ref).norm() / ref.norm()).item():.3e}", flush=True)
print(f"{rounds} samples, {bad} differ from the first")
sys.exit(1 if bad else 0)
