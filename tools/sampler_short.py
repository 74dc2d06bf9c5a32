"""A SHORT run of BASELINE config 4's sampler for counter passes: one DDIM sample at B=16, L=8192, cond_scale 2, S=3 steps (bf16, eager,
reproducible reductions) after a 1-step warm-up -- a few thousand dispatches, so a rocprofv3 --pmc pass keeps one record per dispatch
without the round-3 abort (the profiler's worker thread ran off its buffer under the ~120 k dispatches of S=50).
    python3 tools/sampler_short.py [--steps 3]"""
import argparse, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from osufusion_amd.models.diffusion import OsuFusion

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--length", type=int, default=8192)
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--dim-h", type=int, default=256)
args = ap.parse_args()
torch.manual_seed(0)
model = OsuFusion(args.dim_h)
with torch.no_grad():
    model.unet.final_conv.weight.normal_(0.0, 0.02)
model = model.cuda().eval()
model.set_full_bf16()
g = torch.Generator().manual_seed(7)
a = (torch.randn(args.batch, 96, args.length, generator=g) * 3 - 10).cuda()
c = (torch.rand(args.batch, 5, generator=g) * 2 - 1).cuda()
x0 = torch.randn(args.batch, 6, args.length, generator=g).cuda()
model.sampling_timesteps = 1
model.sample(a, c, x0.clone(), cond_scale=2.0)
model.sampling_timesteps = args.steps
torch.cuda.synchronize(); t0 = time.perf_counter()
y = model.sample(a, c, x0.clone(), cond_scale=2.0)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"steps": args.steps, "seconds": round(dt, 3), "finite": bool(torch.isfinite(y).all().item())}))
