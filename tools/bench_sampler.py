"""Secondary metric (BASELINE.json configs[3]): DDIM sampling on 1 MI355X -- batch 16, L=8192, 50 steps, classifier-free
guidance on (cond_scale 2.0), full UNet dim_h=256, bf16; eager vs hipGraph-captured step.
    python tools/bench_sampler.py [--batch 16 --length 8192 --steps 50]"""
import argparse, json, sys, time
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd.models.diffusion import OsuFusion

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--length", type=int, default=8192)
ap.add_argument("--steps", type=int, default=50); ap.add_argument("--dim-h", type=int, default=256)
args = ap.parse_args()
torch.manual_seed(0)
model = OsuFusion(args.dim_h)
with torch.no_grad():
    model.unet.final_conv.weight.normal_(0.0, 0.02)
model = model.cuda().eval()
model.set_full_bf16()
model.sampling_timesteps = args.steps
g = torch.Generator().manual_seed(7)
a = (torch.randn(args.batch, 96, args.length, generator=g) * 3 - 10).cuda()
c = (torch.rand(args.batch, 5, generator=g) * 2 - 1).cuda()
x0 = torch.randn(args.batch, 6, args.length, generator=g).cuda()
res = {}
outs = {}
for mode in ("eager", "graph", "eager-atomic"):                  # "eager-atomic": the training kernels' atomic reductions (not reproducible)
    model.use_hip_graph = mode == "graph"
    model.reproducible_sampling = mode != "eager-atomic"
    model.sampling_timesteps = 3
    model.sample(a, c, x0.clone(), cond_scale=2.0)              # warm-up
    model.sampling_timesteps = args.steps
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs[mode] = model.sample(a, c, x0.clone(), cond_scale=2.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[mode] = dict(seconds=round(dt, 3), sampling_steps_per_s=round(args.steps / dt, 3), samples_per_s=round(args.batch / dt, 3))
    print(f"[sampler] {mode}: {dt:.2f} s for {args.steps} steps", file=sys.stderr, flush=True)
model.use_hip_graph, model.reproducible_sampling = False, True
again = model.sample(a, c, x0.clone(), cond_scale=2.0)
noise_floor = ((outs["eager"] - again).norm() / outs["eager"].norm()).item()      # eager vs eager, fixed-order reductions: must be 0
diff = ((outs["eager"] - outs["graph"]).norm() / outs["eager"].norm()).item()
atomic_vs_repro = ((outs["eager"] - outs["eager-atomic"]).norm() / outs["eager"].norm()).item()   # chaos of an untrained net over 50 steps
# forward FLOPs: 4,256 GF/sample/eval at L=8192, audio encoder (1,317.5 GF) evaluated once instead of 2*S times
per_eval = 4256.0 * (args.length / 8192) if args.length == 8192 else None
print(json.dumps({"metric": "DDIM sampling (B=16, L=8192, S=50, CFG) on 1 MI355X", "config": vars(args), "dtype": "bf16", **res,
                  "graph_vs_eager_rel_l2": diff, "eager_vs_eager_rel_l2": noise_floor, "bit_identical": bool(torch.equal(outs["eager"], again) and torch.equal(outs["eager"], outs["graph"])),
                  "atomic_reductions_vs_reproducible_rel_l2": atomic_vs_repro}))
