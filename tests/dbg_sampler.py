"""Debug helper (not collected by pytest): sampler vs oracle on the tiny golden config; lives under tests/ because it imports oracle/."""
import json, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import osufusion_amd as oa
from osufusion_amd import ops
from osufusion_amd.models.diffusion import OsuFusion
from osufusion_amd.pattern import param_pattern, synth_inputs
from oracle import diffusion_oracle as DO, unet_oracle as O
meta = json.load(open("tests/golden/unet_cases.json"))["unet_tiny"]
cfgd = {k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["cfg"].items()}
kw = {k: v for k, v in cfgd.items() if not k.startswith("dim_in_")}
model = OsuFusion(kw.pop("dim_h"), **kw).cuda()
sd = {k: torch.from_numpy(param_pattern(k, tuple(v.shape))).cuda() for k, v in model.unet.state_dict().items()}
model.unet.load_state_dict(sd)
cfg = O.UNetConfig(**cfgd)
p = O.make_params(cfg)
x, a, c, t, noise = (torch.from_numpy(v) for v in synth_inputs("sampler", 2, 256))
acp = DO.ddim_alphas_cumprod()
model.scheduler.set_timesteps(5)
xs_ref, xs = noise.clone(), noise.clone().cuda()
def rl(a_, b_): return ((a_.double().cpu() - b_.double()).norm() / b_.double().norm()).item()
with oa.forced_compute_dtype(torch.float32), torch.no_grad():
    for tt in [800, 600, 400, 200, 0]:
        tb = torch.full((2,), tt, dtype=torch.int64)
        pr = O.unet_forward(p, cfg, xs_ref, a, tb, c)
        pg = model.unet(xs, a.cuda(), tb.cuda(), c.cuda())
        print(tt, "pred rel", rl(pg, pr), "x rel", rl(xs, xs_ref), "|pred|", pr.abs().max().item())
        xs_ref = DO.ddim_step(pr, tt, xs_ref, acp, 5)
        coef = torch.tensor([model.scheduler.step_coefficients(tt)] * 2, dtype=torch.float32).cuda()
        xs = ops.ddim_step(xs, pg.contiguous(), None, 1.0, coef)
    print("final", rl(xs, xs_ref))
    model.sampling_timesteps = 5
    got = model.sample(a.cuda(), c.cuda(), noise.cuda(), cond_scale=1.0)
    print("sample() vs manual", rl(got, xs.cpu()), "vs ref", rl(got, xs_ref))
    print("---- determinism")
    tb = torch.full((2,), 800, dtype=torch.int64).cuda()
    xin = noise.cuda()
    outs = [model.unet(xin, a.cuda(), tb, c.cuda()).clone() for _ in range(3)]
    print("unet run-to-run max abs", (outs[0] - outs[1]).abs().max().item(), (outs[0] - outs[2]).abs().max().item(), "scale", outs[0].abs().max().item())
    s1 = model.sample(a.cuda(), c.cuda(), noise.cuda(), cond_scale=1.0)
    s2 = model.sample(a.cuda(), c.cuda(), noise.cuda(), cond_scale=1.0)
    print("sample run-to-run rel", rl(s1, s2.cpu()))
    # layer-by-layer determinism
    from osufusion_amd import runtime as rt
    dt = torch.float32
    xr = [model.unet.init_x.forward_rows(xin, dt) for _ in range(2)]
    print("stem", (xr[0] - xr[1]).abs().max().item())
    blk = model.unet.down_layers[0]
    te = model.unet.embed_time(tb); ce = model.unet.embed_cond(c.cuda(), torch.ones(2, dtype=torch.bool).cuda())
    r = [blk.init_resnet.forward_rows(xr[0], te, ce) for _ in range(2)]
    print("resblock", (r[0] - r[1]).abs().max().item())
    tr = [blk.transformers[0].forward_rows(r[0]) for _ in range(2)]
    print("transformer", (tr[0] - tr[1]).abs().max().item(), tr[0].abs().max().item())
    at = [blk.transformers[0].attn(r[0]) for _ in range(2)]
    print("attn", (at[0] - at[1]).abs().max().item())
