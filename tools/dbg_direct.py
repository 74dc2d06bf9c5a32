import json, sys
sys.path.insert(0, "/root/repo")
import torch
import osufusion_amd as oa
from osufusion_amd import functional as Fn
from osufusion_amd.models.diffusion import OsuFusion
from osufusion_amd.pattern import param_pattern, synth_inputs
from osufusion_amd.train import Trainer
meta = json.load(open("tests/golden/unet_cases.json"))["unet_tiny"]
cfgd = {k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["cfg"].items()}
kw = {k: v for k, v in cfgd.items() if not k.startswith("dim_in_")}
model = OsuFusion(kw.pop("dim_h"), **kw).cuda()
sd = {k: torch.from_numpy(param_pattern(k, tuple(v.shape))).cuda() for k, v in model.unet.state_dict().items()}
model.unet.load_state_dict(sd)
x, a, c, t, noise = (torch.from_numpy(v).cuda() for v in synth_inputs("unet_tiny", meta["B"], meta["L"]))
with oa.forced_compute_dtype(torch.bfloat16):
    model.loss_with(x, a, c, noise, t, cond_drop_prob=0.0).backward()
ref = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
trainer = Trainer(model, compute_dtype=torch.bfloat16)
trainer.flat.zero_grad()
with oa.forced_compute_dtype(torch.bfloat16):
    model.loss_with(x, a, c, noise, t, cond_drop_prob=0.0).backward()
torch.cuda.synchronize()
bad = []
for k, p in model.named_parameters():
    r = ref[k]
    e = ((p.grad - r).abs().max() / (r.abs().max() + 1e-20)).item()
    if e > 2e-3: bad.append((e, k, tuple(p.shape), (p.grad.norm() / (r.norm() + 1e-20)).item()))
for e, k, sh, ratio in sorted(bad, reverse=True)[:25]: print(f"{e:10.3e} ratio {ratio:8.3f} {k} {sh}")
print(len(bad), "bad of", len(ref))
