"""Which completion reports does one backward deliver per parameter: the kernels' direct-accumulation path (param_ready), autograd's
post-accumulate-grad hook, or both -- and in which order?  (GradReducer relies on the answer; run on the GPU box.)"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd.models.diffusion import OsuFusion
from osufusion_amd.pattern import param_pattern, synth_inputs
from osufusion_amd.train import Trainer

model = OsuFusion(32, dim_h_mult=(1, 2), num_layer_blocks=(1, 1), num_middle_transformers=1, cross_embed_kernel_sizes=(3,),
                  attn_dim_head=64, attn_heads=2, attn_kv_heads=1, attn_context_len=256).cuda()
model.unet.load_state_dict({k: torch.from_numpy(param_pattern(k, tuple(v.shape))).cuda() for k, v in model.unet.state_dict().items()})
tr = Trainer(model, compute_dtype=torch.float32)
red = tr.reducer
log = []
orig_ready, orig_hook = red.param_ready, red._hook
red.param_ready = lambda p: log.append(("direct", red.names[id(p)]))
red._hook = lambda p: log.append(("hook", red.names[id(p)]))
from osufusion_amd import functional as Fn
Fn.enable_direct_grads(True, red.param_ready)
for p in tr.flat.params:                                   # re-register so that the patched hook is the one called
    p.register_post_accumulate_grad_hook(red._hook)
x, a, c, t, noise = (torch.from_numpy(v).cuda() for v in synth_inputs("hooks", 2, 256))
tr.flat.zero_grad()
model.loss_with(x, a, c, noise, t, cond_drop_prob=0.0).backward()
torch.cuda.synchronize()
names = [n for n, _ in model.named_parameters()]
direct = [n for k, n in log if k == "direct"]
hook = [n for k, n in log if k == "hook"]
print(f"{len(names)} parameters; direct reports {len(direct)} ({len(set(direct))} distinct); hook reports {len(hook)} ({len(set(hook))} distinct)")
print("direct only:", sorted(set(direct) - set(hook))[:8])
print("hook only  :", sorted(set(hook) - set(direct))[:8])
print("neither    :", sorted(set(names) - set(hook) - set(direct))[:8])
first = {}
for i, (k, n) in enumerate(log):
    first.setdefault((k, n), i)
both = [n for n in set(direct) & set(hook)]
print("both:", len(both), "; hook after direct for", sum(first[("hook", n)] > first[("direct", n)] for n in both))
lag = [first[("hook", n)] - first[("direct", n)] for n in both]
print("reports between a parameter's direct report and its hook: max", max(lag) if lag else None, "median", sorted(lag)[len(lag) // 2] if lag else None)
