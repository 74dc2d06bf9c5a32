#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container, next to the reference:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python3 tests/golden/make_golden.py

It imports the reference's own modules (osu_fusion.modules.{unet,residual,attention}) from
/root/reference -- nothing is copied -- fills weights/inputs with the closed-form integer-hash
pattern of osufusion_amd/pattern.py, runs them on CPU in fp32 and stores inputs-free ``.npz``
fixtures (expected outputs / losses / gradient norms / gradient slices) under tests/golden/.

The one harness shim: on a CPU-only host ``Attend`` never sets
``cuda_config`` (attention.py:68-69) but reads it unconditionally (attention.py:87).  We set
``mod.cuda_config = _config(True, False, False)`` from outside on every Attend instance, which
is what the reference itself selects on any sm>=80 / gfx9 GPU (attention.py:71-73): bf16 SDPA.
"""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from osufusion_amd.pattern import param_pattern, synth_inputs, uniform_pm  # noqa: E402

from osu_fusion.modules import attention as ref_attention  # noqa: E402  (reference, PYTHONPATH)
from osu_fusion.modules import residual as ref_residual  # noqa: E402
from osu_fusion.modules import unet as ref_unet  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def shim_attend(module: torch.nn.Module) -> torch.nn.Module:
    for m in module.modules():
        if isinstance(m, ref_attention.Attend):
            m.cuda_config = ref_attention._config(True, False, False)
    return module


def load_pattern(module: torch.nn.Module, strip: str = "") -> None:
    sd = module.state_dict()
    new = {k: torch.from_numpy(param_pattern(k, tuple(v.shape)).copy()) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)


def T(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name: str, **arrays) -> None:
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(HERE / f"{name}.npz", **out)
    sz = (HERE / f"{name}.npz").stat().st_size
    print(f"  wrote {name}.npz ({sz / 1024:.1f} KiB)")


# --------------------------------------------------------------------------------------
# module-level cases
# --------------------------------------------------------------------------------------

def module_cases() -> None:
    B = 2
    # Block / ResidualBlock / GlobalContext
    x = T(uniform_pm("mod/x48", (B, 48, 96), 1.0))
    t = T(uniform_pm("mod/t", (B, 64), 1.0))
    c = T(uniform_pm("mod/c", (B, 64), 1.0))

    m = ref_residual.Block(48, 80)
    load_pattern(m)
    ss = (T(uniform_pm("mod/scale", (B, 80, 1), 0.5)), T(uniform_pm("mod/shift", (B, 80, 1), 0.5)))
    save("mod_block", y_plain=m(x), y_film=m(x, scale_shift=ss))

    m = ref_residual.GlobalContext(48, 48)
    load_pattern(m)
    save("mod_global_context", y=m(x))

    m = ref_residual.ResidualBlock(48, 80, 64, 64)
    load_pattern(m)
    save("mod_resblock_film", y=m(x, t, c))

    m = ref_residual.ResidualBlock(48, 48, None, None)
    load_pattern(m)
    save("mod_resblock_plain", y=m(x))

    # samplers / stems
    m = ref_unet.Downsample(48, 80)
    load_pattern(m)
    save("mod_downsample", y=m(x))
    m = ref_unet.Upsample(48, 80)
    load_pattern(m)
    save("mod_upsample", y=m(x))
    m = ref_unet.Parallel(torch.nn.Conv1d(48, 80, 3, padding=1), torch.nn.Conv1d(48, 80, 1))
    load_pattern(m)
    save("mod_parallel", y=m(x))
    xa = T(uniform_pm("mod/xa", (B, 96, 64), 1.0))
    m = ref_unet.CrossEmbedLayer(96, 128, (3, 7, 15))
    load_pattern(m)
    save("mod_cross_embed", y=m(xa))
    x6 = T(uniform_pm("mod/x6", (B, 6, 64), 1.0))
    m = ref_unet.CrossEmbedLayer(6, 128, (3, 7, 15))
    load_pattern(m)
    save("mod_cross_embed6", y=m(x6))

    # embeddings
    m = ref_unet.SinusoidalPositionEmbedding(128)
    save("mod_sinusoidal", y=m(torch.tensor([0, 1, 17, 500, 999], dtype=torch.int64)))
    for n, sb in ((512, 512), (520, 256)):
        r = ref_attention.RotaryPositionEmbedding(64, scale_base=sb)
        q = T(uniform_pm(f"mod/ropeq{n}", (1, 2, n, 64), 1.0))
        k = T(uniform_pm(f"mod/ropek{n}", (1, 2, n, 64), 1.0))
        qo, ko = r(q, k)
        save(f"mod_rope_{n}_{sb}", q=qo, k=ko)

    # attention / transformer (x is (B, N, C) for Attention, (B, C, N) for TransformerBlock)
    xt = T(uniform_pm("mod/xt", (B, 128, 96), 1.0))
    m = shim_attend(ref_unet.Attention(96, 64, 4, 1, context_len=256))
    load_pattern(m)
    save("mod_attention", y=m(xt))
    m = shim_attend(ref_unet.Attention(96, 64, 4, 2, context_len=256))         # grouped-query: 4 query heads on 2 K/V heads (unet.py:132-135)
    load_pattern(m)
    save("mod_attention_gqa", y=m(xt))
    xc = T(uniform_pm("mod/xc", (B, 96, 128), 1.0))
    m = shim_attend(ref_unet.TransformerBlock(96, attn_dim_head=64, attn_heads=4, attn_kv_heads=1, attn_context_len=256))
    load_pattern(m)
    save("mod_transformer", y=m(xc))

    # UNetBlocks: down (with Downsample), down last (Parallel), up (with Upsample)
    te = T(uniform_pm("mod/te", (B, 64), 1.0))
    ce = T(uniform_pm("mod/ce", (B, 64), 1.0))
    xb = T(uniform_pm("mod/xb", (B, 64, 64), 1.0))
    m = shim_attend(ref_unet.UNetBlock(64, 96, 64, 64, 0, 2, 1, True, 64, 2, 1, 128))
    load_pattern(m)
    y, s = m(xb, te, ce)
    save("mod_unetblock_down", y=y, skip=s)
    m = shim_attend(ref_unet.UNetBlock(64, 96, 64, 64, 1, 2, 1, True, 64, 2, 1, 128))
    load_pattern(m)
    y, s = m(xb, te, ce)
    save("mod_unetblock_down_last", y=y, skip=s)
    xu = T(uniform_pm("mod/xu", (B, 64 + 48, 64), 1.0))
    m = shim_attend(ref_unet.UNetBlock(64, 48, 64, 64, 0, 2, 1, False, 64, 2, 1, 128))
    load_pattern(m)
    y, s = m(xu, te, ce)
    save("mod_unetblock_up", y=y, skip=s)

    m = shim_attend(ref_unet.AudioEncoder(96, 96, dim_h_mult=(1, 2), num_layer_blocks=(1, 1),
                                          cross_embed_kernel_sizes=(3, 7, 15), attn_dim_head=64, attn_heads=2,
                                          attn_kv_heads=1))
    load_pattern(m)
    save("mod_audio_encoder", y=m(T(uniform_pm("mod/xae", (B, 96, 64), 1.0))))

    # round 2: the squeeze-excite gate (residual.py:40-59) and a ResidualBlock built with use_gca=False (residual.py:116),
    # with the gradients of its input and of two of its parameters (the reference's autograd)
    m = ref_residual.SqueezeExcite(48, 48)
    load_pattern(m)
    save("mod_squeeze_excite", y=m(x))
    m = ref_residual.ResidualBlock(48, 80, 64, 64, use_gca=False)
    load_pattern(m)
    xg = x.clone().requires_grad_()
    y = m(xg, t, c)
    gy = T(uniform_pm("mod/gy_se", tuple(y.shape), 1.0))
    y.backward(gy)
    save("mod_resblock_se", y=y, dx=xg.grad, dw_se0=m.se.layers[0].weight.grad, dw_proj1=m.block1.proj.weight.grad)
    # Block(norm=False): nn.Identity in place of the GroupNorm (residual.py:71), with FiLM, and its gradients
    m = ref_residual.Block(48, 80, norm=False)
    load_pattern(m)
    xg = x.clone().requires_grad_()
    ssg = tuple(v.clone().requires_grad_() for v in ss)
    y = m(xg, scale_shift=ssg)
    y.backward(T(uniform_pm("mod/gy_nonorm", tuple(y.shape), 1.0)))
    save("mod_block_nonorm", y_plain=m(x), y_film=y, dx=xg.grad, dw=m.proj.weight.grad, db=m.proj.bias.grad, dscale=ssg[0].grad, dshift=ssg[1].grad)


# --------------------------------------------------------------------------------------
# UNet-level cases
# --------------------------------------------------------------------------------------

UNET_CASES = {
    # BASELINE config 1 (tiny CPU case): C=32, 2 levels x 2 blocks, L=512, batch 2.  3-kernel stem needs
    # dim_h > 72 with 96 audio bins so this one uses the single-kernel stem.
    "unet_tiny": dict(cfg=dict(dim_in_x=6, dim_in_a=96, dim_in_c=5, dim_h=32, dim_h_mult=(1, 2), num_layer_blocks=(2, 2),
                               num_middle_transformers=1, cross_embed_kernel_sizes=(3,), attn_dim_head=64, attn_heads=2,
                               attn_kv_heads=1, attn_context_len=512), B=2, L=512, L_odd=500),
    # SURVEY §8c's suggested head geometry (dim_head 16, 4 heads) with the default 3-kernel stem
    "unet_small16": dict(cfg=dict(dim_in_x=6, dim_in_a=96, dim_in_c=5, dim_h=96, dim_h_mult=(1, 2, 3), num_layer_blocks=(1, 2, 1),
                                  num_middle_transformers=1, cross_embed_kernel_sizes=(3, 7, 15), attn_dim_head=16,
                                  attn_heads=4, attn_kv_heads=1, attn_context_len=256), B=2, L=256, L_odd=250),
    # four levels, default stem, d=64 heads: the structure of the full model at 1/2 width and 1 block per level
    "unet_mid": dict(cfg=dict(dim_in_x=6, dim_in_a=96, dim_in_c=5, dim_h=128, dim_h_mult=(1, 2, 3, 4), num_layer_blocks=(1, 1, 1, 1),
                              num_middle_transformers=1, cross_embed_kernel_sizes=(3, 7, 15), attn_dim_head=64,
                              attn_heads=4, attn_kv_heads=1, attn_context_len=1024), B=2, L=1024, L_odd=1000),
}

GRAD_SLICE = 24


def unet_cases() -> None:
    meta = {}
    for name, spec in UNET_CASES.items():
        print(name)
        cfg, B, L = spec["cfg"], spec["B"], spec["L"]
        net = shim_attend(ref_unet.UNet(**cfg))
        load_pattern(net)
        net.train()
        x, a, c, t, noise = (T(v) for v in synth_inputs(name, B, L))
        # forward, cond kept / cond dropped / mixed mask via where() on hand-made mask equivalent
        with torch.no_grad():
            y_cond = net(x, a, t, c, cond_drop_prob=0.0)
            y_null = net(x, a, t, c, cond_drop_prob=1.0)
            Lo = spec["L_odd"]
            y_odd = net(x[..., :Lo], a[..., :Lo], t, c, cond_drop_prob=0.0)
        # training loss as diffusion.py:96-111 computes it (add_noise restated: diffusers absent), cond kept
        betas = torch.linspace(1e-4, 0.02, 1000, dtype=torch.float32)
        acp = torch.cumprod(1.0 - betas, 0)
        xn = acp[t].sqrt()[:, None, None] * x + (1 - acp[t]).sqrt()[:, None, None] * noise
        net.zero_grad()
        pred = net(xn, a, t, c, cond_drop_prob=0.0)
        loss = torch.nn.functional.mse_loss(pred, noise)
        loss.backward()
        names = [k for k, _ in net.named_parameters()]
        gnorm = np.array([p.grad.norm().item() for _, p in net.named_parameters()], dtype=np.float64)
        gslice = {f"g/{k}": p.grad.flatten()[:GRAD_SLICE].clone() for k, p in net.named_parameters()
                  if k.endswith(("final_conv.weight", "init_x.convs.0.weight", "down_layers.0.resnets.0.block1.proj.weight",
                                 "down_layers.0.transformers.0.attn.to_q.weight", "down_layers.0.transformers.0.attn.to_kv.weight",
                                 "middle_transformer.0.ff.0.weight", "audio_encoder.layers.0.init_resnet.se.to_k.weight",
                                 "up_layers.0.init_resnet.mlp.1.weight", "null_cond", "time_mlp.1.weight",
                                 "down_layers.0.sampler.conv.weight", "up_layers.0.sampler.conv.weight"))}
        save(name, y_cond=y_cond, y_null=y_null, y_odd=y_odd, pred=pred, loss=loss, grad_norms=gnorm, **gslice)
        meta[name] = dict(cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, B=B, L=L,
                          L_odd=spec["L_odd"], param_names=names, n_params=int(sum(p.numel() for p in net.parameters())))
    (HERE / "unet_cases.json").write_text(json.dumps(meta, indent=1))


def _unet_outputs(net, name, B, L, autocast: bool):
    """Forward (cond kept) + the training loss and every parameter gradient, in fp32 or under the trainer's autocast context."""
    import contextlib
    x, a, c, t, noise = (T(v) for v in synth_inputs(name, B, L))
    ctx = (lambda: torch.autocast("cpu", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
    with torch.no_grad(), ctx():
        y_cond = net(x, a, t, c, cond_drop_prob=0.0).float()
    betas = torch.linspace(1e-4, 0.02, 1000, dtype=torch.float32)
    acp = torch.cumprod(1.0 - betas, 0)
    xn = acp[t].sqrt()[:, None, None] * x + (1 - acp[t]).sqrt()[:, None, None] * noise
    net.zero_grad()
    with ctx():                                            # trainer.py:295: model(...) (loss included) inside accelerator.autocast()
        pred = net(xn, a, t, c, cond_drop_prob=0.0)
        loss = torch.nn.functional.mse_loss(pred, noise)
    loss.backward()
    grads = {k: p.grad.detach().float().clone() for k, p in net.named_parameters()}
    return y_cond, pred.detach().float(), loss.detach().float(), grads


def unet_autocast_cases() -> None:
    """The reference's OWN bf16 arithmetic (trainer.py:295,374: `accelerator.autocast()`, mixed precision bf16 by default) on the
    same patterned weights and inputs as the fp32 fixtures: output, loss, per-parameter gradient norms, and how far each of them sits
    from the reference's fp32 run -- the floor the timed (bf16) mode of the HIP path is held to (round-4 review, missing 2)."""
    for name in ("unet_tiny", "unet_mid"):
        spec = UNET_CASES[name]
        cfg, B, L = spec["cfg"], spec["B"], spec["L"]
        net = shim_attend(ref_unet.UNet(**cfg))
        load_pattern(net)
        net.train()
        y32, p32, l32, g32 = _unet_outputs(net, name, B, L, autocast=False)
        y16, p16, l16, g16 = _unet_outputs(net, name, B, L, autocast=True)
        rel = lambda u, v: float((u - v).norm() / v.norm().clamp_min(1e-30))       # noqa: E731
        names = [k for k, _ in net.named_parameters()]
        f32, f16 = torch.cat([g32[k].flatten() for k in names]), torch.cat([g16[k].flatten() for k in names])
        save(f"{name}_autocast", y_cond=y16, pred=p16, loss=l16, loss_fp32=l32,
             grad_norms=np.array([g16[k].norm().item() for k in names], dtype=np.float64),
             grad_dist=np.array([rel(g16[k], g32[k]) for k in names], dtype=np.float64),   # per parameter: autocast vs fp32, rel-L2
             out_dist=np.float64(rel(y16, y32)), pred_dist=np.float64(rel(p16, p32)), flat_grad_dist=np.float64(rel(f16, f32)),
             **{f"g/{k}": g16[k].flatten()[:GRAD_SLICE] for k in names
                if k.endswith(("final_conv.weight", "down_layers.0.transformers.0.attn.to_q.weight", "middle_transformer.0.ff.0.weight"))})
        print(f"  {name}: reference autocast vs reference fp32: output {rel(y16, y32):.3e}, pred {rel(p16, p32):.3e}, "
              f"flat gradient {rel(f16, f32):.3e}, loss {abs(l16.item() - l32.item()) / l32.item():.2e}")


def inventory() -> None:
    """State-dict key/shape inventory of the full default model (dim_h=256): 1,239 entries."""
    net = ref_unet.UNet(6, 96, 5, 256)
    sd = net.state_dict()
    inv = {k: list(v.shape) for k, v in sd.items()}
    (HERE / "state_dict_dim256.json").write_text(json.dumps(inv))
    print(f"inventory: {len(inv)} keys, {sum(int(np.prod(s)) for s in inv.values())} elements")


if __name__ == "__main__":
    assert os.path.isdir("/root/reference"), "generator must run next to the reference"
    only = sys.argv[1:]
    for fn in (module_cases, unet_cases, unet_autocast_cases, inventory):
        if not only or fn.__name__ in only:
            fn()
