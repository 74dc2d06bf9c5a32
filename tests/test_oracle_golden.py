"""CPU: the oracle (oracle/) against the golden vectors produced by the imported reference
(tests/golden/make_golden.py).  fp32 mode, rtol 1e-5-class tolerances (same torch ops, other order)."""
import json

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as D
from oracle import unet_oracle as O
from osufusion_amd.pattern import param_pattern, synth_inputs, uniform_pm

NM = O.Numerics("fp32")
B = 2


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def P(shapes, pre="m."):
    return {pre + k: T(param_pattern(k, tuple(s))) for k, s in shapes}


def close(a, b, tol=2e-5):
    a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
    assert err < tol, f"max-rel err {err:.3e}"


def G(golden_dir, name):
    return np.load(golden_dir / f"{name}.npz")


def strip(shapes, pre):
    return [(k[len(pre) + 1:], s) for k, s in shapes]


def test_block_and_residual(golden_dir):
    x = T(uniform_pm("mod/x48", (B, 48, 96), 1.0))
    t = T(uniform_pm("mod/t", (B, 64), 1.0))
    c = T(uniform_pm("mod/c", (B, 64), 1.0))
    g = G(golden_dir, "mod_block")
    p = P([("proj.weight", (80, 48, 3)), ("proj.bias", (80,)), ("norm.weight", (80,)), ("norm.bias", (80,))], "m.")
    ss = (T(uniform_pm("mod/scale", (B, 80, 1), 0.5)), T(uniform_pm("mod/shift", (B, 80, 1), 0.5)))
    close(O.block(p, "m", x, None, NM), g["y_plain"])
    close(O.block(p, "m", x, ss, NM), g["y_film"])

    p = P(strip(O._resblock_shapes("m", 48, 80, 128), "m"))
    close(O.residual_block(p, "m", x, t, c, NM), G(golden_dir, "mod_resblock_film")["y"])
    p = P(strip(O._resblock_shapes("m", 48, 48, None), "m"))
    close(O.residual_block(p, "m", x, None, None, NM), G(golden_dir, "mod_resblock_plain")["y"])

    p = P([("to_k.weight", (1, 48, 1)), ("to_k.bias", (1,)), ("layers.0.weight", (24, 48, 1)), ("layers.0.bias", (24,)),
           ("layers.2.weight", (48, 24, 1)), ("layers.2.bias", (48,))])
    close(O.global_context(p, "m", x, NM), G(golden_dir, "mod_global_context")["y"])

    # squeeze-excite gate and a ResidualBlock(use_gca=False), with the reference's own gradients (round 2 fixtures)
    p = P([("layers.0.weight", (24, 48, 1)), ("layers.0.bias", (24,)), ("layers.2.weight", (48, 24, 1)), ("layers.2.bias", (48,))])
    close(O.squeeze_excite(p, "m", x, NM), G(golden_dir, "mod_squeeze_excite")["y"])
    shapes = [(k, v) for k, v in strip(O._resblock_shapes("m", 48, 80, 128), "m") if "se.to_k" not in k]
    p = {k: v.clone().requires_grad_() for k, v in P(shapes).items()}
    xg = x.clone().requires_grad_()
    g = G(golden_dir, "mod_resblock_se")
    y = O.residual_block(p, "m", xg, t, c, NM)
    close(y, g["y"])
    y.backward(T(uniform_pm("mod/gy_se", tuple(y.shape), 1.0)))
    close(xg.grad, g["dx"], tol=2e-4)
    close(p["m.se.layers.0.weight"].grad, g["dw_se0"], tol=2e-4)
    close(p["m.block1.proj.weight"].grad, g["dw_proj1"], tol=2e-4)

    # Block(norm=False): identity instead of the GroupNorm
    g = G(golden_dir, "mod_block_nonorm")
    p = {k: v.clone().requires_grad_() for k, v in P([("proj.weight", (80, 48, 3)), ("proj.bias", (80,))], "m.").items()}
    xg = x.clone().requires_grad_()
    ssg = tuple(v.clone().requires_grad_() for v in ss)
    close(O.block(p, "m", x, None, NM), g["y_plain"])
    y = O.block(p, "m", xg, ssg, NM)
    close(y, g["y_film"])
    y.backward(T(uniform_pm("mod/gy_nonorm", tuple(y.shape), 1.0)))
    for got, key in ((xg.grad, "dx"), (p["m.proj.weight"].grad, "dw"), (p["m.proj.bias"].grad, "db"), (ssg[0].grad, "dscale"), (ssg[1].grad, "dshift")):
        close(got, g[key], tol=2e-4)


def test_samplers_and_stems(golden_dir):
    x = T(uniform_pm("mod/x48", (B, 48, 96), 1.0))
    p = P([("conv.weight", (80, 48, 3)), ("conv.bias", (80,))])
    close(O.downsample(p, "m", x, NM), G(golden_dir, "mod_downsample")["y"])
    close(O.upsample(p, "m", x, NM), G(golden_dir, "mod_upsample")["y"])
    p = P([("fns.0.weight", (80, 48, 3)), ("fns.0.bias", (80,)), ("fns.1.weight", (80, 48, 1)), ("fns.1.bias", (80,))])
    close(O.parallel_conv(p, "m", x, NM), G(golden_dir, "mod_parallel")["y"])
    xa = T(uniform_pm("mod/xa", (B, 96, 64), 1.0))
    p = P(strip(O._cross_embed_shapes("m", 96, 128, (3, 7, 15)), "m"))
    close(O.cross_embed(p, "m", xa, (3, 7, 15), NM), G(golden_dir, "mod_cross_embed")["y"])
    x6 = T(uniform_pm("mod/x6", (B, 6, 64), 1.0))
    p = P(strip(O._cross_embed_shapes("m", 6, 128, (3, 7, 15)), "m"))
    close(O.cross_embed(p, "m", x6, (3, 7, 15), NM), G(golden_dir, "mod_cross_embed6")["y"])


def test_embeddings(golden_dir):
    y = O.sinusoidal_embedding(torch.tensor([0, 1, 17, 500, 999], dtype=torch.int64), 128)
    close(y, G(golden_dir, "mod_sinusoidal")["y"], 1e-5)
    for n, sb in ((512, 512), (520, 256)):
        g = G(golden_dir, f"mod_rope_{n}_{sb}")
        q = T(uniform_pm(f"mod/ropeq{n}", (1, 2, n, 64), 1.0))
        k = T(uniform_pm(f"mod/ropek{n}", (1, 2, n, 64), 1.0))
        cos, sin = O.rope_tables(n, 64, sb)
        close(O.apply_rope(q, cos, sin), g["q"], 1e-5)
        close(O.apply_rope(k, cos, sin), g["k"], 1e-5)


def test_attention_transformer(golden_dir):
    cfg = O.UNetConfig(dim_h=96, attn_dim_head=64, attn_heads=4, attn_kv_heads=1)
    xt = T(uniform_pm("mod/xt", (B, 128, 96), 1.0))
    p = P(strip(O._transformer_shapes("m", 96, cfg), "m"))
    # Attention case was generated from a standalone Attention module: keys are local (no "attn." prefix)
    pa = P([(k[len("attn."):], s) for k, s in strip(O._transformer_shapes("m", 96, cfg), "m") if k.startswith("attn.")])
    close(O.attention(pa, "m", xt, cfg, 256, NM), G(golden_dir, "mod_attention")["y"], 2e-5)
    cfg2 = O.UNetConfig(dim_h=96, attn_dim_head=64, attn_heads=4, attn_kv_heads=2)          # grouped-query: head j reads K/V head j mod 2
    pg = P([(k[len("attn."):], s) for k, s in strip(O._transformer_shapes("m", 96, cfg2), "m") if k.startswith("attn.")])
    close(O.attention(pg, "m", xt, cfg2, 256, NM), G(golden_dir, "mod_attention_gqa")["y"], 2e-5)
    xc = T(uniform_pm("mod/xc", (B, 96, 128), 1.0))
    close(O.transformer_block(p, "m", xc, cfg, 256, NM), G(golden_dir, "mod_transformer")["y"], 2e-5)


def test_unet_blocks_and_audio_encoder(golden_dir):
    cfg = O.UNetConfig(dim_h=64, attn_dim_head=64, attn_heads=2, attn_kv_heads=1)
    te = T(uniform_pm("mod/te", (B, 64), 1.0))
    ce = T(uniform_pm("mod/ce", (B, 64), 1.0))
    xb = T(uniform_pm("mod/xb", (B, 64, 64), 1.0))
    for name, li, down, x, dout in (("mod_unetblock_down", 0, True, xb, 96), ("mod_unetblock_down_last", 1, True, xb, 96),
                                    ("mod_unetblock_up", 0, False, T(uniform_pm("mod/xu", (B, 112, 64), 1.0)), 48)):
        p = P(strip(O._unet_block_shapes("m", 64, dout, 128, li, 2, 1, down, cfg), "m"))
        y, s = O.unet_block(p, "m", x, te, ce, cfg, 1, 128, down, NM)
        g = G(golden_dir, name)
        close(y, g["y"], 3e-5)
        close(s, g["skip"], 3e-5)
    cfg = O.UNetConfig(dim_in_a=96, dim_h=96, dim_h_mult=(1, 2), num_layer_blocks=(1, 1), attn_dim_head=64, attn_heads=2)
    shapes = [(k, s) for k, s in O.param_shapes(cfg) if k.startswith("audio_encoder.")]
    p = P([(k[len("audio_encoder."):], s) for k, s in shapes])
    y = O.audio_encoder(p, "m", T(uniform_pm("mod/xae", (B, 96, 64), 1.0)), cfg, NM)
    close(y, G(golden_dir, "mod_audio_encoder")["y"], 3e-5)


@pytest.mark.parametrize("case", ["unet_tiny", "unet_small16", "unet_mid"])
def test_unet_forward_backward(golden_dir, case):
    meta = json.loads((golden_dir / "unet_cases.json").read_text())[case]
    cfgd = {k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["cfg"].items()}
    cfg = O.UNetConfig(**cfgd)
    names = [k for k, _ in O.param_shapes(cfg)]
    assert sorted(names) == sorted(meta["param_names"])          # every nn.Parameter name of the reference
    p = O.make_params(cfg, requires_grad=True)
    assert sum(v.numel() for v in p.values()) == meta["n_params"]
    x, a, c, t, noise = (T(v) for v in synth_inputs(case, meta["B"], meta["L"]))
    g = G(golden_dir, case)
    with torch.no_grad():
        close(O.unet_forward(p, cfg, x, a, t, c, cond_drop_prob=0.0), g["y_cond"], 5e-5)
        close(O.unet_forward(p, cfg, x, a, t, c, cond_drop_prob=1.0), g["y_null"], 5e-5)
        Lo = meta["L_odd"]
        close(O.unet_forward(p, cfg, x[..., :Lo], a[..., :Lo], t, c), g["y_odd"], 5e-5)
    loss = D.training_loss(p, cfg, x, a, c, noise, t, cond_drop_prob=0.0, prefix="")
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    loss.backward()
    gn = np.array([p[k].grad.norm().item() for k in meta["param_names"]])
    ref = g["grad_norms"]
    rel = np.abs(gn - ref) / (ref + 1e-8 * ref.max())
    assert rel.max() < 2e-3, f"grad-norm mismatch {rel.max():.3e} at {meta['param_names'][int(rel.argmax())]}"
    for key in g.files:
        if key.startswith("g/"):
            got = p[key[2:]].grad.flatten()[:24]
            close(got, g[key], 2e-3)


@pytest.mark.parametrize("case", ["unet_tiny", "unet_mid"])
def test_oracle_bf16_emulation_vs_reference_autocast(golden_dir, case):
    """The oracle's bf16 mode (the rounding points of the HIP bf16 path) against the reference's OWN autocast run
    (`{case}_autocast.npz`: the imported reference UNet under torch.autocast("cpu", bfloat16), trainer.py:295,374).  Two different
    placements of bf16 roundings cannot agree to better than either's distance from fp32; what is pinned is that the emulation sits
    as far from fp32 as the reference's autocast does (0.5x .. 1.5x) and no further from the autocast run than 2x that distance."""
    meta = json.loads((golden_dir / "unet_cases.json").read_text())[case]
    cfgd = {k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["cfg"].items()}
    cfg = O.UNetConfig(**cfgd)
    p = O.make_params(cfg)
    x, a, c, t, noise = (T(v) for v in synth_inputs(case, meta["B"], meta["L"]))
    g32, g16 = G(golden_dir, case), G(golden_dir, f"{case}_autocast")
    rel = lambda u, v: float((u.double() - T(v).double()).norm() / T(v).double().norm())       # noqa: E731
    with torch.no_grad():
        y16 = O.unet_forward(p, cfg, x, a, t, c, cond_drop_prob=0.0, mode="bf16")
    ref = float(g16["out_dist"])
    assert abs(rel(T(g16["y_cond"]), g32["y_cond"]) - ref) < 1e-6 * ref + 1e-9      # the fixture's own bookkeeping
    mine = rel(y16, g32["y_cond"])
    assert 0.5 * ref < mine < 1.5 * ref, (mine, ref)
    assert rel(y16, g16["y_cond"]) < 2.0 * ref
    assert g16["grad_dist"].shape == (len(meta["param_names"]),) and np.isfinite(g16["grad_dist"]).all()
    assert 5e-3 < float(g16["flat_grad_dist"]) < 2e-2                                # unet_mid 8.0e-3, unet_tiny 1.07e-2


def test_state_dict_inventory_full_model(golden_dir):
    inv = json.loads((golden_dir / "state_dict_dim256.json").read_text())
    shapes = dict(O.param_shapes(O.UNetConfig(dim_h=256)))
    assert len(inv) == 1239
    assert set(inv) == set(shapes)
    for k, s in inv.items():
        assert tuple(s) == tuple(shapes[k]), k
    assert sum(int(np.prod(s)) for s in shapes.values()) == 343_493_297


def test_ddim_known_answers():
    """SURVEY §8a row 15 known-answer constants (diffusers 0.29.2 absent: parity otherwise unpinned)."""
    acp = D.ddim_alphas_cumprod()
    for i, v in ((0, 0.99989998), (1, 0.99978006), (500, 0.07779665), (999, 4.0358304e-05)):
        assert abs(acp[i].item() - v) / v < 2e-6
    ts = D.ddim_timesteps(35).tolist()
    assert ts[:2] == [952, 924] and ts[-2:] == [28, 0] and len(ts) == 35
    ts = D.ddim_timesteps(50).tolist()
    assert ts[:2] == [980, 960] and ts[-2:] == [20, 0]
    # step algebra: with eps exact and |x0|<=1 the step returns the exact x_{t_prev}
    x0 = torch.rand(2, 6, 16) * 2 - 1
    eps = torch.randn(2, 6, 16)
    t, S = 980, 50
    xt = acp[t].sqrt() * x0 + (1 - acp[t]).sqrt() * eps
    xp = D.ddim_step(eps, t, xt, acp, S)
    want = acp[t - 20].sqrt() * x0 + (1 - acp[t - 20]).sqrt() * eps
    assert torch.allclose(xp, want, atol=2e-3)
    xp0 = D.ddim_step(eps, 0, acp[0].sqrt() * x0 + (1 - acp[0]).sqrt() * eps, acp, S)
    assert torch.allclose(xp0, x0, atol=1e-3)                         # prev_t < 0 -> alpha_prev = 1
