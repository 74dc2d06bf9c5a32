"""LoRA / DoRA fine-tune path (reference trainer_peft.py:236-244, osu_fusion/modules/lora_layers.py).

CPU: the oracle's literal three-convolution formula == its merged (effective-weight) form, which is what the HIP path runs; the
injected module tree, key names and merge / unload bookkeeping.  GPU (-m gpu): kernels and adapted modules vs the oracle.
Tolerances: fp32 compute 1e-3 (north_star), bf16 compute 3e-2 -- same as the un-adapted modules."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: N812

from oracle import diffusion_oracle as DO
from oracle import lora_oracle as LO
from oracle import unet_oracle as O
from osufusion_amd.modules import lora_layers as LL
from osufusion_amd.modules import unet as U
from osufusion_amd.pattern import param_pattern, synth_inputs, uniform_pm

TINY = dict(dim_h=32, dim_h_mult=(1, 2), num_layer_blocks=(2, 2), num_middle_transformers=1, cross_embed_kernel_sizes=(3,),
            attn_dim_head=64, attn_heads=2, attn_kv_heads=1, attn_context_len=512)


def _adapter_tensors(tag, w_shape, r, base_w, dora=True):
    """Deterministic non-trivial adapter state: A, B ~ U(+-1/sqrt(fan_in)); magnitude = ||W + BA|| * (1 + U(+-0.2))."""
    sa, sb, sm = LO.adapter_shapes(tuple(w_shape), r)
    a = torch.from_numpy(param_pattern(tag + ".lora_A.default.weight", sa))
    b = torch.from_numpy(uniform_pm(tag + ".lora_B", sb, 1.0 / np.sqrt(r)))
    if not dora:
        return a, b, None
    norm = LO.weight_norm(base_w, LO.delta_weight(a, b, base_w), 1.0)
    m = (norm * (1.0 + torch.from_numpy(uniform_pm(tag + ".mag", (w_shape[0],), 0.2)))).reshape(sm)
    return a, b, m


# ---------------------------------------------------------------------------------------------------------
# CPU
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dora", [True, False])
@pytest.mark.parametrize("conv", [True, False])
def test_oracle_literal_formula_equals_effective_weight_form(dora, conv):
    torch.manual_seed(0)
    O_, I, r, s = 12, 10, 8, 0.75
    w = torch.randn(O_, I, 3) * 0.2 if conv else torch.randn(O_, I) * 0.2
    bias = torch.randn(O_) * 0.1
    a, b, m = _adapter_tensors("t", w.shape, r, w, dora)
    x = torch.randn(2, I, 20) if conv else torch.randn(2, 20, I)
    leaves = [t.clone().double().requires_grad_() for t in ([a, b, m] if dora else [a, b])] + [x.clone().double().requires_grad_()]
    wd, bd = w.double(), bias.double()

    def run(form):
        aa, bb = leaves[0], leaves[1]
        mm = leaves[2] if dora else None
        xx = leaves[-1]
        if form == "literal":
            y = LO.lora_conv1d(xx, wd, bd, aa, bb, mm, s) if conv else LO.lora_linear(xx, wd, bd, aa, bb, mm, s)
        else:
            we = LO.effective_weight(wd, aa, bb, mm, s)
            y = F.conv1d(xx, we, bd, padding=1) if conv else F.linear(xx, we, bd)
        gs = torch.autograd.grad((y * torch.cos(torch.arange(y.numel(), dtype=torch.float64).reshape(y.shape))).sum(), leaves)
        return y.detach(), gs

    y1, g1 = run("literal")
    y2, g2 = run("effective")
    assert torch.allclose(y1, y2, rtol=1e-10, atol=1e-12)
    for u, v in zip(g1, g2):
        assert torch.allclose(u, v, rtol=1e-9, atol=1e-11)


def test_injected_tree_keys_and_bookkeeping():
    net = U.UNet(6, 96, 5, **TINY)
    base_keys = list(net.state_dict())
    LL.get_peft_model(net, LL.LoraConfig(r=8, lora_alpha=8, use_dora=True))
    targets = LO.target_names(base_keys)
    assert len(targets) == 68 and len(LL.lora_modules(net)) == 68          # 24 ResidualBlocks x 2 convs + 10 attentions x 2 linears
    sd = net.state_dict()
    for t in targets:
        for suffix in ("base_layer.weight", "lora_A.default.weight", "lora_B.default.weight", "lora_magnitude_vector.default.weight"):
            assert f"{t}.{suffix}" in sd
        sa, sb, sm = LO.adapter_shapes(tuple(sd[f"{t}.base_layer.weight"].shape), 8)
        assert tuple(sd[f"{t}.lora_A.default.weight"].shape) == sa and tuple(sd[f"{t}.lora_B.default.weight"].shape) == sb
        assert tuple(sd[f"{t}.lora_magnitude_vector.default.weight"].shape) == sm
        # init: B = 0 and magnitude = ||W||  =>  the adapted layer is its base layer (lora_layers.py:183-197)
        w = sd[f"{t}.base_layer.weight"]
        assert torch.count_nonzero(sd[f"{t}.lora_B.default.weight"]) == 0
        assert torch.allclose(sd[f"{t}.lora_magnitude_vector.default.weight"].reshape(-1), w.reshape(w.shape[0], -1).norm(dim=1))
    trainable = [n for n, p in net.named_parameters() if p.requires_grad]
    assert trainable and all("lora_" in n for n in trainable)              # base frozen, adapters train (trainer_peft.py:244-246)
    ad = LL.get_adapter_state_dict(net)
    assert len(ad) == 3 * 68 and all(k.startswith("base_model.model.") and ".default" not in k for k in ad)
    assert sum(k.endswith(".lora_magnitude_vector") for k in ad) == 68     # peft 0.12 stores the magnitude without ".weight"
    # round trip into a fresh injected model
    net2 = U.UNet(6, 96, 5, **TINY)
    LL.get_peft_model(net2, LL.LoraConfig(r=8, lora_alpha=8, use_dora=True))
    for v in ad.values():
        v.add_(0.25)
    LL.load_adapter_state_dict(net2, ad)
    for k, v in LL.get_adapter_state_dict(net2).items():
        assert torch.equal(v, ad[k])
    with pytest.raises(ValueError):
        LL.get_peft_model(nn.Sequential(nn.ReLU()), LL.LoraConfig(r=8))
    with pytest.raises(ValueError):
        LL.LoraConv1d(nn.Conv1d(4, 4, 3, padding=1), "default", r=0)


@pytest.mark.parametrize("dora", [True, False])
def test_merge_unmerge_and_unload_match_oracle(dora):
    net = U.UNet(6, 96, 5, **TINY)
    base = {k: v.clone() for k, v in net.state_dict().items()}
    LL.get_peft_model(net, LL.LoraConfig(r=8, lora_alpha=16, use_dora=dora))
    mods = dict(net.named_modules())
    expect = {}
    for t in LO.target_names(list(base)):
        m = mods[t]
        w = base[t + ".weight"]
        a, b, mag = _adapter_tensors(t, w.shape, 8, w, dora)
        with torch.no_grad():
            m.lora_A["default"].weight.copy_(a); m.lora_B["default"].weight.copy_(b)
            if dora:
                m.lora_magnitude_vector["default"].weight.copy_(mag)
        expect[t] = LO.merged_weight(w, a, b, mag, 2.0)
        assert torch.allclose(m.get_delta_weight("default"), 2.0 * LO.delta_weight(a, b, w), atol=1e-6)
    for t, m in mods.items():
        if isinstance(m, LL._LoraBase):
            m.merge()
            assert m.merged and m.adapter() is None
            assert torch.allclose(m.base_layer.weight, expect[t], rtol=1e-5, atol=1e-6)
            m.unmerge()
            assert torch.allclose(m.base_layer.weight, base[t + ".weight"], rtol=1e-4, atol=1e-6)
    LL.merge_and_unload(net)
    sd = net.state_dict()
    assert list(sd) == list(base)                                           # the reference's plain key layout again
    for t, w in expect.items():
        assert torch.allclose(sd[t + ".weight"], w, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


@gpu
@pytest.mark.parametrize("shape,r,dora", [((40, 24, 3), 8, True), ((64, 2048, 3), 32, True), ((96, 32), 16, True), ((33, 20, 7), 8, False)])
def test_dora_effective_kernel(shape, r, dora):
    from osufusion_amd import ops
    w = torch.from_numpy(uniform_pm("eff.w", shape, 0.3))
    a, b, m = _adapter_tensors("eff", shape, r, w, dora)
    ref = LO.effective_weight(w.double(), a.double(), b.double(), m.double() if dora else None, 0.5)
    weff, g = ops.dora_effective(w.to(DEV), a.to(DEV).contiguous(), b.to(DEV).contiguous(), m.reshape(-1).to(DEV) if dora else None, 0.5)
    assert _rel(weff, ref) < 1e-6
    if dora:
        gref = m.double().reshape(-1) / LO.weight_norm(w.double(), LO.delta_weight(a.double(), b.double(), w.double()), 0.5)
        assert _rel(g, gref) < 1e-6
    else:
        assert torch.equal(g.cpu(), torch.ones(shape[0]))


@gpu
@pytest.mark.parametrize("shape,r,dora", [((40, 24, 3), 8, True), ((64, 2048, 3), 32, True), ((96, 32), 16, True), ((33, 20, 7), 8, False),
                                          ((70, 50, 3), 16, True)])
def test_adapted_pack_and_gain_kernels(shape, r, dora):
    """The training path never forms the fp32 effective weight: osuf_dora_gain + osuf_pack_weight_adapted must give the GEMM
    operand layouts of the oracle's effective weight (fp32 packs to 1e-6, bf16 packs to one rounding), g, and (s g B)^T."""
    from osufusion_amd import ops
    s = 0.5
    w = torch.from_numpy(uniform_pm("eff.w", shape, 0.3))
    a, b, m = _adapter_tensors("eff", shape, r, w, dora)
    ref = LO.effective_weight(w.double(), a.double(), b.double(), m.double() if dora else None, s)
    ref3 = ref if ref.dim() == 3 else ref.unsqueeze(-1)
    k = ref3.shape[2]
    wd, ad_, bd = w.to(DEV), a.to(DEV).contiguous(), b.to(DEV).contiguous()
    g, t32, t16 = ops.dora_gain(wd, ad_, bd, m.reshape(-1).to(DEV) if dora else None, s)
    gref = (m.double().reshape(-1) / LO.weight_norm(w.double(), LO.delta_weight(a.double(), b.double(), w.double()), s)) if dora \
        else torch.ones(shape[0], dtype=torch.float64)
    assert _rel(g, gref) < 1e-6
    sgbt = (s * gref[:, None] * b.double().reshape(shape[0], r)).t()
    assert _rel(t32[0], sgbt) < 1e-6 and _rel(t16[0].float(), sgbt) < 4e-3
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 4e-3)):
        fwd, dgr = ops.pack_weight(wd, dt, "same", adapt=(ad_, bd, g if dora else None, s))
        assert fwd.shape == (k, shape[0], shape[1]) and dgr.shape == (k, shape[1], shape[0])
        assert _rel(fwd.float(), ref3.permute(2, 0, 1)) < tol
        assert _rel(dgr.float(), ref3.flip(2).permute(2, 1, 0)) < tol
    if k == 3:                                              # Downsample / Upsample operand layouts of an adapted k3 conv
        for kind in ("down", "up"):
            want = ops.pack_weight(ref.float().to(DEV).contiguous(), torch.float32, kind)[1]
            got = ops.pack_weight(wd, torch.float32, kind, adapt=(ad_, bd, g if dora else None, s))[1]
            assert _rel(got, want) < 1e-6


@gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("conv,dora", [(True, True), (True, False), (False, True)])
def test_adapted_layer_vs_oracle(dtype, tol, conv, dora):
    import osufusion_amd as oa
    O_, I, r, L = 48, 40, 8, 72
    base = nn.Conv1d(I, O_, 3, padding=1) if conv else nn.Linear(I, O_)
    with torch.no_grad():
        base.weight.copy_(torch.from_numpy(uniform_pm("al.w", tuple(base.weight.shape), 0.15)))
        base.bias.copy_(torch.from_numpy(uniform_pm("al.b", (O_,), 0.1)))
    w, bias = base.weight.detach().clone(), base.bias.detach().clone()
    a, b, m = _adapter_tensors("al", w.shape, r, w, dora)
    cls = LL.LoraConv1d if conv else LL.LoraLinear
    mod = cls(base, "default", r=r, lora_alpha=2 * r, use_dora=dora).to(DEV)
    with torch.no_grad():
        mod.lora_A["default"].weight.copy_(a); mod.lora_B["default"].weight.copy_(b)
        if dora:
            mod.lora_magnitude_vector["default"].weight.copy_(m)
    x = torch.from_numpy(uniform_pm("al.x", (2, I, L) if conv else (2, L, I), 1.0))
    wgt = torch.cos(torch.arange(2 * O_ * L, dtype=torch.float32)).reshape((2, O_, L) if conv else (2, L, O_))
    # oracle (literal reference formula), fp64
    leaves = [t.clone().double().requires_grad_() for t in ([a, b, m] if dora else [a, b])] + [x.clone().double().requires_grad_()]
    mm = leaves[2] if dora else None
    yo = LO.lora_conv1d(leaves[-1], w.double(), bias.double(), leaves[0], leaves[1], mm, 2.0) if conv \
        else LO.lora_linear(leaves[-1], w.double(), bias.double(), leaves[0], leaves[1], mm, 2.0)
    go = torch.autograd.grad((yo * wgt.double()).sum(), leaves)
    xg = x.to(DEV).requires_grad_()
    with oa.forced_compute_dtype(dtype):
        y = mod(xg)
        (y.float() * wgt.to(DEV)).sum().backward()
    assert _rel(y, yo) < tol
    got = [mod.lora_A["default"].weight.grad, mod.lora_B["default"].weight.grad] + \
        ([mod.lora_magnitude_vector["default"].weight.grad] if dora else []) + [xg.grad]
    for name, u, v in zip(["dA", "dB"] + (["dm"] if dora else []) + ["dx"], got, go):
        assert u is not None and u.shape == v.shape, name
        assert _rel(u, v) < (tol if dtype == torch.float32 else 5e-2), (name, _rel(u, v))
    assert base.weight.grad is None                                        # the base weight is never differentiated on this path


@gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 3e-2)])
def test_unet_dora_finetune_step_vs_oracle(dtype, tol):
    """Tiny UNet, every trainer_peft target adapted with DoRA r=8: loss and all 204 adapter gradients vs the golden-pinned UNet
    oracle evaluated on effective weights; base parameters receive no gradient; one Trainer step moves only adapter tensors."""
    import osufusion_amd as oa
    from osufusion_amd.models.diffusion import OsuFusion
    from osufusion_amd.train import Trainer
    cfg = O.UNetConfig(dim_in_x=6, dim_in_a=96, dim_in_c=5, **TINY)
    p = O.make_params(cfg, prefix="unet.")
    model = OsuFusion(TINY["dim_h"], **{k: v for k, v in TINY.items() if k != "dim_h"})
    model.unet.load_state_dict({k[len("unet."):]: v for k, v in p.items()}, strict=True)
    model.to(DEV)
    LL.get_peft_model(model, LL.LoraConfig(r=8, lora_alpha=8, use_dora=True))
    mods = dict(model.named_modules())
    adapters = {}
    for t in LO.target_names(list(p)):
        a, b, m = _adapter_tensors(t, p[t + ".weight"].shape, 8, p[t + ".weight"], True)
        with torch.no_grad():
            mods[t].lora_A["default"].weight.copy_(a); mods[t].lora_B["default"].weight.copy_(b)
            mods[t].lora_magnitude_vector["default"].weight.copy_(m)
        adapters[t] = tuple(v.clone().requires_grad_() for v in (a, b, m))
    B_, L = 2, 256
    x, a_, c, t_, noise = (torch.from_numpy(v) for v in synth_inputs("lora_tiny", B_, L))
    loss_ref = DO.training_loss(LO.effective_params(p, adapters, 1.0), cfg, x, a_, c, noise, t_, cond_drop_prob=0.0,
                                mode="fp32" if dtype == torch.float32 else "bf16")
    loss_ref.backward()
    with oa.forced_compute_dtype(dtype):
        loss = model.loss_with(x.to(DEV), a_.to(DEV), c.to(DEV), noise.to(DEV), t_.to(DEV), cond_drop_prob=0.0)
        loss.backward()
    assert abs(loss.item() - loss_ref.item()) < tol * abs(loss_ref.item())
    gmax = max(v.grad.abs().max().item() for vs in adapters.values() for v in vs)
    worst, worst_k = 0.0, ""
    for t, (ra, rb, rm) in adapters.items():
        m = mods[t]
        for nm, got, ref in (("A", m.lora_A["default"].weight.grad, ra.grad), ("B", m.lora_B["default"].weight.grad, rb.grad),
                             ("m", m.lora_magnitude_vector["default"].weight.grad, rm.grad)):
            assert got is not None, (t, nm)
            e = ((got.cpu() - ref).abs().max() / (ref.abs().max() + 1e-3 * gmax)).item()
            if e > worst:
                worst, worst_k = e, f"{t}.{nm}"
    assert worst < (2e-2 if dtype == torch.float32 else 1.5e-1), (worst, worst_k)
    assert all(q.grad is None for n, q in model.named_parameters() if "lora_" not in n)
    # one fused optimizer step over the adapter-only flat buffer
    before = {n: q.detach().clone() for n, q in model.named_parameters()}
    for q in model.parameters():
        q.grad = None
    from osufusion_amd import functional as Fn
    try:
        tr = Trainer(model, lr=1e-3, compute_dtype=dtype)
        assert tr.flat.numel < 0.2 * sum(q.numel() for q in model.parameters())
        l2, gn = tr.step(x.to(DEV), a_.to(DEV), c.to(DEV), noise.to(DEV), t_.to(DEV))
        assert torch.isfinite(l2).item() and torch.isfinite(gn).item() and gn.item() > 0
        for n, q in model.named_parameters():
            changed = not torch.equal(q.detach(), before[n])
            assert changed == ("lora_" in n), n
    finally:
        Fn.enable_direct_grads(False)


@gpu
def test_merged_model_equals_adapted_model_on_gpu():
    """PeftModel.merge_and_unload (trainer_peft.py:161-164): after folding the DoRA adapters into the base weights the plain UNet
    computes what the adapted one did (inference path of a fine-tuned model), and the module tree is the reference's again."""
    import osufusion_amd as oa
    cfg = O.UNetConfig(dim_in_x=6, dim_in_a=96, dim_in_c=5, **TINY)
    p = O.make_params(cfg)
    net = U.UNet(6, 96, 5, **TINY)
    net.load_state_dict(p, strict=True)
    net.to(DEV).eval()
    n_keys = len(net.state_dict())
    LL.get_peft_model(net, LL.LoraConfig(r=8, lora_alpha=16, use_dora=True))
    mods = dict(net.named_modules())
    for t in LO.target_names(list(p)):
        a, b, m = _adapter_tensors(t, p[t + ".weight"].shape, 8, p[t + ".weight"], True)
        with torch.no_grad():
            mods[t].lora_A["default"].weight.copy_(a); mods[t].lora_B["default"].weight.copy_(b)
            mods[t].lora_magnitude_vector["default"].weight.copy_(m)
    x, a_, c, t_, _ = (torch.from_numpy(v).to(DEV) for v in synth_inputs("lora_merge", 2, 256))
    with torch.no_grad(), oa.forced_compute_dtype(torch.float32):
        y_adapted = net(x, a_, t_, c)
        base_only = None
        for m in LL.lora_modules(net):
            m.disable_adapters = True
        base_only = net(x, a_, t_, c)
        for m in LL.lora_modules(net):
            m.disable_adapters = False
        LL.merge_and_unload(net)
        assert len(net.state_dict()) == n_keys and not LL.lora_modules(net)
        y_merged = net(x, a_, t_, c)
    assert _rel(y_merged, y_adapted) < 1e-3
    assert _rel(base_only, y_adapted) > 1e-2                               # the adapters do change the function in this fixture
