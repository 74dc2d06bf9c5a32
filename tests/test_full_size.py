"""Parity at BASELINE.json's full size: the 343.5 M-parameter UNet (dim_h=256) on B=32 sequences of L=4096 -- the very shapes
bench.py times (N=4096 attention, 256x256 GEMM tiles, split wgrads).  The oracle cannot run 32 samples in seconds, so:
  * samples of a batch are independent in this model (GroupNorm(1, C), per-sample attention), hence two samples picked out of the
    full fp32-mode batch are compared with the oracle run on each of them alone (1e-3 rel-L2, north_star's bound);
  * the bf16 training loss and its 343.5 M gradients at B=32 must be the mean of the same quantities over its four B=8 shards
    (linearity of the mean-reduced loss in the batch; the shards take different tile / split plans than the full batch).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as O

B, L, DIM_H = 32, 4096, 256


def report(name, **vals):
    """Append achieved errors to gpurun_out/parity_metrics.jsonl (quoted in DESIGN.md)."""
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_metrics.jsonl", "a") as f:
        f.write(json.dumps({"test": name, **{k: (v if isinstance(v, (list, str)) else float(v)) for k, v in vals.items()}}) + "\n")


@pytest.fixture(scope="module")
def full():
    from osufusion_amd.models.diffusion import OsuFusion
    torch.manual_seed(0)
    model = OsuFusion(DIM_H)
    with torch.no_grad():
        model.unet.final_conv.weight.normal_(0.0, 0.02)            # the reference zero-inits the head (unet.py:354): give it life
    model = model.to("cuda")
    g = torch.Generator().manual_seed(99)
    x = (torch.randn(B, 6, L, generator=g) * 0.5).clamp_(-1, 1)
    a = torch.randn(B, 96, L, generator=g) * 3 - 10
    c = torch.rand(B, 5, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn(B, 6, L, generator=g)
    return model, x, a, c, t, noise


def test_full_size_forward_samples_vs_oracle(full):
    """Both fp32 compute modes -- "exact" (f32 MFMA) and "x3" (split-bf16 GEMMs, the mode bench.py reports as fp32x3_mode_ms_per_step) --
    against the oracle at north_star's 1e-3."""
    import osufusion_amd as oa
    from osufusion_amd import ops
    model, x, a, c, t, _ = full
    got = {}
    for mm in ("exact", "x3"):
        prev = ops.set_f32_matmul(mm)
        try:
            with torch.no_grad(), oa.forced_compute_dtype(torch.float32):
                got[mm] = model.unet(x.cuda(), a.cuda(), t.cuda(), c.cuda()).cpu()
        finally:
            ops.set_f32_matmul(prev)
        assert got[mm].shape == (B, 6, L) and torch.isfinite(got[mm]).all()
    p = {k: v.detach().float().cpu() for k, v in model.unet.state_dict().items()}
    cfg = O.UNetConfig(dim_h=DIM_H)
    for i in (0, 21):
        with torch.no_grad():
            ref = O.unet_forward(p, cfg, x[i:i + 1], a[i:i + 1], t[i:i + 1], c[i:i + 1])
        for mm in ("exact", "x3"):
            err = ((got[mm][i:i + 1] - ref).norm() / ref.norm()).item()
            report("full_size_forward_vs_oracle", sample=i, f32_matmul=mm, rel_l2=err)
            assert err < 1e-3, (mm, i, err)


def test_full_size_bf16_loss_and_gradients_are_the_mean_over_shards(full):
    import osufusion_amd as oa
    from osufusion_amd import functional as Fn
    from osufusion_amd.train import Trainer
    model, x, a, c, t, noise = full
    x, a, c, t, noise = (v.cuda() for v in (x, a, c, t, noise))
    try:
        trainer = Trainer(model, compute_dtype=torch.bfloat16)

        def run(sl):
            trainer.flat.zero_grad()
            with oa.forced_compute_dtype(torch.bfloat16):
                loss = model.loss_with(x[sl], a[sl], c[sl], noise[sl], t[sl], cond_drop_prob=0.0)
                loss.backward()
            return loss.detach().double(), trainer.flat.grad.detach().clone()

        loss_full, g_full = run(slice(0, B))
        assert torch.isfinite(loss_full) and torch.isfinite(g_full).all() and g_full.abs().max() > 0
        loss_sum, g_sum = 0.0, torch.zeros_like(g_full)
        for s in range(0, B, 8):
            ls, gs = run(slice(s, s + 8))
            loss_sum, g_sum = loss_sum + ls, g_sum + gs
        loss_mean, g_mean = loss_sum / 4, g_sum / 4
        assert abs(loss_full - loss_mean) / abs(loss_mean) < 1e-3
        rel = ((g_full - g_mean).double().norm() / g_mean.double().norm()).item()
        report("full_size_shard_linearity", loss_rel=abs(loss_full - loss_mean) / abs(loss_mean), grad_rel_l2=rel)
        assert rel < 1e-2, rel                                      # bf16 attention / atomics-order noise; a wrong split plan is O(1)
    finally:
        Fn.enable_direct_grads(False)


def test_full_size_one_sample_gradient_vs_oracle(full):
    """The full 343.5 M-parameter UNet on one L=4096 sample: loss and all 1,239 parameter gradients of the HIP train path (direct
    accumulation into the flat buffer, N=4096 pipelined attention backward, 256^2 split-wgrad plans) against the oracle's
    autograd (the CPU restatement pinned to the reference's goldens; fp32 + bf16 SDPA, unet.py:125-141, attention.py:87-101)."""
    import osufusion_amd as oa
    from oracle import diffusion_oracle as DO
    from osufusion_amd import functional as Fn
    from osufusion_amd import ops
    from osufusion_amd.train import Trainer
    model, x, a, c, t, noise = full
    i = 5
    xs, as_, cs, ts, ns = (v[i:i + 1] for v in (x, a, c, t, noise))
    cfg = O.UNetConfig(dim_h=DIM_H)
    p = {k: v.detach().float().cpu().clone().requires_grad_() for k, v in model.state_dict().items()}
    loss_ref = DO.training_loss(p, cfg, xs, as_, cs, ns, ts, cond_drop_prob=0.0)
    loss_ref.backward()
    gref = {k: v.grad for k, v in p.items()}
    ref_flat = torch.cat([g.reshape(-1) for g in gref.values()]).double()
    # The oracle's OWN bf16-autocast emulation on the same sample: its per-parameter distance from its fp32 self is the floor that
    # bf16 arithmetic (not a kernel defect) puts under every parameter; the HIP bf16 gradients are held to a small multiple of it.
    for v in p.values():
        v.grad = None
    DO.training_loss(p, cfg, xs, as_, cs, ns, ts, cond_drop_prob=0.0, mode="bf16").backward()
    gorc = {k: v.grad for k, v in p.items()}
    try:
        trainer = Trainer(model, compute_dtype=torch.float32, reorder_buckets=False)
        names = {id(q): n for n, q in model.named_parameters()}
        # per-parameter errors are taken relative to max(|ref_i|, 1e-4 x the mean parameter-gradient norm): a handful of gradients
        # are exactly zero in exact arithmetic (GlobalContext's to_k.bias: the softmax over L ignores a constant shift)
        floor = 1e-4 * ref_flat.norm().item() / len(gref) ** 0.5
        # (max over parameters, fp32 mode: always one of the audio encoder's attn.to_q weights, whose gradient is ~8 floors small and
        #  sits behind the reference's bf16 cast of q / k / v -- 1.7e-2 .. 2.1e-2 from run to run with the order of the fp32 atomics)
        # fp32 compute mode twice: exact-f32 MFMA GEMMs and the split-bf16 "x3" GEMMs (same bounds), then the timed bf16 mode
        for mode, mm, tol_loss, tol_flat, tol_norm_max, tol_norm_med in ((torch.float32, "exact", 1e-5, 2e-3, 3e-2, 1e-3),
                                                                         (torch.float32, "x3", 1e-5, 2e-3, 3e-2, 1e-3),
                                                                         (torch.bfloat16, "exact", 1e-3, 2e-2, 3e-1, 2e-2)):
            trainer.flat.zero_grad()
            prev_mm = ops.set_f32_matmul(mm)
            try:
                with oa.forced_compute_dtype(mode):
                    loss = model.loss_with(xs.cuda(), as_.cuda(), cs.cuda(), ns.cuda(), ts.cuda(), cond_drop_prob=0.0)
                    loss.backward()
            finally:
                ops.set_f32_matmul(prev_mm)
            got = {names[id(q)]: trainer.flat.grad[o:o + q.numel()].view_as(q).cpu() for q, o in zip(trainer.flat.params, trainer.flat.offsets)}
            assert set(got) == set(gref) and len(got) == 1239
            got_flat = torch.cat([got[k].reshape(-1) for k in gref]).double()
            e_loss = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
            e_flat = ((got_flat - ref_flat).norm() / ref_flat.norm()).item()
            rel_norm = torch.tensor([(got[k].double() - gref[k].double()).norm().item() / max(gref[k].double().norm().item(), floor)
                                     for k in gref])
            tag = ("fp32" if mm == "exact" else "fp32x3") if mode == torch.float32 else "bf16"
            worst = sorted(zip(rel_norm.tolist(), gref), reverse=True)[:3]
            report(f"full_size_gradient_vs_oracle/{tag}", loss_rel=e_loss, flat_grad_rel_l2=e_flat, per_param_rel_l2_max=rel_norm.max(),
                   per_param_rel_l2_median=rel_norm.median(),
                   worst=[(k, round(e, 5), float(gref[k].double().norm() / max(floor, 1e-30))) for e, k in worst])
            assert e_loss < tol_loss, (tag, e_loss)
            assert e_flat < tol_flat, (tag, e_flat)
            assert rel_norm.max() < tol_norm_max and rel_norm.median() < tol_norm_med, (tag, rel_norm.max(), rel_norm.median())
            if mode == torch.bfloat16:
                # the timed mode against the oracle's bf16 emulation, parameter by parameter
                orc_flat = torch.cat([gorc[k].reshape(-1) for k in gref]).double()
                den = {k: max(gref[k].double().norm().item(), floor) for k in gref}
                d_orc = torch.tensor([(gorc[k].double() - gref[k].double()).norm().item() / den[k] for k in gref])       # oracle bf16 vs oracle fp32
                d_hip = rel_norm                                                                                          # HIP bf16 vs oracle fp32
                d_x = torch.tensor([(got[k].double() - gorc[k].double()).norm().item() / den[k] for k in gref])          # HIP bf16 vs oracle bf16
                ratio = d_hip / d_orc.clamp_min(2e-3)
                keys = list(gref)
                iw = int(ratio.argmax())
                ih = int(d_hip.argmax())
                report("full_size_gradient_bf16_floor", oracle_bf16_vs_fp32_flat=((orc_flat - ref_flat).norm() / ref_flat.norm()).item(),
                       hip_bf16_vs_oracle_bf16_flat=((got_flat - orc_flat).norm() / orc_flat.norm()).item(),
                       oracle_per_param_max=d_orc.max(), oracle_per_param_median=d_orc.median(), hip_per_param_max=d_hip.max(),
                       hip_vs_oracle_bf16_per_param_max=d_x.max(), hip_vs_oracle_bf16_per_param_median=d_x.median(),
                       worst_ratio=ratio.max(), worst_ratio_param=keys[iw], hip_worst_param=keys[ih], hip_worst=d_hip[ih],
                       oracle_at_hip_worst=d_orc[ih], ratio_median=ratio.median())
                # every parameter's HIP-bf16 error is within 3x the oracle's own bf16-vs-fp32 error on that parameter (floored at 2e-3):
                # the 0.2 outliers of the table above are parameters on which the reference's autocast is just as far from fp32
                # (measured: worst ratio 1.97, median 0.94; the largest HIP error, 0.204 on an audio-encoder attn.to_q.weight, sits on a
                #  parameter where the oracle's own bf16 run is 0.211 from its fp32 run)
                assert ratio.max() < 2.5, (keys[iw], d_hip[iw].item(), d_orc[iw].item())
                assert ratio.median() < 1.25, ratio.median()
    finally:
        Fn.enable_direct_grads(False)


def test_config4_ddim_step_vs_oracle(full):
    """BASELINE config 4 (inference_gradio.py:105,128 -> diffusion.py:59-77): B=16, L=8192, S=50, cond_scale 2.  The first DDIM step
    (t = 980; 2B-batched classifier-free guidance, cached audio code, fused step kernel) is checked for two samples against the
    oracle's sequential cond / null forwards + scheduler step; the hipGraph-replayed second step must equal the eager one."""
    import osufusion_amd as oa
    from oracle import diffusion_oracle as DO
    model = full[0]
    Bc, Lc, S, cs = 16, 8192, 50, 2.0
    g = torch.Generator().manual_seed(404)
    a = torch.randn(Bc, 96, Lc, generator=g) * 3 - 10
    c = torch.rand(Bc, 5, generator=g) * 2 - 1
    x0 = torch.randn(Bc, 6, Lc, generator=g)
    model.sampling_timesteps = S
    try:
        model.stop_after = 1
        with oa.forced_compute_dtype(torch.float32):
            got = model.sample(a.cuda(), c.cuda(), x0.cuda(), cond_scale=cs).cpu()
        assert got.shape == (Bc, 6, Lc) and torch.isfinite(got).all()
        p = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        cfg = O.UNetConfig(dim_h=DIM_H)
        acp = DO.ddim_alphas_cumprod()
        t0 = int(DO.ddim_timesteps(S)[0])
        for i in (0, 11):
            tb = torch.full((1,), t0, dtype=torch.int64)
            with torch.no_grad():
                cond = O.unet_forward(p, cfg, x0[i:i + 1], a[i:i + 1], tb, c[i:i + 1], cond_drop_prob=0.0, prefix="unet.")
                null = O.unet_forward(p, cfg, x0[i:i + 1], a[i:i + 1], tb, c[i:i + 1], cond_drop_prob=1.0, prefix="unet.")
            want = DO.ddim_step(null + (cond - null) * cs, t0, x0[i:i + 1], acp, S)
            eps_scale = (1 - acp[t0 - 1000 // S]).sqrt().item()               # x_prev = sqrt(a_prev) x0_clamped + eps_scale * eps
            err = ((got[i:i + 1] - want).norm() / want.norm()).item()
            report("config4_first_ddim_step_vs_oracle", sample=i, rel_l2=err, eps_scale=eps_scale)
            assert err < 2e-3, (i, err)                                        # CFG doubles the eps error (s = 2): 2 x the 1e-3 bound
        # two steps in bf16 (the benchmarked mode): eager == eager, graph == eager
        model.stop_after = 3
        with oa.forced_compute_dtype(torch.bfloat16):
            e1 = model.sample(a.cuda(), c.cuda(), x0.cuda(), cond_scale=cs)
            e2 = model.sample(a.cuda(), c.cuda(), x0.cuda(), cond_scale=cs)
            model.use_hip_graph = True
            g1 = model.sample(a.cuda(), c.cuda(), x0.cuda(), cond_scale=cs)
        assert torch.equal(e1, e2) and torch.equal(e1, g1)
    finally:
        model.stop_after = None
        model.use_hip_graph = False
        model.sampling_timesteps = 35


def test_full_size_fresh_dora_adapters_leave_the_denoiser_unchanged(full):
    """Idempotence at full size (runs last: it wraps the shared model in place).  peft zero-inits lora_B and DoRA starts its
    magnitude at ||W||, so g = 1 and the adapted UNet must reproduce the base UNet up to the rounding of g; one adapted
    backward at B=8 must then give finite gradients on every adapter tensor and none on the frozen base."""
    import osufusion_amd as oa
    from osufusion_amd.modules import lora_layers as LL
    model, x, a, c, t, noise = full
    sl = slice(0, 8)
    xs, as_, cs, ts, ns = (v[sl].cuda() for v in (x, a, c, t, noise))
    with torch.no_grad(), oa.forced_compute_dtype(torch.float32):      # fp32 mode: in bf16 a 1e-7 change of g re-rolls the rounding
        base = model.unet(xs, as_, ts, cs).float()                      # noise of ~90 blocks (1.2e-2 rel-L2, the whole-UNet bf16 floor)
    LL.get_peft_model(model, LL.LoraConfig(r=16, lora_alpha=16, use_dora=True))
    n_train, n_all = LL.trainable_parameter_counts(model)
    assert 0 < n_train < 0.03 * n_all
    with torch.no_grad(), oa.forced_compute_dtype(torch.float32):
        adapted = model.unet(xs, as_, ts, cs).float()
    rel = ((adapted - base).norm() / base.norm()).item()
    report("full_size_fresh_dora_vs_base_fp32", rel_l2=rel)
    assert rel < 1e-3, rel                                          # attention is bf16 inside the fp32 mode too (floor ~3e-4)
    for p in model.parameters():                                        # drop what the previous test left in the flat buffer
        p.grad = None
    with oa.forced_compute_dtype(torch.bfloat16):
        model.loss_with(xs, as_, cs, ns, ts, cond_drop_prob=0.0).backward()
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
        else:
            assert p.grad is None or not p.grad.any(), n


def test_config5_shard_b64_dora_r16_gradients(full):
    """BASELINE config 5 at its per-GPU size (global batch 512 on 8 GPUs = B=64 per GPU; trainer_peft.py:236-244, r = 16): one DoRA
    train backward at B=64, L=4096 in the timed bf16 mode -- finite gradients on every adapter tensor, none on the frozen base, and the
    loss / adapter gradients equal the mean over the two B=32 halves (linearity of the mean-reduced loss in the batch; the halves take
    other tile / split plans than the full shard)."""
    import osufusion_amd as oa
    from osufusion_amd import functional as Fn
    from osufusion_amd.modules import lora_layers as LL
    from osufusion_amd.train import Trainer
    model = full[0]
    if not LL.lora_modules(model):                                      # (the previous test wraps the shared model; stand-alone runs do it here)
        LL.get_peft_model(model, LL.LoraConfig(r=16, lora_alpha=16, use_dora=True))
    with torch.no_grad():
        gl = torch.Generator().manual_seed(5)
        for m in LL.lora_modules(model):                                # peft zero-inits lora_B: give it life so dA is exercised too
            w = m.lora_B["default"].weight
            w.copy_((torch.randn(w.shape, generator=gl) * 0.02).to(w.device))
    Bs = 64
    g = torch.Generator().manual_seed(512)
    x = (torch.randn(Bs, 6, L, generator=g) * 0.5).clamp_(-1, 1).cuda()
    a = (torch.randn(Bs, 96, L, generator=g) * 3 - 10).cuda()
    c = (torch.rand(Bs, 5, generator=g) * 2 - 1).cuda()
    t = torch.randint(0, 1000, (Bs,), generator=g).cuda()
    noise = torch.randn(Bs, 6, L, generator=g).cuda()
    for p in model.parameters():
        p.grad = None
    try:
        trainer = Trainer(model, compute_dtype=torch.bfloat16)
        n_train, n_all = LL.trainable_parameter_counts(model)
        assert trainer.flat.numel >= n_train and n_train < 0.03 * n_all

        def run(sl):
            trainer.flat.zero_grad()
            with oa.forced_compute_dtype(torch.bfloat16):
                loss = model.loss_with(x[sl], a[sl], c[sl], noise[sl], t[sl], cond_drop_prob=0.0)
                loss.backward()
            return loss.detach().double(), trainer.flat.grad.detach().clone()

        loss_full, g_full = run(slice(0, Bs))
        assert torch.isfinite(loss_full) and torch.isfinite(g_full).all() and g_full.abs().max() > 0
        for n, p in model.named_parameters():
            if not p.requires_grad:
                assert p.grad is None or not p.grad.any(), n
        names = {id(q): n for n, q in model.named_parameters()}
        for q, o in zip(trainer.flat.params, trainer.flat.offsets):    # every adapter tensor received a gradient
            assert g_full[o:o + q.numel()].abs().max() > 0, names[id(q)]
        l0, g0 = run(slice(0, 32))
        l1, g1 = run(slice(32, 64))
        loss_mean, g_mean = (l0 + l1) / 2, (g0 + g1) / 2
        rel = ((g_full - g_mean).double().norm() / g_mean.double().norm()).item()
        report("config5_b64_dora16_shard_linearity", loss_rel=abs(loss_full - loss_mean) / abs(loss_mean), grad_rel_l2=rel,
               trainable=n_train, total=n_all)
        assert abs(loss_full - loss_mean) / abs(loss_mean) < 1e-3
        assert rel < 1e-2, rel
    finally:
        Fn.enable_direct_grads(False)
