"""Round 5 GPU checks: the 8-phase 256 x 256 GEMM loops against the kernels they replace."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from osufusion_amd import functional as Fn
from osufusion_amd import forced_compute_dtype, ops
from tests.test_hip_parity import B, DEV, relmax, report


@pytest.mark.parametrize("kind,k,L,Cin", [("same", 1, 520, 64),      # one K-tile: prologue only, no steady state
                                          ("same", 1, 200, 128),     # two K-tiles
                                          ("same", 1, 264, 192),     # odd number of K-tiles
                                          ("same", 3, 200, 64),      # taps switch on every K-tile (LDS table of source rows), zero rows at sample edges
                                          ("same", 3, 136, 256),
                                          ("down", 3, 96, 128),      # stride 2 + right reflect (row-map mode 1); its input gradient: mode 3, four taps
                                          ("up", 3, 56, 64),         # nearest x2 (mode 2); its input gradient: stride 2
                                          ("same", 15, 64, 64)])     # 15 taps (the table's upper range is 16)
def test_gemm_8phase_kernel_bit_exact(monkeypatch, kind, k, L, Cin):
    """gemm_nt_big8_kernel (8-phase schedule: counted vmcnt, raw barriers, staggered wave halves; the default 256 x 256 forward / input-gradient
    kernel of residual.py:70,115, unet.py:118-123,149-156 for K % 64 == 0) forced onto small ragged shapes -- M, N tails, every row-map mode,
    every epilogue option (bias, SiLU, pre-activation copy, residual x per-sample scale, GroupNorm sums; the dgrad's residual) -- against
    (a) the one-barrier-per-K-step 256 x 256 kernel (OSUF_GEMM_NO8P): same accumulation order, same epilogue code: bit-identical outputs;
    (b) the 128 x 128 kernel: bit-identical pre-activations (the epilogues differ in mul+add contraction by at most a bf16 ulp)."""
    Cout = 328
    x = torch.randn(B, L, Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, k, device=DEV) / (Cin * k) ** 0.5)
    bias = torch.randn(Cout, device=DEV)
    Lout = {"same": L, "down": L // 2, "up": 2 * L}[kind]
    res = torch.randn(B, Lout, Cout, device=DEV).to(torch.bfloat16)
    rscale = torch.rand(B, Cout, device=DEV)
    outs = {}
    for name, big, no8p in (("small", "0", None), ("plain", "1", "1"), ("p8", "1", None)):
        monkeypatch.setenv("OSUF_GEMM_BIG_MIN_TILES", big)
        if no8p:
            monkeypatch.setenv("OSUF_GEMM_NO8P", no8p)
        else:
            monkeypatch.delenv("OSUF_GEMM_NO8P", raising=False)
        stats = torch.zeros(B, 2, dtype=torch.float64, device=DEV)
        y, pre = Fn.conv_forward(x, w, bias, Fn.PackCache(), kind, None, act=1, residual=res, rscale=rscale, stats=stats, want_pre=True)
        dx = Fn.conv_dgrad(y, w, Fn.PackCache(), kind, L, residual=x)
        # the launches without residual / pre-activation copy: transposed accumulators + gemm_big_epilogue_tr (bias, SiLU, GroupNorm sums)
        stats2 = torch.zeros(B, 2, dtype=torch.float64, device=DEV)
        y2 = Fn.conv_forward(x, w, bias, Fn.PackCache(), kind, None, act=1, stats=stats2)
        y3 = Fn.conv_forward(x, w, None, Fn.PackCache(), kind, None)
        dx2 = Fn.conv_dgrad(y, w, Fn.PackCache(), kind, L)
        outs[name] = (y.float(), pre.float(), stats.clone(), dx.float(), y2.float(), y3.float(), dx2.float(), stats2.clone())
    monkeypatch.delenv("OSUF_GEMM_BIG_MIN_TILES")
    monkeypatch.delenv("OSUF_GEMM_NO8P", raising=False)
    for a, b_ in zip(outs["p8"][:2] + outs["p8"][3:7], outs["plain"][:2] + outs["plain"][3:7]):
        assert torch.equal(a, b_)
    assert torch.allclose(outs["p8"][2], outs["plain"][2], rtol=1e-6)          # LDS float atomics inside a tile: order only
    assert torch.allclose(outs["p8"][7], outs["plain"][7], rtol=1e-6)
    assert torch.equal(outs["p8"][5], outs["small"][5])                        # no activation: also the 128 x 128 kernel's output bit for bit
    assert torch.equal(outs["p8"][1], outs["small"][1])
    xq, wq = x.float(), w.to(torch.bfloat16).float()
    if kind == "same":
        ref = F.conv1d(xq.permute(0, 2, 1), wq, bias, padding=k // 2)
    elif kind == "down":
        ref = F.conv1d(F.pad(xq.permute(0, 2, 1), (0, 1), mode="reflect"), wq, bias, stride=2)
    else:
        ref = F.conv1d(F.interpolate(xq.permute(0, 2, 1), scale_factor=2.0, mode="nearest"), wq, bias, padding=1)
    e = relmax(outs["p8"][1].permute(0, 2, 1), ref)
    report(f"gemm_8phase/{kind}_k{k}_L{L}_C{Cin}", pre_vs_conv1d=e)
    assert e < 1e-2


def test_gemm_8phase_chip_filling_shapes_repeatable():
    """The same comparison on chip-filling shapes (every CU busy, DMA queues loaded -- where a fragment read overtaking its LDS-DMA would
    show), ten launches each: bit-identical to the one-barrier loop every time."""
    import os
    for M, N, K, taps, halo in ((131072, 256, 256, 1, False), (65536, 1152, 256, 1, False), (32768, 512, 1024, 1, False), (32768, 768, 768, 3, False),
                                (16384, 1024, 2048, 1, False),
                                # the shared-panel k = 3 kernel (gemm_nt_big8_halo3_kernel) against gemm_nt_big_halo3_kernel: odd / even K-step counts, one K-step
                                (65536, 256, 256, 3, True), (32768, 768, 768, 3, True), (32768, 512, 64, 3, True), (16384, 1024, 1024, 3, True), (32768, 328, 192, 3, True)):
        L = 4096 if not halo else 512
        x = torch.randn(M, K, device=DEV).bfloat16()
        w = (torch.randn(taps, N, K, device=DEV) * 0.05).bfloat16()
        kw = dict(taps=taps, lin=L, lout=L, stride=1, pad=taps // 2) if taps > 1 else {}
        os.environ["OSUF_GEMM_NO8P"] = "1"
        if not halo:
            os.environ["OSUF_GEMM_NOHALO"] = "1"
        try:
            ref = ops.gemm_nt(x, w, None, **kw)
            del os.environ["OSUF_GEMM_NO8P"]
            for _ in range(10):
                got = ops.gemm_nt(x, w, None, **kw)
                assert torch.equal(got, ref), (M, N, K, taps)
        finally:
            os.environ.pop("OSUF_GEMM_NO8P", None)
            os.environ.pop("OSUF_GEMM_NOHALO", None)


@pytest.mark.parametrize("Bn,N,H,G,qsplit", [(2, 1024, 1, 1, 0), (2, 1024, 1, 1, 2), (1, 2048, 1, 1, 0), (1, 2048, 1, 1, 2),      # one query head per call
                                             (2, 1024, 4, 4, 0),                                                                  # kv_heads == heads: four H = 1 launches
                                             (1, 512, 2, 1, 16)])                                                                 # minimum trip count: two pairs per part
def test_generated_attention_backward_one_head_paths(Bn, N, H, G, qsplit):
    """the hand-placed 512-key sweep (mqa_bwd_fused512a_kernel, the default for N % 512 == 0) has dedicated H == 1 state --
    `s_cmp_eq_u32 s61, 1`, one block step fewer for the request offsets, the first request advanced by the block-wrap delta -- that the H >= 2
    cases never execute; per call H == 1 is what attn_kv_heads == attn_heads issues (unet.py:132-135).  Bit-identity of dK / dV with the
    compiled 512-key sweep, dQ up to the order of its atomics, and both against autograd of the fp32 formula.  (tools/check_bwd512a_addresses.py
    replays the same paths address by address on the CPU.)"""
    D = 64
    r = H // G
    qkv = torch.randn(Bn, N, (H + 2 * G) * D, device=DEV).to(torch.bfloat16)
    do = torch.randn(Bn, N, H * D, device=DEV).to(torch.bfloat16)
    x = qkv.float().cpu().requires_grad_()
    outs = []
    for g in range(G):
        q = x[..., g * r * D:(g + 1) * r * D].view(Bn, N, r, D).permute(0, 2, 1, 3)
        k = x[..., (H + g) * D:(H + g + 1) * D][:, None]
        v = x[..., (H + G + g) * D:(H + G + g + 1) * D][:, None]
        s = (q @ k.transpose(-1, -2)) * D ** -0.5
        outs.append((s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(Bn, N, r * D))
    torch.cat(outs, -1).backward(do.float().cpu())
    g_ref = x.grad.to(DEV)
    o, lse = ops.mqa_fwd(qkv, Bn, N, H, D, torch.bfloat16, D ** -0.5, kv_heads=G)
    got = ops.mqa_bwd(qkv, o, do, lse, Bn, N, H, D, D ** -0.5, variant=ops.ATTN_FUSED512A, qsplit=qsplit, kv_heads=G)
    ref = ops.mqa_bwd(qkv, o, do, lse, Bn, N, H, D, D ** -0.5, variant=ops.ATTN_FUSED512, qsplit=qsplit, kv_heads=G)
    assert torch.isfinite(got).all()
    assert torch.equal(got[..., H * D:], ref[..., H * D:])
    e_dq = ((got[..., :H * D] - ref[..., :H * D]).norm() / ref[..., :H * D].norm()).item()
    e = ((got - g_ref).norm() / g_ref.norm()).item()
    report(f"attn_bwd512a_one_head/B{Bn}_N{N}_H{H}_G{G}_qs{qsplit}", dq_vs_compiled=e_dq, vs_autograd=e)
    assert e_dq < 1e-5 and e < 1e-2


@pytest.mark.parametrize("Bn,N,H,G", [(2, 512, 4, 1), (1, 2048, 16, 1), (2, 1024, 4, 2), (2, 1024, 1, 1), (2, 200, 2, 1)])
def test_attention_with_prescaled_queries(Bn, N, H, G):
    """The softmax scale folded into the queries' bf16 rounding (osuf_rope_cast_qs -> osuf_mqa_fwd_qs -> osuf_mqa_bwd_fused_qs; attention.py:87-99
    rounds q to bf16 and scales the scores, here q * scale * log2 e is rounded once and the scores ARE the exponents): forward and every gradient
    (of the un-rotated, un-scaled projections) against autograd of the fp32 formula on the SAME pre-RoPE inputs, next to the unscaled path; the
    generated 512-key loop's pre-scaled variant bit-identical (dK / dV) to the compiled sweep fed the same pre-scaled queries."""
    D = 64
    r = H // G
    scale = D ** -0.5
    raw = torch.randn(Bn, N, (H + 2 * G) * D, device=DEV).to(torch.bfloat16)          # projections before RoPE
    do = torch.randn(Bn, N, H * D, device=DEV).to(torch.bfloat16)
    cos, sin = Fn.rope_tables(N, D, 2 * N, DEV)
    # fp32 reference: RoPE (half-split), per-group attention, autograd
    x = raw.float().cpu().requires_grad_()
    c_, s_ = cos.cpu(), sin.cpu()

    def rot(t):                                                                       # (Bn, N, heads, D)
        t1, t2 = t[..., :32], t[..., 32:]
        return torch.cat([t1 * c_[None, :, None] - t2 * s_[None, :, None], t2 * c_[None, :, None] + t1 * s_[None, :, None]], -1)
    q = rot(x[..., : H * D].view(Bn, N, H, D))
    k = rot(x[..., H * D:(H + G) * D].view(Bn, N, G, D))
    v = x[..., (H + G) * D:].view(Bn, N, G, D)
    outs = []
    for g in range(G):
        sc = torch.einsum("bnhd,bmd->bhnm", q[:, :, g * r:(g + 1) * r], k[:, :, g]) * scale
        outs.append(torch.einsum("bhnm,bmd->bnhd", sc.softmax(-1), v[:, :, g]).reshape(Bn, N, r * D))
    o_ref = torch.cat(outs, -1)
    o_ref.backward(do.float().cpu())
    g_ref = x.grad.to(DEV)
    res = {}
    for name, qs in (("plain", False), ("qs", True)):
        qkv_r = ops.rope_cast(raw, cos, sin, N, H + G, H + 2 * G, D, q_mul=scale * ops.LOG2E if qs else 1.0, n_q_heads=H)
        o, lse = ops.mqa_fwd(qkv_r, Bn, N, H, D, torch.bfloat16, scale, kv_heads=G, qs=qs)
        dqkv = ops.mqa_bwd(qkv_r, o, do, lse, Bn, N, H, D, scale, torch.float32, cos, sin, variant=ops.ATTN_FUSED, kv_heads=G, qs=qs)
        res[name] = (o.float(), lse, dqkv)
        e_o = ((o.float().cpu() - o_ref.detach()).norm() / o_ref.detach().norm()).item()
        e_g = ((dqkv - g_ref).norm() / g_ref.norm()).item()
        parts = {nm: ((dqkv[..., sl] - g_ref[..., sl]).norm() / g_ref[..., sl].norm()).item()
                 for nm, sl in (("dq", slice(0, H * D)), ("dk", slice(H * D, (H + G) * D)), ("dv", slice((H + G) * D, None)))}
        report(f"attn_prescaled_q/B{Bn}_N{N}_H{H}_G{G}/{name}", out=e_o, grad=e_g, **parts)
        assert e_o < 5e-3 and e_g < 1e-2 and max(parts.values()) < 1e-2, (name, e_o, e_g, parts)
    # the two roundings of q differ by bf16 noise, no more; the log2-domain lse agrees to fp32 noise of the scores
    assert ((res["qs"][0] - res["plain"][0]).norm() / res["plain"][0].norm()).item() < 5e-3
    assert (res["qs"][1] - res["plain"][1]).abs().max().item() < 5e-2
    if N % 512 == 0:
        qkv_r = ops.rope_cast(raw, cos, sin, N, H + G, H + 2 * G, D, q_mul=scale * ops.LOG2E, n_q_heads=H)
        o, lse = ops.mqa_fwd(qkv_r, Bn, N, H, D, torch.bfloat16, scale, kv_heads=G, qs=True)
        a = ops.mqa_bwd(qkv_r, o, do, lse, Bn, N, H, D, scale, torch.float32, cos, sin, variant=ops.ATTN_FUSED512A, kv_heads=G, qs=True)
        b = ops.mqa_bwd(qkv_r, o, do, lse, Bn, N, H, D, scale, torch.float32, cos, sin, variant=ops.ATTN_FUSED512, kv_heads=G, qs=True)
        assert torch.equal(a[..., H * D:], b[..., H * D:])
        assert ((a[..., : H * D] - b[..., : H * D]).norm() / b[..., : H * D].norm()).item() < 1e-5


@pytest.mark.parametrize("Bn,N,H,qs", [(2, 200, 3, False), (1, 512, 2, True), (2, 1024, 4, True), (1, 2048, 16, True), (2, 1024, 1, False)])
def test_forward_zero_fills_the_backward_dq_accumulator(Bn, N, H, qs):
    """osuf_mqa_fwd_zdq + dq_mode | OSUF_DQ_PREZEROED: the forward kernel clears the fp32 dQ accumulator of the layer's fused backward sweep (the
    memset in front of the sweep is gone).  The forward's own results do not change by a bit; exactly the accumulator's B*N*H*64 floats are
    cleared (a NaN-filled workspace: zeros there, NaN behind); the backward on that workspace equals the memset path -- bit for bit where at most two
    key blocks add into an element (fp32 addition commutes), to rounding where more do."""
    D = 64
    scale = D ** -0.5
    raw = torch.randn(Bn, N, (H + 2) * D, device=DEV).to(torch.bfloat16)
    do = torch.randn(Bn, N, H * D, device=DEV).to(torch.bfloat16)
    cos, sin = Fn.rope_tables(N, D, 2 * N, DEV)
    qkv_r = ops.rope_cast(raw, cos, sin, N, H + 1, H + 2, D, q_mul=scale * ops.LOG2E if qs else 1.0, n_q_heads=H)
    o0, lse0 = ops.mqa_fwd(qkv_r, Bn, N, H, D, torch.bfloat16, scale, qs=qs)
    g0 = ops.mqa_bwd(qkv_r, o0, do, lse0, Bn, N, H, D, scale, torch.float32, cos, sin, variant=ops.ATTN_FUSED, qs=qs)
    ws = ops.fused_bwd_workspace(Bn, N, H, D, torch.float32, DEV, variant=ops.ATTN_FUSED)
    assert ws is not None and ws.numel() >= Bn * N * H * D
    ws.fill_(float("nan"))
    o1, lse1 = ops.mqa_fwd(qkv_r, Bn, N, H, D, torch.bfloat16, scale, qs=qs, zero_dq=ws)
    assert torch.equal(o0, o1) and torch.equal(lse0, lse1)
    n = Bn * N * H * D
    assert int((ws[:n] != 0).sum().item()) == 0
    assert ws.numel() == n or bool(torch.isnan(ws[n:]).all())
    if ws.numel() > n:
        ws[n:].zero_()                                     # (the dK / dV partial-sum region behind it is written whole by the sweep; keep NaN out of the test's way)
    g1 = ops.mqa_bwd(qkv_r, o1, do, lse1, Bn, N, H, D, scale, torch.float32, cos, sin, variant=ops.ATTN_FUSED, qs=qs, workspace=ws)
    assert torch.isfinite(g1).all()
    if N <= 1024:
        assert torch.equal(g0, g1)
    else:
        assert ((g0 - g1).norm() / g0.norm()).item() < 1e-5
    # a workspace made for another variant is not trusted: the shared workspace + memset path runs (same result)
    g2 = ops.mqa_bwd(qkv_r, o1, do, lse1, Bn, N, H, D, scale, torch.float32, cos, sin, variant=ops.ATTN_FUSED256, qs=qs, workspace=ws)
    g256 = ops.mqa_bwd(qkv_r, o1, do, lse1, Bn, N, H, D, scale, torch.float32, cos, sin, variant=ops.ATTN_FUSED256, qs=qs)
    assert ((g256 - g2).norm() / g256.norm()).item() < 1e-5 and ((g0 - g2).norm() / g0.norm()).item() < 1e-3


def test_attention_module_gradients_with_and_without_the_forward_zero_fill(monkeypatch):
    """AttentionFn end to end (LayerNorm -> q|kv -> RoPE -> attention -> to_out): every gradient with the dQ accumulator cleared by the forward kernel
    against OSUF_ATTN_NO_ZDQ=1 (the memset in the backward entry point), and a no-grad forward allocates no workspace."""
    from osufusion_amd.modules.unet import Attention
    torch.manual_seed(3)
    att = Attention(256, heads=4, dim_head=64, kv_heads=1, context_len=1024).to(DEV)
    x = torch.randn(2, 1024, 256, device=DEV)
    res = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("OSUF_ATTN_NO_ZDQ", "1")
        else:
            monkeypatch.delenv("OSUF_ATTN_NO_ZDQ", raising=False)
        for p in att.parameters():
            p.grad = None
        xi = x.clone().requires_grad_()
        with forced_compute_dtype(torch.bfloat16):
            y = att(xi)
            y.float().square().mean().backward()
        res[off] = [xi.grad.clone()] + [p.grad.clone() for p in att.parameters()]
    for a, b in zip(res[False], res[True]):
        assert torch.isfinite(a).all() and ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item() < 1e-5
    monkeypatch.delenv("OSUF_ATTN_NO_ZDQ", raising=False)
    calls = []
    real = ops.fused_bwd_workspace
    monkeypatch.setattr(ops, "fused_bwd_workspace", lambda *a, **k: calls.append(1) or real(*a, **k))
    with torch.no_grad(), forced_compute_dtype(torch.bfloat16):
        att(x)
    assert not calls


@pytest.mark.parametrize("Bn,N,H", [(2, 200, 3), (1, 512, 2), (2, 1024, 4), (1, 2048, 16)])
def test_forward_rotates_its_own_queries(Bn, N, H):
    """osuf_mqa_fwd_rope: the attention kernel rotates, scales and rounds its query tiles itself (attention.py:52-58,87-92: RoPE, then the bf16 cast in
    front of SDPA) and stores them for the backward; only the K | V head blocks take the stand-alone RoPE + cast pass.  Against osuf_rope_cast_qs +
    osuf_mqa_fwd_qs on the same raw projections: K / V columns bit-equal (same kernel), the stored queries equal to at most one bf16 step (two ulps of the smaller neighbour at a binade edge) (the two
    kernels' fp32 expressions may contract differently), output and log-sum-exp equal to bf16 / fp32 noise -- bit-equal when the queries are;
    without q_out (inference) the same output."""
    D = 64
    scale = D ** -0.5
    raw = torch.randn(Bn, N, (H + 2) * D, device=DEV).to(torch.bfloat16)
    cos, sin = Fn.rope_tables(N, D, 2 * N, DEV)
    ref = ops.rope_cast(raw, cos, sin, N, H + 1, H + 2, D, q_mul=scale * ops.LOG2E, n_q_heads=H)
    o0, lse0 = ops.mqa_fwd(ref, Bn, N, H, D, torch.bfloat16, scale, qs=True)
    assert ops.fwd_rope_ok(raw, D, 1, True)
    qkv_r, o1, lse1 = ops.mqa_fwd_rope(raw, cos, sin, Bn, N, H, D, torch.bfloat16, scale, write_q=True)
    assert torch.equal(qkv_r[..., H * D:], ref[..., H * D:])
    q0, q1 = ref[..., : H * D].float(), qkv_r[..., : H * D].float()
    same = (q0 == q1).float().mean().item()
    ulp = (q0.abs().clamp_min(1e-30).log2().floor() - 7).exp2()
    tiny = q0.abs().amax() * 2.0 ** -16                    # a cancelling x1 cos - x2 sin: the two contractions differ by fp32 noise of the PRODUCTS
    assert ((q0 - q1).abs() <= torch.maximum(2 * ulp, tiny)).all() and same > 0.99, same
    report(f"fwd_rope/B{Bn}_N{N}_H{H}", q_bit_equal_share=same)
    if same == 1.0:
        assert torch.equal(o0, o1) and torch.equal(lse0, lse1)
    else:
        assert ((o0.float() - o1.float()).norm() / o0.float().norm()).item() < 2e-3 and (lse0 - lse1).abs().max().item() < 2e-2
    _, o2, lse2 = ops.mqa_fwd_rope(raw, cos, sin, Bn, N, H, D, torch.bfloat16, scale, write_q=False)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)
    # with the dQ accumulator's zero fill riding along
    ws = ops.fused_bwd_workspace(Bn, N, H, D, torch.float32, DEV, variant=ops.ATTN_FUSED)
    ws.fill_(float("nan"))
    _, o3, lse3 = ops.mqa_fwd_rope(raw, cos, sin, Bn, N, H, D, torch.bfloat16, scale, write_q=True, zero_dq=ws)
    assert torch.equal(o1, o3) and int((ws[: Bn * N * H * D] != 0).sum().item()) == 0


def test_attention_module_with_and_without_the_rotating_forward(monkeypatch):
    """AttentionFn end to end, default path (the forward rotates its queries) against OSUF_ATTN_NO_FWD_ROPE=1 (rope_cast over all heads): output and every
    gradient to bf16 rounding noise; and the inference forward (no_grad: no stored queries) equals the training forward."""
    from osufusion_amd.modules.unet import Attention
    torch.manual_seed(5)
    att = Attention(256, heads=4, dim_head=64, kv_heads=1, context_len=2048).to(DEV)
    x = torch.randn(1, 2048, 256, device=DEV)                  # (N >= 2048: where the training forward takes the rotating kernel)
    res = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("OSUF_ATTN_NO_FWD_ROPE", "1")
        else:
            monkeypatch.delenv("OSUF_ATTN_NO_FWD_ROPE", raising=False)
        for p in att.parameters():
            p.grad = None
        xi = x.clone().requires_grad_()
        with forced_compute_dtype(torch.bfloat16):
            y = att(xi)
            y.float().square().mean().backward()
        res[off] = [y.detach().float(), xi.grad.clone()] + [p.grad.clone() for p in att.parameters()]
    for a, b in zip(res[False], res[True]):
        assert torch.isfinite(a).all() and ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item() < 2e-3
    monkeypatch.delenv("OSUF_ATTN_NO_FWD_ROPE", raising=False)
    with torch.no_grad(), forced_compute_dtype(torch.bfloat16):
        y_inf = att(x)
    assert torch.equal(y_inf.float(), res[False][0])


@pytest.mark.parametrize("Bn,L,C", [(3, 520, 96), (2, 1024, 256), (2, 512, 768), (1, 2048, 1024), (2, 64, 2048), (2, 4096, 256)])
def test_global_context_pooling_in_one_pass(Bn, L, C):
    """osuf_gca_pool (residual.py:29-31: softmax over the sequence of to_k(h), pooled = sum_n p[n] h[n]) against the three-kernel form it replaces
    (osuf_rowdot + osuf_softmax_rows + osuf_wcolsum) and against torch in fp64: probabilities and pooled vector to fp32 noise, identical bits on
    every call (no atomics), ragged L, every channel-chunk geometry (C = 96 ... 2048), logits with a large spread."""
    for dt in (torch.float32, torch.bfloat16):
        h = (torch.randn(Bn, L, C, device=DEV) * 2).to(dt)
        wk = torch.randn(C, device=DEV) * (6.0 / C ** 0.5)          # logits of a few units: a peaked softmax
        bk = torch.randn(1, device=DEV)
        pooled, p = ops.gca_pool(h.reshape(Bn * L, C), wk, bk, L)
        pooled2, p2 = ops.gca_pool(h.reshape(Bn * L, C), wk, bk, L)
        assert torch.equal(pooled, pooled2) and torch.equal(p, p2)
        logits = (h.double() @ wk.double()) + bk.double()
        pref = logits.softmax(-1)
        want = (pref.unsqueeze(-1) * h.double()).sum(1)
        assert relmax(p.view(Bn, L), pref.float()) < 1e-4 and relmax(pooled, want.float()) < 1e-4
        assert torch.allclose(p.view(Bn, L).sum(-1), torch.ones(Bn, device=DEV), atol=1e-5)
        with ops.reproducible_mode(True):
            p_old = ops.rowdot(h.reshape(Bn * L, C), wk, bk, L)
            ops.softmax_rows_(p_old, Bn, L)
            pooled_old = ops.wcolsum(h.reshape(Bn * L, C), None, p_old, Bn, L)
        assert relmax(p, p_old) < 1e-4 and relmax(pooled, pooled_old) < 1e-4
