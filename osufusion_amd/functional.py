"""autograd.Functions over the HIP kernels: one Function per reference block, backward hand-scheduled.

Tensors between Functions are channels-last "rows" (B, L, C) in the compute dtype (bf16 or fp32).  Parameters
stay fp32 masters; GEMM operands are packed per (layout, dtype) and cached until the weights change.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops

_WEIGHT_EPOCH = 0          # bumped by the fused optimizer (it writes parameters through raw pointers)


def bump_weight_epoch() -> None:
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


class PackCache:
    """Per-module cache of GEMM-ready weight layouts, invalidated when any source parameter changes."""

    def __init__(self) -> None:
        self._d: Dict[Tuple, Tuple[Tuple, torch.Tensor]] = {}

    def get(self, key: Tuple, params: Tuple[torch.Tensor, ...], builder) -> torch.Tensor:
        ver = (_WEIGHT_EPOCH, *[(p._version, p.data_ptr()) for p in params])
        hit = self._d.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            t = builder().contiguous()
        self._d[key] = (ver, t)
        return t

    def packs(self, key: Tuple, params: Tuple[torch.Tensor, ...], weights, kind: str, dt: torch.dtype):
        """(fwd [k][O][I], dgrad [k'][I][O]) operands of a weight -- or of several weights stacked along O (the fused q|kv
        projection) -- built from the fp32 masters by one osuf_pack_weight launch per weight and cached like get()."""
        ver = (_WEIGHT_EPOCH, *[(p._version, p.data_ptr()) for p in params])
        hit = self._d.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        ws = weights if isinstance(weights, (tuple, list)) else (weights,)
        with torch.no_grad():
            ws = [w.detach().float().contiguous() for w in ws]
            if len(ws) == 1:
                pair = ops.pack_weight(ws[0], dt, kind)
            else:
                I = ws[0].shape[1]
                total = sum(w.shape[0] for w in ws)
                fwd = torch.empty((1, total, I), dtype=dt, device=ws[0].device)
                dgr = torch.empty((1, I, total), dtype=dt, device=ws[0].device)
                off = 0
                for w in ws:
                    ops.pack_weight(w, dt, kind, fwd=fwd, dgrad=dgr, row_offset=off)
                    off += w.shape[0]
                pair = (fwd, dgr)
        self._d[key] = (ver, pair)
        return pair


# ---- direct gradient accumulation ---------------------------------------------------------------------------
# With a Trainer, every parameter owns a persistent fp32 .grad view into one flat buffer.  The backward kernels then add
# straight into it (wgrad in torch's weight layout, bias / norm column sums by atomics) and return None to autograd, instead
# of returning ~1.2 k fresh tensors that autograd adds one small kernel at a time (measured: 1,124 adds + ~700 fills = 11.6 ms
# of a 300 ms step).  The reducer is told explicitly when a parameter's gradient is complete.
_DIRECT = False
_GRAD_DONE_CB = None


def enable_direct_grads(flag: bool, done_callback=None) -> None:
    global _DIRECT, _GRAD_DONE_CB
    _DIRECT, _GRAD_DONE_CB = bool(flag), done_callback


def grad_target(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if not _DIRECT or p is None or not p.is_leaf or not p.requires_grad:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or g.shape != p.shape or not g.is_contiguous():
        return None
    return g


def grad_done(p: torch.Tensor) -> None:
    if _GRAD_DONE_CB is not None:
        _GRAD_DONE_CB(p)


def _bias_grad(dy: torch.Tensor, bias: Optional[torch.Tensor], n: Optional[int] = None):
    """Column sums of dy into bias.grad (direct) or a fresh tensor (returned)."""
    tgt = grad_target(bias)
    if tgt is not None:
        ops.colsum(dy, n, out=tgt)
        grad_done(bias)
        return None
    return ops.colsum(dy, n)


def _rc(t: torch.Tensor) -> torch.Tensor:
    """Make a grad tensor kernel-consumable (last dim contiguous, dense rows, 16-B aligned)."""
    if t.stride(-1) != 1 or (t.dim() == 3 and t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)) or t.data_ptr() % 16:
        return t.contiguous()
    if t.dim() == 3 and t.stride(1) % 8:
        return t.contiguous()
    return t


_CONV_KINDS = {
    # kind: fwd (stride, mode), dgrad (taps, stride, pad, mode)
    "same": dict(stride=1, mode=0),
    "down": dict(stride=2, mode=1),
    "up": dict(stride=1, mode=2),
}


def _conv_geom(kind: str, k: int, Lin: int) -> Tuple[int, int, int, int]:
    """-> (Lout, stride, pad, mode) of the forward row map."""
    if kind == "same":
        return Lin, 1, k // 2, 0
    if kind == "down":
        if Lin % 2:
            raise ValueError("Downsample needs an even length (UNet pads to a multiple of 2^depth)")
        return Lin // 2, 2, 0, 1
    if kind == "up":
        return 2 * Lin, 1, 1, 2
    raise ValueError(kind)


def _vkey(vp, w):
    """vp = (tag, *source_params) for weights derived on the fly (merged stems, Parallel, padded heads)."""
    return (vp[0], tuple(vp[1:])) if vp is not None else ("", (w,))


def conv_forward(x, w, bias, cache: PackCache, kind: str, vp=None, **epi):
    B, Lin, Cin = x.shape
    k = w.shape[2] if w.dim() == 3 else 1
    Lout, stride, pad, mode = _conv_geom(kind, k, Lin)
    tag, vparams = _vkey(vp, w)
    wp = cache.packs(("p", kind, x.dtype, tag), vparams, w, kind, x.dtype)[0]
    return ops.gemm_nt(x, wp, bias, taps=k, lin=Lin, lout=Lout, stride=stride, pad=pad, mode=mode, out_shape=(B, Lout, wp.shape[1]), **epi)


def conv_dgrad(dy, w, cache: PackCache, kind: str, Lin: int, residual=None, vp=None):
    """Input gradient of conv_forward: dy rows (B, Lout, Cout) -> (B, Lin, Cin)."""
    B, Lout, Cout = dy.shape
    k = w.shape[2] if w.dim() == 3 else 1
    tag, vparams = _vkey(vp, w)
    wp = cache.packs(("p", kind, dy.dtype, tag), vparams, w, kind, dy.dtype)[1]
    Cin = wp.shape[1]
    if kind == "same":
        geom = dict(taps=k, stride=1, pad=k - 1 - k // 2, mode=0)
    elif kind == "down":
        geom = dict(taps=4, stride=1, pad=0, mode=3)
    else:
        geom = dict(taps=4, stride=2, pad=1, mode=0)
    return ops.gemm_nt(dy, wp, None, lin=Lout, lout=Lin, residual=residual, out_shape=(B, Lin, Cin), **geom)


def conv_wgrad(dy, x, w, kind: str, direct: bool = True):
    """Weight gradient in w's own layout: added straight into w.grad (returns None) when direct accumulation is on and w is a
    leaf parameter with a dense fp32 .grad; otherwise a fresh tensor shaped like w."""
    B, Lin, Cin = x.shape
    conv = w.dim() == 3
    k = w.shape[2] if conv else 1
    Lout, stride, pad, mode = _conv_geom(kind, k, Lin)
    tgt = grad_target(w) if direct else None
    g = ops.gemm_tn(dy, x, taps=k, lin=Lin, lout=Lout, stride=stride, pad=pad, mode=mode, n1=w.shape[0], out=tgt, conv_layout=conv,
                    accumulate=tgt is not None)
    if tgt is not None:
        grad_done(w)
        return None
    return g if conv else g[0]


# ---------------------------------------------------------------------------------------------------------
class ConvFn(torch.autograd.Function):
    """Conv1d (same / Downsample / Upsample geometry) or Linear on rows.  unet.py:61-101, residual.py:115."""

    @staticmethod
    def forward(ctx, x, w, bias, cache, kind, vp=None):
        ctx.save_for_backward(x, w)
        ctx.cache, ctx.kind, ctx.has_bias, ctx.vp = cache, kind, bias is not None, vp
        ctx.bias_ref = bias                               # the Parameter itself (for direct .grad accumulation)
        return conv_forward(x, w, bias, cache, kind, vp)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _rc(dy)
        dx = conv_dgrad(dy, w, ctx.cache, ctx.kind, x.shape[1], vp=ctx.vp) if ctx.needs_input_grad[0] else None
        dw = conv_wgrad(dy, x, w, ctx.kind) if ctx.needs_input_grad[1] else None
        db = _bias_grad(dy, ctx.bias_ref, w.shape[0]) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None, None


class BlockFn(torch.autograd.Function):
    """conv k3 -> GroupNorm(1,C) -> FiLM -> SiLU   (residual.py:75-84).  ss: fp32 (B, 2C) = (scale | shift) or None."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, ss, cache):
        B, L, _ = x.shape
        C = w.shape[0]
        stats = torch.zeros((B, 2), dtype=torch.float64, device=x.device)
        y = conv_forward(x, w, bias, cache, "same", None, stats=stats)
        mr = ops.gn_finalize(stats, L * C)
        ssc = ss.contiguous() if ss is not None else None
        h = ops.gn_apply(y, mr, gamma, beta, ssc, L)
        ctx.save_for_backward(x, w, y, mr, gamma, beta, ssc if ssc is not None else mr)
        ctx.cache, ctx.has_ss = cache, ss is not None
        ctx.bias_ref = bias
        return h

    @staticmethod
    def backward(ctx, dh):
        x, w, y, mr, gamma, beta, ss = ctx.saved_tensors
        ss = ss if ctx.has_ss else None
        L = x.shape[1]
        tg, tb = grad_target(gamma), grad_target(beta)
        direct_norm = tg is not None and tb is not None
        dy, dgamma, dbeta, dss = ops.gn_bwd(_rc(dh), y, mr, gamma, beta, ss, L, tg if direct_norm else None, tb if direct_norm else None)
        if direct_norm:
            grad_done(gamma); grad_done(beta)
            dgamma = dbeta = None
        dx = conv_dgrad(dy, w, ctx.cache, "same", L) if ctx.needs_input_grad[0] else None
        dw = conv_wgrad(dy, x, w, "same")
        db = _bias_grad(dy, ctx.bias_ref)
        return dx, dw, db, dgamma, dbeta, dss, None


class GCAPoolFn(torch.autograd.Function):
    """GlobalContext pooling: pooled[b,c] = sum_n softmax_n(h.wk + bk)[n] * h[b,n,c]   (residual.py:29-31) -> fp32 (B,C)."""

    @staticmethod
    def forward(ctx, h, wk, bk):
        B, L, C = h.shape
        wkv = wk.reshape(-1).contiguous()
        p = ops.rowdot(h, wkv, bk.reshape(-1), L)
        ops.softmax_rows_(p, B, L)
        pooled = ops.wcolsum(h, None, p, B, L)
        ctx.save_for_backward(h, wkv, p, pooled)
        ctx.wshape = wk.shape
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        h, wkv, p, pooled = ctx.saved_tensors
        B, L, C = h.shape
        dpooled = dpooled.contiguous().float()
        sdot = (pooled * dpooled).sum(1).contiguous()
        zero_gate = torch.zeros((B, C), dtype=torch.float32, device=h.device)
        # dh = p*dpooled + dlogit*wk  (the `dout*gate` term of the fused kernel is disabled with a zero gate)
        dh, dlogit = ops.gca_bwd_apply(h, h, p, zero_gate, dpooled, sdot, wkv, L)
        dwk = ops.wcolsum(h, None, dlogit, B, L).sum(0)
        dbk = dlogit.sum().reshape(1)
        return dh, dwk.reshape(ctx.wshape), dbk


class GateResFn(torch.autograd.Function):
    """out = h * gate + res   (residual.py:135-137, identity res_conv).  gate fp32 (B, C)."""

    @staticmethod
    def forward(ctx, h, gate, res):
        L = h.shape[1]
        gate = gate.contiguous()
        ctx.save_for_backward(h, gate)
        return ops.gate_residual(h, gate, res, L)

    @staticmethod
    def backward(ctx, dout):
        h, gate = ctx.saved_tensors
        B, L, C = h.shape
        dout = _rc(dout)
        dgate = ops.wcolsum(dout, h, None, B, L)
        dh = ops.gate_residual(dout, gate, None, L)                 # dout * gate
        return dh, dgate, dout


class GateResConvFn(torch.autograd.Function):
    """out = h * gate + res_conv(x)  (residual.py:135-137, 1x1 res_conv): the gate-multiply rides the GEMM epilogue."""

    @staticmethod
    def forward(ctx, h, gate, x, w, bias, cache):
        gate = gate.contiguous()
        ctx.save_for_backward(h, gate, x, w)
        ctx.cache, ctx.bias_ref = cache, bias
        return conv_forward(x, w, bias, cache, "same", None, residual=h, rscale=gate)

    @staticmethod
    def backward(ctx, dout):
        h, gate, x, w = ctx.saved_tensors
        B, L, C = h.shape
        dout = _rc(dout)
        dgate = ops.wcolsum(dout, h, None, B, L)
        dh = ops.gate_residual(dout, gate, None, L)
        dx = conv_dgrad(dout, w, ctx.cache, "same", L)
        dw = conv_wgrad(dout, x, w, "same")
        db = _bias_grad(dout, ctx.bias_ref)
        return dh, dgate, dx, dw, db, None


class FeedForwardFn(torch.autograd.Function):
    """x + W2 silu(W1 x + b1) + b2   (unet.py:149-156,182): SiLU and the residual ride the GEMM epilogues."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, cache):
        h, pre = conv_forward(x, w1, b1, cache, "same", None, act=1, want_pre=True)
        wp2 = cache.packs(("p2", x.dtype), (w2,), w2, "same", x.dtype)[0]
        out = ops.gemm_nt(h, wp2, b2, residual=x, out_shape=x.shape)
        ctx.save_for_backward(x, w1, w2, h, pre)
        ctx.cache, ctx.b1, ctx.b2 = cache, b1, b2
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, w2, h, pre = ctx.saved_tensors
        dout = _rc(dout)
        cache = ctx.cache
        wd2 = cache.packs(("p2", x.dtype), (w2,), w2, "same", x.dtype)[1]
        dpre = ops.gemm_nt(dout, wd2, None, dact=pre, out_shape=pre.shape)          # (dout W2) * silu'(pre)
        dw2 = conv_wgrad(dout, h, w2, "same")
        db2 = _bias_grad(dout, ctx.b2)
        dw1 = conv_wgrad(dpre, x, w1, "same")
        db1 = _bias_grad(dpre, ctx.b1)
        wd1 = cache.packs(("p", "same", x.dtype, ""), (w1,), w1, "same", x.dtype)[1]
        dx = ops.gemm_nt(dpre, wd1, None, residual=dout, out_shape=x.shape)
        return dx, dw1, db1, dw2, db2, None


_ROPE_TABLES: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def rope_tables(n: int, dim: int, scale_base: int, device, theta: float = 10000.0):
    """cos/sin tables [n][dim/2] in fp32, built exactly as attention.py:24-47 does for an fp32 q (on the host)."""
    key = (n, dim, scale_base, str(device))
    hit = _ROPE_TABLES.get(key)
    if hit is None:
        inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim))
        t = torch.arange(n, dtype=torch.float32)
        t *= scale_base / n
        freqs = torch.einsum("i , j -> i j", t, inv_freq)
        hit = (freqs.cos().contiguous().to(device), freqs.sin().contiguous().to(device))
        _ROPE_TABLES[key] = hit
    return hit


class AttentionFn(torch.autograd.Function):
    """LayerNorm -> to_q/to_kv -> RoPE -> MQA flash attention -> to_out + residual(normed x)   (unet.py:125-141)."""

    @staticmethod
    def forward(ctx, x, nw, nb, wq, wkv, wo, bo, cache, heads, dim_head, scale_base):
        B, N, C = x.shape
        H, D = heads, dim_head
        dt = x.dtype
        xn, mr = ops.ln_fwd(x, nw, nb)
        wqkv = cache.packs(("qkv", dt), (wq, wkv), (wq, wkv), "same", dt)[0]
        qkv = ops.gemm_nt(xn, wqkv, None, out_shape=(B, N, (H + 2) * D))
        cos, sin = rope_tables(N, D, scale_base, x.device)
        qkv_r = ops.rope_cast(qkv, cos, sin, N, H + 1, H + 2, D)                # rotate q heads and k; cast v
        del qkv
        scale = D ** -0.5
        o, lse = ops.mqa_fwd(qkv_r, B, N, H, D, dt, scale)
        wpo = cache.packs(("po", dt), (wo,), wo, "same", dt)[0]
        out = ops.gemm_nt(o, wpo, bo, residual=xn, out_shape=x.shape)
        ctx.save_for_backward(x, nw, mr, xn, wq, wkv, wo, qkv_r, o, lse)
        ctx.cache, ctx.geom = cache, (H, D, scale_base, scale)
        ctx.nb, ctx.bo = nb, bo
        return out

    @staticmethod
    def backward(ctx, dout):
        x, nw, mr, xn, wq, wkv, wo, qkv_r, o, lse = ctx.saved_tensors
        H, D, scale_base, scale = ctx.geom
        cache = ctx.cache
        B, N, C = x.shape
        dt = x.dtype
        dout = _rc(dout)
        # to_out
        dwo = conv_wgrad(dout, o, wo, "same")
        dbo = _bias_grad(dout, ctx.bo)
        wdo = cache.packs(("po", dt), (wo,), wo, "same", dt)[1]
        do = ops.gemm_nt(dout, wdo, None, out_shape=o.shape)
        do16 = ops.cast_rows(do, torch.bfloat16)                                 # SDPA backward runs in bf16 (attention.py:101)
        # attention + rope
        dqkv32 = ops.mqa_bwd(qkv_r, o, do16, lse, B, N, H, D, scale)
        cos, sin = rope_tables(N, D, scale_base, x.device)
        dqkv = ops.rope_bwd(dqkv32, dt, cos, sin, N, H + 1, H + 2, D)
        del dqkv32
        # to_q / to_kv
        dwq = conv_wgrad(dqkv[..., : H * D], xn, wq, "same")
        dwkv = conv_wgrad(dqkv[..., H * D:], xn, wkv, "same")
        wdqkv = cache.packs(("qkv", dt), (wq, wkv), (wq, wkv), "same", dt)[1]
        dxn = ops.gemm_nt(dqkv, wdqkv, None, residual=dout, out_shape=x.shape)   # + residual path (x + to_out(..), x = normed)
        tg, tb = grad_target(nw), grad_target(ctx.nb)
        direct_norm = tg is not None and tb is not None
        dx, dnw, dnb = ops.ln_bwd(dxn, x, mr, nw, tg if direct_norm else None, tb if direct_norm else None)
        if direct_norm:
            grad_done(nw); grad_done(ctx.nb)
            dnw = dnb = None
        return dx, dnw, dnb, dwq, dwkv, dwo, dbo, None, None, None, None


class RowsFromNCLFn(torch.autograd.Function):
    """(B, C, L) fp32 -> rows (B, L, width) in the compute dtype [im2col over kt taps]; model-boundary layout change."""

    @staticmethod
    def forward(ctx, x, dtype, width, kt):
        ctx.shape, ctx.kt = x.shape, kt
        return ops.ncl_to_rows(x.contiguous().float(), dtype, width, kt)

    @staticmethod
    def backward(ctx, d):
        if ctx.kt != 1:
            raise NotImplementedError("gradient w.r.t. the raw x input of the im2col stem is not provided")
        return ops.rows_to_ncl(_rc(d), ctx.shape[1]), None, None, None


class NCLFromRowsFn(torch.autograd.Function):
    """rows (B, L, >=C) -> (B, C, L) fp32."""

    @staticmethod
    def forward(ctx, rows, C):
        ctx.meta = (rows.dtype, rows.shape[2])
        return ops.rows_to_ncl(rows, C)

    @staticmethod
    def backward(ctx, d):
        dtype, width = ctx.meta
        return ops.ncl_to_rows(d.contiguous().float(), dtype, width, 1), None
