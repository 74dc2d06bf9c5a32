"""autograd.Functions over the HIP kernels: one Function per reference block, backward hand-scheduled.

Tensors between Functions are channels-last "rows" (B, L, C) in the compute dtype (bf16 or fp32).  Parameters
stay fp32 masters; GEMM operands are packed per (layout, dtype) and cached until the weights change.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Tuple

import os

import torch

from . import ops
from .tracing import scope

_WEIGHT_EPOCH = 0          # bumped by the fused optimizer (it writes parameters through raw pointers)


def bump_weight_epoch() -> None:
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


class PackCache:
    """Per-module cache of GEMM-ready weight layouts, invalidated when any source parameter changes."""

    def __init__(self) -> None:
        self._d: Dict[Tuple, Tuple] = {}          # key -> (version, value, grouped-refresh job or None)

    def get(self, key: Tuple, params: Tuple[torch.Tensor, ...], builder) -> torch.Tensor:
        ver = (_WEIGHT_EPOCH, *[(p._version, p.data_ptr()) for p in params])
        hit = self._d.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            t = builder().contiguous()
        self._d[key] = (ver, t, None)
        return t

    def packs(self, key: Tuple, params: Tuple[torch.Tensor, ...], weights, kind: str, dt: torch.dtype):
        """(fwd [k][O][I], dgrad [k'][I][O]) operands of a weight -- or of several weights stacked along O (the fused q|kv
        projection) -- built from the fp32 masters by one osuf_pack_weight launch per weight and cached like get().  Packs whose
        sources are plain fp32 leaf parameters also register with the grouped refresh (refresh_packs)."""
        ver = (_WEIGHT_EPOCH, *[(p._version, p.data_ptr()) for p in params])
        hit = self._d.get(key)
        if hit is not None and hit[0] == ver:
            if hit[2] is not None:
                hit[2].used = True
            return hit[1]
        ws = weights if isinstance(weights, (tuple, list)) else (weights,)
        with torch.no_grad():
            if len(ws) == 1:
                pair = _pack_one(ws[0], dt, kind)
            else:
                I = ws[0].shape[1]
                total = sum(w.shape[0] for w in ws)
                dev = ws[0].device
                fwd = torch.empty((1, total, I), dtype=dt, device=dev)
                dgr = torch.empty((1, I, total), dtype=dt, device=dev)
                off = 0
                for w in ws:
                    _pack_one(w, dt, kind, fwd=fwd, dgrad=dgr, row_offset=off)
                    off += w.shape[0]
                pair = (fwd, dgr)
        job = None
        if all(_plain_master(w, params) for w in ws):
            job = _PackJob(self, key, tuple(params), tuple(ws), kind, dt, pair)
            _PACK_JOBS[(id(self), key)] = job
        self._d[key] = (ver, pair, job)
        return pair


def _plain_master(w, params=()) -> bool:
    """A weight the grouped refresh can re-read in place: an fp32, contiguous nn.Parameter that IS one of the cache entry's source
    parameters (not an adapter's effective weight, not a tensor derived from parameters such as a merged stem, `Parallel`'s
    c3 + pad(c1), the padded `final_conv` or a GQA-permuted view).  `is_leaf` must not decide this: under no_grad / inference_mode
    (reentrant activation checkpointing, `sample()` between two optimizer steps) every derived tensor is a leaf too, and a job
    registered on such a temporary would re-pack from its stale copy after the next optimizer step."""
    return (isinstance(w, torch.nn.Parameter) and any(w is p for p in params) and w.dtype == torch.float32
            and w.is_contiguous() and w.is_cuda)


class _PackJob:
    """One cached pack of plain master weights: enough to re-run it from a descriptor table and re-validate its cache entry."""
    __slots__ = ("cache", "key", "params", "ws", "kind", "dt", "pair", "used")

    def __init__(self, cache, key, params, ws, kind, dt, pair) -> None:
        self.cache, self.key, self.params, self.ws, self.kind, self.dt, self.pair = weakref.ref(cache), key, params, ws, kind, dt, pair
        self.used = True


_PACK_JOBS: Dict[Tuple, _PackJob] = {}
_PACK_TABLES: Dict[torch.dtype, Tuple] = {}          # dtype -> (signature, device table, n, total blocks)


def refresh_packs() -> int:
    """Re-pack, in ONE launch per compute dtype, every registered weight pack that was used since the last refresh, and mark those
    cache entries valid for the current parameters.  Called by the Trainer right after the optimizer step (all ~320 packed
    operands go stale together there; one osuf_pack_weight launch each was 2 ms per step).  Packs it does not cover (adapter
    weights, merged stems) and packs not used last step are rebuilt lazily by PackCache as before.  -> number of packs refreshed."""
    by_dt: Dict[torch.dtype, list] = {}
    for k, job in list(_PACK_JOBS.items()):
        cache = job.cache()
        ent = cache._d.get(job.key) if cache is not None else None
        if ent is None or ent[2] is not job:                      # module gone, or the entry was rebuilt under another job
            del _PACK_JOBS[k]
            continue
        if job.used:
            by_dt.setdefault(job.dt, []).append(job)
    done = 0
    with torch.no_grad():
        for dt, jobs in by_dt.items():
            sig = tuple((id(j), tuple(w.data_ptr() for w in j.ws)) for j in jobs)
            tab = _PACK_TABLES.get(dt)
            if tab is None or tab[0] != sig:
                items = []
                for j in jobs:
                    off = 0
                    for w in j.ws:
                        items.append((w.detach(), j.kind, j.pair[0], j.pair[1], off))
                        off += w.shape[0]
                dev, n, blocks = ops.pack_desc_table(items, dt, jobs[0].ws[0].device)
                tab = (sig, dev, n, blocks)
                _PACK_TABLES[dt] = tab
            ops.pack_weight_group(tab[1], tab[2], tab[3], dt)
            for j in jobs:
                ver = (_WEIGHT_EPOCH, *[(p._version, p.data_ptr()) for p in j.params])
                j.cache()._d[j.key] = (ver, j.pair, j)
                j.used = False
            done += len(jobs)
    return done


def _pack_one(w, dt, kind, **kw):
    """osuf_pack_weight of a master weight, or of an adapter's effective weight straight from its factors."""
    if isinstance(w, EffWeight):
        ad = w.ad
        return ops.pack_weight(ad.w.detach().float().contiguous(), dt, kind, adapt=(ad.a.detach().float().contiguous(),
                               ad.b.detach().float().contiguous(), w.g if ad.m is not None else None, ad.scaling), **kw)
    return ops.pack_weight(w.detach().float().contiguous(), dt, kind, **kw)


class EffWeight:
    """Stand-in for an Adapter's effective weight g*(W + s*BA): it has the base weight's shape and is only ever materialised
    inside the pack kernel (osuf_pack_weight_adapted), in the GEMM operand layouts."""

    def __init__(self, ad, g) -> None:
        self.ad, self.g, self.shape, self.device = ad, g, ad.w.shape, ad.w.device

    def dim(self) -> int:
        return len(self.shape)


# ---- direct gradient accumulation ---------------------------------------------------------------------------
# With a Trainer, every parameter owns a persistent fp32 .grad view into one flat buffer.  The backward kernels then add
# straight into it (wgrad in torch's weight layout, bias / norm column sums by atomics) and return None to autograd, instead
# of returning ~1.2 k fresh tensors that autograd adds one small kernel at a time (measured: 1,124 adds + ~700 fills = 11.6 ms
# of a 300 ms step).  The reducer is told explicitly when a parameter's gradient is complete.
_DIRECT = False
_GRAD_DONE_CB = None
_GRAD_COMPLETE_CB = None
_PENDING_USES: Dict[int, int] = {}     # id(param) -> forward uses whose backward hook has not run yet (see grad_use / grad_used)


def enable_direct_grads(flag: bool, done_callback=None, complete_callback=None) -> None:
    """complete_callback(p): the gradient of p is COMPLETE for this backward.  Only for parameters that are deliberately kept out of the
    autograd graph (no AccumulateGrad node, hence no post-accumulate hook): the grouped FiLM projections, whose weight gradients are
    written by tensor hooks (film_group)."""
    global _DIRECT, _GRAD_DONE_CB, _GRAD_COMPLETE_CB
    _DIRECT, _GRAD_DONE_CB, _GRAD_COMPLETE_CB = bool(flag), done_callback, complete_callback
    _PENDING_USES.clear()


def grad_use(p: torch.Tensor) -> None:
    """A forward used p outside the autograd graph; its gradient is complete when every such use has reported grad_used."""
    _PENDING_USES[id(p)] = _PENDING_USES.get(id(p), 0) + 1


def grad_used(p: torch.Tensor) -> None:
    left = _PENDING_USES.get(id(p), 1) - 1
    if left > 0:
        _PENDING_USES[id(p)] = left
        return
    _PENDING_USES.pop(id(p), None)
    if _GRAD_DONE_CB is not None:
        _GRAD_DONE_CB(p)
    if _GRAD_COMPLETE_CB is not None:
        _GRAD_COMPLETE_CB(p)


def reset_grad_uses() -> None:
    """End of a backward (GradReducer.finish): uses whose hook never ran (an output nobody differentiated) must not linger."""
    _PENDING_USES.clear()


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


FUSE_BIAS_GRAD = os.environ.get("OSUF_NO_FUSED_BIAS") is None      # weight + bias gradient of a layer from one pass over dy (osuf_gemm_tn_bias); off: osuf_gemm_tn + osuf_colsum


def conv_wgrad_bias(dy, x, w, kind: str, bias, need_w: bool, need_b: bool, n: Optional[int] = None):
    """(dw, db) of one Conv1d / Linear.  When both are wanted and the bias has a direct .grad target, its column sums ride the weight
    gradient's pass over dy; otherwise the two separate calls."""
    if need_w and need_b and bias is not None and FUSE_BIAS_GRAD:
        tgt = grad_target(bias)
        if tgt is not None and tgt.numel() == w.shape[0]:
            dw = conv_wgrad(dy, x, w, kind, bias_out=tgt)
            grad_done(bias)
            return dw, None
    dw = conv_wgrad(dy, x, w, kind) if need_w else None
    db = _bias_grad(dy, bias, n) if (need_b and bias is not None) else None
    return dw, db


def conv_wgrad(dy, x, w, kind: str, direct: bool = True, bias_out: Optional[torch.Tensor] = None):
    """Weight gradient in w's own layout: added straight into w.grad (returns None) when direct accumulation is on and w is a
    leaf parameter with a dense fp32 .grad; otherwise a fresh tensor shaped like w.  bias_out: see ops.gemm_tn."""
    B, Lin, Cin = x.shape
    conv = w.dim() == 3
    k = w.shape[2] if conv else 1
    Lout, stride, pad, mode = _conv_geom(kind, k, Lin)
    tgt = grad_target(w) if direct else None
    g = ops.gemm_tn(dy, x, taps=k, lin=Lin, lout=Lout, stride=stride, pad=pad, mode=mode, n1=w.shape[0], out=tgt, conv_layout=conv,
                    accumulate=tgt is not None, bias_out=bias_out)
    if tgt is not None:
        grad_done(w)
        return None
    return g if conv else g[0]


# ---- LoRA / DoRA adapters (modules/lora_layers.py; §8f row 2) ------------------------------------------------
class Adapter:
    """Frozen base weight + (lora_A, lora_B[, DoRA magnitude]) of one adapted Conv1d / Linear.

    The reference evaluates  base(x) + (g-1)*conv(x,W) + g*s*B(A(x)),  g = m/||W + s*BA|| (norm detached)
    (lora_layers.py:72-92) -- three convolutions per layer.  With the norm a constant of the step this is exactly
    conv(x, Weff) + bias with Weff = g*(W + s*BA), so here the forward and the input gradient run the SAME single GEMM as the
    un-adapted layer on an effective weight rebuilt once per optimizer step (osuf_dora_effective), and only the adapter
    gradients use the factored rank-r form (adapter_grads): no full-size weight gradient is ever formed."""

    def __init__(self, base_weight, lora_a, lora_b, magnitude, scaling: float) -> None:
        self.w, self.a, self.b, self.m, self.scaling = base_weight, lora_a, lora_b, magnitude, float(scaling)
        self._eff = None

    @property
    def params(self) -> Tuple[torch.Tensor, ...]:
        return (self.w, self.a, self.b) + ((self.m,) if self.m is not None else ())

    def effective(self):
        """-> (EffWeight standing for g*(W + s*BA), g (O,)); the gain is cached until a source parameter changes."""
        ver = (_WEIGHT_EPOCH, *[(p._version, p.data_ptr()) for p in self.params])
        if self._eff is None or self._eff[0] != ver:
            with torch.no_grad():
                m = self.m.detach().float().reshape(-1).contiguous() if self.m is not None else None
                g, t32, t16 = ops.dora_gain(self.w.detach().float().contiguous(), self.a.detach().float().contiguous(),
                                            self.b.detach().float().contiguous(), m, self.scaling)
                self.sg = g * self.scaling
                self.sgbt = {torch.float32: t32, torch.bfloat16: t16}          # (s g B)^T, the operand of du = dy (s g B)
            self._eff = (ver, EffWeight(self, g), g)
        return self._eff[1], self._eff[2]


def adapter_grads(ad: Adapter, dy, x, y, bias, kind: str, cache: PackCache, sums=None):
    """Gradients of (lora_A, lora_B, magnitude) from rows dy (B, Lout, O), the layer input x (B, Lin, I) and the layer output
    y (B, Lout, O) (= g*z + bias, as produced by the forward).  All GEMMs are rank-r:
        u  = A(x)                       (B, Lout, r)   tap-GEMM, N = r
        du = dy . (s g B)               (B, Lout, r)   K = O
        dB = s g (dy^T u),  dA = du^T x (per tap),  dm = (sum dy*y - bias * sum dy) / m      [g = m/norm, norm detached]
    sums: (sum_m dy*y, sum_m dy) per output channel when the caller already has them (BlockFn: from the GroupNorm backward)."""
    B_, Lin, I = x.shape
    O = dy.shape[-1]
    r = ad.a.shape[0]
    conv = ad.a.dim() == 3
    k = ad.a.shape[2] if conv else 1
    dt = x.dtype
    _, g = ad.effective()
    Lout, stride, pad, mode = _conv_geom(kind, k, Lin)
    with torch.no_grad():
        pa = cache.packs(("lora_a", id(ad), kind, dt), (ad.a,), ad.a, "same", dt)[0]                       # [k][r][I]
        sg, pbt = ad.sg, ad.sgbt[dt]                                                               # [1][r][O]
        u = ops.gemm_nt(x, pa, None, taps=k, lin=Lin, lout=Lout, stride=stride, pad=pad, mode=mode, out_shape=(B_, Lout, r))
        du = ops.gemm_nt(dy, pbt, None, out_shape=(B_, Lout, r))
        # (outputs of the skinny weight-gradient kernel: fp32 atomics into zeros.  Taken from the step's accumulator arena and added into
        #  -- one fill per step instead of a hipMemsetAsync per call: 360 ten-microsecond fills per DoRA step went that way)
        tb = ops.gemm_tn(dy, u, n1=O, out=ops.zeros((1, O, r), torch.float32, x.device), accumulate=True)   # [1][O][r] = dy^T u
        s0 = s1 = None
        if ad.m is not None:
            if sums is not None:                           # (sum dy*y, sum dy) already produced by the GroupNorm backward
                s0, s1 = sums
            else:
                s0 = ops.wcolsum(dy, y, None, B_, Lout).sum(0)
                s1 = ops.colsum(dy, O) if bias is not None else None
        bias32 = bias.detach().float().contiguous() if bias is not None else None
        if kind == "same":
            # dA^T instead of dA, so that the rank-r operand is the SECOND one (the skinny wgrad kernel wants N2 = r):
            #   G[t'][i][q] = sum_m' x[m'][i] du[m' + t' - (k-1-pad)][q]  ==  dA[q][i][k-1-t']      (same set of (row, tap) pairs)
            gt = ops.gemm_tn(x, du, taps=k, lin=Lin, lout=Lout, stride=1, pad=k - 1 - pad, mode=0, n1=I,
                             out=ops.zeros((k, I, r), torch.float32, x.device), accumulate=True)
            # one kernel scales dB, un-flips / transposes dA and finishes dm -- straight into the parameters' .grad when the
            # Trainer's flat gradient buffer is in use
            tgt = [grad_target(ad.a), grad_target(ad.b)] + ([grad_target(ad.m)] if ad.m is not None else [])
            direct = all(t is not None for t in tgt)
            da = tgt[0] if direct else torch.empty(ad.a.shape, dtype=torch.float32, device=x.device)
            db = tgt[1] if direct else torch.empty(ad.b.shape, dtype=torch.float32, device=x.device)
            dm = None
            if ad.m is not None:
                dm = tgt[2] if direct else torch.empty(ad.m.shape, dtype=torch.float32, device=x.device)
            ops.adapter_finish(tb, sg, db, gt, da, s0, s1, bias32, ad.m.detach().float().contiguous() if ad.m is not None else None, dm,
                               O, I, k, r, accumulate=direct)
            if direct:
                for p_ in (ad.a, ad.b) + ((ad.m,) if ad.m is not None else ()):
                    grad_done(p_)
                return None, None, None
            return da, db, dm
        db = tb[0] * sg[:, None]
        da = ops.gemm_tn(du, x, taps=k, lin=Lin, lout=Lout, stride=stride, pad=pad, mode=mode, n1=r, conv_layout=conv)
        da = da if conv else da[0]
        dm = None
        if ad.m is not None:
            sm = s0 - bias32 * s1 if bias is not None else s0
            dm = (sm / ad.m.detach().float().reshape(-1)).reshape(ad.m.shape)
    return da.reshape(ad.a.shape), db.reshape(ad.b.shape), dm


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
        with scope({"up": "Upsample (backward)", "down": "Downsample (backward)"}.get(ctx.kind, "Conv (backward)")):
            return ConvFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _rc(dy)
        dx = conv_dgrad(dy, w, ctx.cache, ctx.kind, x.shape[1], vp=ctx.vp) if ctx.needs_input_grad[0] else None
        dw, db = conv_wgrad_bias(dy, x, w, ctx.kind, ctx.bias_ref if ctx.has_bias else None, ctx.needs_input_grad[1], ctx.needs_input_grad[2], w.shape[0])
        return dx, dw, db, None, None, None


class AdaptedConvFn(torch.autograd.Function):
    """A LoRA / DoRA-adapted Conv1d / Linear on rows, on its own (lora_layers.py:312-328): one GEMM on the effective weight."""

    @staticmethod
    def forward(ctx, x, bias, cache, kind, adapter, la, lb, lm):
        y = conv_forward(x, adapter.effective()[0], bias, cache, kind, ("dora", *adapter.params))
        ctx.save_for_backward(x, y)
        ctx.cache, ctx.kind, ctx.adapter, ctx.bias_ref = cache, kind, adapter, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        ad, need = ctx.adapter, ctx.needs_input_grad
        dy = _rc(dy)
        dx = conv_dgrad(dy, ad.effective()[0], ctx.cache, ctx.kind, x.shape[1], vp=("dora", *ad.params)) if need[0] else None
        db = _bias_grad(dy, ctx.bias_ref, dy.shape[-1]) if ctx.bias_ref is not None and need[1] else None
        da = dlb = dm = None
        if need[5] or need[6] or need[7]:
            da, dlb, dm = adapter_grads(ad, dy, x, y, ctx.bias_ref, ctx.kind, ctx.cache)
        return dx, db, None, None, None, da, dlb, dm


class SkinnyLinearFn(torch.autograd.Function):
    """y = out_act(in_act(x) W^T + b) on (B, features) fp32 rows: the embedding-sized nn.Linear / 1x1 Conv1d layers
    (unet.py:356-367, residual.py:20-26,104-111), one kernel forward, two backward, master weights read in place."""

    @staticmethod
    def forward(ctx, x, w, bias, mode_dtype, in_act, out_act):
        x = x.contiguous()
        y = ops.skinny_fwd(x, w, bias, mode_dtype, in_act, out_act)
        ctx.save_for_backward(x, w, y if out_act else x)
        ctx.cfg, ctx.bias_ref = (mode_dtype, in_act, out_act), bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        mode_dtype, in_act, out_act = ctx.cfg
        need = ctx.needs_input_grad
        bias = ctx.bias_ref
        dy = dy.contiguous().float()
        dw = db = None
        tw = tb = None
        direct = False
        if need[1]:
            tw = grad_target(w)
            direct = tw is not None
            if not direct:
                dw = tw = torch.empty(w.shape, dtype=torch.float32, device=w.device)
            if bias is not None and need[2]:
                tb = grad_target(bias) if direct else None
                if tb is None:
                    db = tb = torch.zeros(bias.shape, dtype=torch.float32, device=w.device)
        dx = ops.skinny_bwd(dy, y if out_act else None, x, w, mode_dtype, in_act, out_act, need[0], tw, tb, direct)
        if direct:
            grad_done(w)
            if tb is not None and db is None:
                grad_done(bias)
        return dx, dw, db, None, None, None


class LinearGroupFn(torch.autograd.Function):
    """[in_act(x) W_i^T + b_i] for linears that share the input x -- the FiLM projections of every conditioned ResidualBlock of a
    UNet forward (residual.py:104-111,126-131: Sequential(SiLU, Linear) on the same cat(t, c)) -- one launch forward, one launch for
    dx = sum_i dy_i W_i backward.  Weight gradients: with a Trainer (direct gradient accumulation) every output passes through a
    _FilmTapFn in its block, which adds dW_i / db_i into the parameters' .grad the moment that block's backward has produced dy_i
    -- as early as the per-block form did, which is what the bucketed all-reduce overlaps with; without one they are returned here."""

    @staticmethod
    def forward(ctx, x, mode_dtype, in_act, direct, *wb):
        n = len(wb) // 2
        ws, bs = wb[:n], wb[n:]
        x = x.contiguous()
        ys = ops.skinny_fwd_group(x, ws, bs, mode_dtype, in_act)
        ctx.save_for_backward(x, *ws)
        ctx.cfg, ctx.bs = (mode_dtype, in_act, direct, n), bs
        ctx.set_materialize_grads(False)                    # an output nobody used stays None in backward (no zero tensors)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gouts):
        x, *ws = ctx.saved_tensors
        mode_dtype, in_act, direct, n = ctx.cfg
        dys = [g.contiguous().float() if g is not None else None for g in gouts]
        dx = ops.skinny_dx_group(dys, ws, x, mode_dtype, in_act) if ctx.needs_input_grad[0] else None
        dws, dbs = [None] * n, [None] * n
        if not direct:
            for i, (dy, w, b) in enumerate(zip(dys, ws, ctx.bs)):
                if dy is None or not ctx.needs_input_grad[4 + i]:
                    continue
                dws[i] = torch.empty(w.shape, dtype=torch.float32, device=w.device)
                if b is not None and ctx.needs_input_grad[4 + n + i]:
                    dbs[i] = torch.zeros(b.shape, dtype=torch.float32, device=w.device)
                ops.skinny_bwd(dy, None, x, w, mode_dtype, in_act, 0, False, dws[i], dbs[i], False)
        return (dx, None, None, None, *dws, *dbs)


def film_group(x: torch.Tensor, linears, mode_dtype: torch.dtype, in_act: int):
    """-> {id(weight): (output, tap info)} of LinearGroupFn over nn.Linear modules `linears`; blocks take theirs through film_tap().
    With a Trainer the weights enter the node DETACHED: they have no AccumulateGrad node (whose post-accumulate hook would only fire
    after the group node, at the very end of the backward); _FilmTapFn writes dW_i / db_i into .grad and reports the parameter
    complete (grad_used) when its block's backward has produced dy_i."""
    ws = tuple(l.weight for l in linears)
    bs = tuple(l.bias for l in linears)
    # (x must carry a graph: the taps hang off the group node's outputs, and with detached weights an input without one would leave
    #  the outputs -- and so the FiLM weights -- without any gradient path)
    direct = _DIRECT and _GRAD_COMPLETE_CB is not None and torch.is_grad_enabled() and x.requires_grad and \
        all(not w.requires_grad or grad_target(w) is not None for w in ws) and \
        all(b is None or not b.requires_grad or grad_target(b) is not None for b in bs)
    if direct:
        ys = LinearGroupFn.apply(x, mode_dtype, in_act, True, *(w.detach() for w in ws), *(None if b is None else b.detach() for b in bs))
        xs = x.detach()
        return {id(w): (y, (xs, w, b, mode_dtype, in_act) if y.requires_grad and w.requires_grad else None) for w, b, y in zip(ws, bs, ys)}
    ys = LinearGroupFn.apply(x, mode_dtype, in_act, False, *ws, *bs)
    return {id(w): (y, None) for w, y in zip(ws, ys)}


def film_tap(entry):
    """One parked output of film_group, taken by its block.  With a Trainer it passes through _FilmTapFn HERE, in the block's forward:
    autograd runs ready nodes in reverse creation order, so a tap created next to its block runs right after that block's backward
    (created up front with the group, all 35 would run last, together)."""
    y, info = entry
    if info is None:
        return y
    xs, w, b, mode_dtype, in_act = info
    grad_use(w)
    if b is not None and b.requires_grad:
        grad_use(b)
    return _FilmTapFn.apply(y, xs, (w, b), mode_dtype, in_act)        # (w, b) as a tuple: not autograd inputs


class _FilmTapFn(torch.autograd.Function):
    """Identity on one output of LinearGroupFn.  Its backward runs the moment the consuming block has produced dy_i (a tensor hook
    on the group's output would only run with the group node itself, after ALL blocks): it adds dW_i / db_i into the parameters'
    .grad and reports them complete, then hands dy_i on to the group node for dx."""

    @staticmethod
    def forward(ctx, y, x, wb, mode_dtype, in_act):
        ctx.refs = (x, wb[0], wb[1], mode_dtype, in_act)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        x, w, b, mode_dtype, in_act = ctx.refs
        dy = dy.contiguous().float()
        tb = grad_target(b) if b is not None and b.requires_grad else None
        ops.skinny_bwd(dy, None, x, w.detach(), mode_dtype, in_act, 0, False, grad_target(w), tb, True)
        grad_used(w)
        if tb is not None:
            grad_used(b)
        return dy, None, None, None, None


_UNIT_NORM: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def _unit_norm(C: int, device):
    """(ones(C), zeros(C)) standing for gamma / beta of the identity norm of Block(norm=False)."""
    key = (C, str(device))
    if key not in _UNIT_NORM:
        _UNIT_NORM[key] = (torch.ones(C, dtype=torch.float32, device=device), torch.zeros(C, dtype=torch.float32, device=device))
    return _UNIT_NORM[key]


class BlockFn(torch.autograd.Function):
    """conv k3 -> GroupNorm(1,C) -> FiLM -> SiLU   (residual.py:75-84).  ss: fp32 (B, 2C) = (scale | shift) or None.
    gamma / beta None: Block(norm=False) (residual.py:71, nn.Identity instead of the GroupNorm) -- the same kernels with mean 0, rstd 1.
    adapter (+ its parameters la, lb, lm as autograd inputs): LoRA / DoRA on the conv (block{1,2}.proj, trainer_peft.py:241)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, ss, cache, adapter=None, la=None, lb=None, lm=None, reslink=None):
        B, L, _ = x.shape
        C = w.shape[0]
        ctx.identity_norm = gamma is None
        if ctx.identity_norm:
            gamma, beta = _unit_norm(C, x.device)
        repro = ops.reproducible() or ctx.identity_norm   # sampler: statistics by fixed-order reductions, not epilogue atomics
        stats = None if repro else ops.zeros((B, 2), torch.float64, x.device)
        if adapter is not None:
            weff, _ = adapter.effective()
            y = conv_forward(x, weff, bias, cache, "same", ("dora", *adapter.params), stats=stats)
        else:
            y = conv_forward(x, w, bias, cache, "same", None, stats=stats)
        ssc = ss.contiguous() if ss is not None else None
        if ctx.identity_norm:
            mr = torch.tensor([0.0, 1.0], dtype=torch.float32, device=x.device).repeat(B, 1)
            h = ops.gn_apply(y, mr, gamma, beta, ssc, L)
        elif repro:
            h, mr = ops.gn_apply_reproducible(y, gamma, beta, ssc, L)         # partial sums, finished in fixed order inside the apply kernel
        else:
            h, mr = ops.gn_apply_from_stats(y, stats, gamma, beta, ssc, L)       # the epilogue's raw sums are finalised in the kernel
        ctx.save_for_backward(x, w, y, mr, gamma, beta, ssc if ssc is not None else mr)
        ctx.cache, ctx.has_ss, ctx.adapter = cache, ss is not None, adapter
        ctx.bias_ref, ctx.reslink = bias, reslink
        return h

    @staticmethod
    def backward(ctx, dh):
        with scope("Residual's Block (backward)"):
            return BlockFn._backward(ctx, dh)

    @staticmethod
    def _backward(ctx, dh):
        x, w, y, mr, gamma, beta, ss = ctx.saved_tensors
        ss = ss if ctx.has_ss else None
        L = x.shape[1]
        need = ctx.needs_input_grad
        tg, tb = (None, None) if ctx.identity_norm else (grad_target(gamma), grad_target(beta))
        direct_norm = tg is not None and tb is not None
        # conv-bias gradient (= column sums of dy) comes out of the GroupNorm backward in closed form: no extra pass over dy
        bias = ctx.bias_ref
        ad = ctx.adapter
        db = None
        tbias = dyy = sdy = None
        if need[2] and bias is not None:
            tbias = grad_target(bias)
            if tbias is None:
                db = tbias = torch.zeros(bias.shape, dtype=torch.float32, device=dh.device)
        want_dm = ad is not None and ad.m is not None and need[10]
        if want_dm:                                        # DoRA magnitude: column sums of dy*y and of dy, same closed form
            dyy = ops.zeros((y.shape[-1],), torch.float32, dh.device)
            sdy = tbias if db is not None else ops.zeros((y.shape[-1],), torch.float32, dh.device)
        dy, dgamma, dbeta, dss = ops.gn_bwd(_rc(dh), y, mr, gamma, beta, ss, L, tg if direct_norm else None, tb if direct_norm else None,
                                            sdy if want_dm else tbias, dyy, identity_norm=ctx.identity_norm)
        if tbias is not None and db is None:
            if want_dm:
                tbias.add_(sdy)
            grad_done(bias)
        if direct_norm:
            grad_done(gamma); grad_done(beta)
            dgamma = dbeta = None
        dx = None
        if need[0]:
            # + the gradient that reached x through the block's residual path (parked by GateRes*Fn.backward, see ResLink): it rides
            # this GEMM's epilogue instead of a separate elementwise add over (B, L, C)
            extra = None
            if ctx.reslink is not None and ctx.reslink.dx is not None:
                extra, ctx.reslink.dx = ctx.reslink.dx, None
            dx = conv_dgrad(dy, ad.effective()[0], ctx.cache, "same", L, residual=extra, vp=("dora", *ad.params)) if ad is not None \
                else conv_dgrad(dy, w, ctx.cache, "same", L, residual=extra)
        dw = conv_wgrad(dy, x, w, "same") if need[1] else None
        da = dlb = dm = None
        if ad is not None and (need[8] or need[9] or need[10]):
            da, dlb, dm = adapter_grads(ad, dy, x, y, ctx.bias_ref, "same", ctx.cache, sums=(dyy, sdy) if want_dm else None)
        return dx, dw, db, dgamma if need[3] else None, dbeta if need[4] else None, dss, None, None, da, dlb, dm, None


class GateLink:
    """Per-forward hand-off between the two Functions that both differentiate the block output h (gate * h + res, and the
    GlobalContext pooling of h).  GateRes*Fn.backward runs first (it is downstream): instead of materialising dout * gate and
    letting autograd add the pooling path's gradient to it later (one elementwise kernel + one add over (B, L, C) per block), it
    parks (dout, gate) here and returns no gradient for h; GCAPoolFn.backward then emits the complete
    dh = dout * gate + p * dpooled + dlogit * wk from the one fused kernel."""
    __slots__ = ("dout", "gate")

    def __init__(self) -> None:
        self.dout = self.gate = None


class ResLink:
    """Per-forward hand-off of the residual path's input gradient inside one ResidualBlock (residual.py:137: h + res_conv(x)): x
    feeds both block1's convolution and the residual add, so autograd would add two (B, L, C) gradients with an elementwise kernel.
    GateRes*Fn.backward (it runs first: it is downstream of block1) parks its contribution here and returns none for x; block1's
    BlockFn.backward then passes it as the residual operand of its input-gradient GEMM (added in the epilogue)."""
    __slots__ = ("dx",)

    def __init__(self) -> None:
        self.dx = None


class GCAPoolFn(torch.autograd.Function):
    """GlobalContext pooling: pooled[b,c] = sum_n softmax_n(h.wk + bk)[n] * h[b,n,c]   (residual.py:29-31) -> fp32 (B,C)."""

    @staticmethod
    def forward(ctx, h, wk, bk, link=None):
        B, L, C = h.shape
        wkv = wk.reshape(-1).contiguous()
        if ops.gca_pool_fused_ok():                        # one pass over h: logits, running softmax, weighted column sums (round 5)
            pooled, p = ops.gca_pool(h, wkv, bk.reshape(-1), L)
        else:
            p = ops.rowdot(h, wkv, bk.reshape(-1), L)
            ops.softmax_rows_(p, B, L)
            pooled = ops.wcolsum(h, None, p, B, L)
        ctx.save_for_backward(h, wkv, p, pooled)
        ctx.wshape, ctx.link = wk.shape, link
        ctx.wk_ref, ctx.bk_ref = wk, bk                   # the Parameters themselves (direct .grad accumulation)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        with scope("GlobalContext (backward)"):
            return GCAPoolFn._backward(ctx, dpooled)

    @staticmethod
    def _backward(ctx, dpooled):
        h, wkv, p, pooled = ctx.saved_tensors
        B, L, C = h.shape
        dpooled = dpooled.contiguous().float()
        sdot = ops.rowdot(pooled, dpooled, None, 1, per_sample=True)       # sum_c pooled * dpooled per sample, one launch
        link = ctx.link
        if link is not None and link.dout is not None:
            dout, gate = link.dout, link.gate
            link.dout = link.gate = None
        else:                                              # stand-alone GlobalContext: no dout * gate term
            dout, gate = h, torch.zeros((B, C), dtype=torch.float32, device=h.device)
        # to_k's weight / bias gradients come out of the same kernel (it holds the h rows); straight into .grad under a Trainer
        need_w, need_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dwk = dbk = tw = tb = None
        direct = False
        if need_w or need_b:                               # (frozen base: neither, the kernel skips the sums)
            gw, gb = grad_target(ctx.wk_ref), grad_target(ctx.bk_ref)
            direct = gw is not None and gb is not None
            if direct:
                tw, tb = gw.view(-1), gb.view(-1)
            else:
                tw = ops.zeros(C, torch.float32, h.device)
                tb = ops.zeros(1, torch.float32, h.device)
        dh, dlogit = ops.gca_bwd_apply(dout, h, p, gate, dpooled, sdot, wkv, L, tw, tb)
        if direct:
            grad_done(ctx.wk_ref); grad_done(ctx.bk_ref)
        elif tw is not None:
            dwk = tw.reshape(ctx.wshape) if need_w else None
            dbk = tb.reshape(ctx.bk_ref.shape) if need_b else None
        return dh, dwk, dbk, None


class MeanPoolFn(torch.autograd.Function):
    """SqueezeExcite's AdaptiveAvgPool1d(1) (residual.py:42,51): pooled[b, c] = mean_l h[b, l, c] -> fp32 (B, C).  Same hand-off with
    the gate * h consumer as GCAPoolFn (GateLink); the backward reuses osuf_gca_bwd_apply with uniform weights 1/L and a zero logit
    weight: dh = dout * gate + dpooled / L."""

    @staticmethod
    def forward(ctx, h, link=None):
        B, L, C = h.shape
        ctx.save_for_backward(h)
        ctx.link = link
        return ops.wcolsum(h, None, None, B, L) * (1.0 / L)

    @staticmethod
    def backward(ctx, dpooled):
        (h,) = ctx.saved_tensors
        B, L, C = h.shape
        dev = h.device
        dpooled = dpooled.contiguous().float()
        link = ctx.link
        if link is not None and link.dout is not None:
            dout, gate = link.dout, link.gate
            link.dout = link.gate = None
        else:
            dout, gate = h, torch.zeros((B, C), dtype=torch.float32, device=dev)
        p = torch.full((B * L,), 1.0 / L, dtype=torch.float32, device=dev)
        dh, _ = ops.gca_bwd_apply(dout, h, p, gate, dpooled, torch.zeros(B, dtype=torch.float32, device=dev),
                                  torch.zeros(C, dtype=torch.float32, device=dev), L)
        return dh, None


def _gate_dh(link, dout, gate, L):
    """Gradient of gate * h w.r.t. h: deferred to GCAPoolFn.backward when the two are linked (see GateLink)."""
    if link is not None:
        link.dout, link.gate = dout, gate
        return None
    return ops.gate_residual(dout, gate, None, L)


class GateResFn(torch.autograd.Function):
    """out = h * gate + res   (residual.py:135-137, identity res_conv).  gate fp32 (B, C)."""

    @staticmethod
    def forward(ctx, h, gate, res, link=None, rlink=None):
        L = h.shape[1]
        gate = gate.contiguous()
        ctx.save_for_backward(h, gate)
        ctx.link, ctx.rlink = link, rlink
        return ops.gate_residual(h, gate, res, L)

    @staticmethod
    def backward(ctx, dout):
        h, gate = ctx.saved_tensors
        B, L, C = h.shape
        dout = _rc(dout)
        dgate = ops.wcolsum(dout, h, None, B, L)
        dres = dout
        if ctx.rlink is not None and ctx.needs_input_grad[2]:
            ctx.rlink.dx, dres = dout, None
        return _gate_dh(ctx.link, dout, gate, L), dgate, dres, None, None


class GateResConvFn(torch.autograd.Function):
    """out = h * gate + res_conv(x)  (residual.py:135-137, 1x1 res_conv): the gate-multiply rides the GEMM epilogue."""

    @staticmethod
    def forward(ctx, h, gate, x, w, bias, cache, link=None, rlink=None):
        gate = gate.contiguous()
        ctx.save_for_backward(h, gate, x, w)
        ctx.cache, ctx.bias_ref, ctx.link, ctx.rlink = cache, bias, link, rlink
        return conv_forward(x, w, bias, cache, "same", None, residual=h, rscale=gate)

    @staticmethod
    def backward(ctx, dout):
        h, gate, x, w = ctx.saved_tensors
        B, L, C = h.shape
        dout = _rc(dout)
        dgate = ops.wcolsum(dout, h, None, B, L)
        dh = _gate_dh(ctx.link, dout, gate, L)
        dx = conv_dgrad(dout, w, ctx.cache, "same", L) if ctx.needs_input_grad[2] else None
        if ctx.rlink is not None and dx is not None:
            ctx.rlink.dx, dx = dx, None
        dw, db = conv_wgrad_bias(dout, x, w, "same", ctx.bias_ref, ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        return dh, dgate, dx, dw, db, None, None, None


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
        need = ctx.needs_input_grad
        dw2, db2 = conv_wgrad_bias(dout, h, w2, "same", ctx.b2, need[3], need[4])
        dw1, db1 = conv_wgrad_bias(dpre, x, w1, "same", ctx.b1, need[1], need[2])
        wd1 = cache.packs(("p", "same", x.dtype, ""), (w1,), w1, "same", x.dtype)[1]
        dx = ops.gemm_nt(dpre, wd1, None, residual=dout, out_shape=x.shape)
        return dx, dw1, db1, dw2, db2, None


_ROPE_TABLES: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def rope_tables(n: int, dim: int, scale_base: int, device, theta: float = 10000.0):
    """cos/sin tables [n][dim/2] in fp32, built exactly as attention.py:24-47 does for an fp32 q (on the host)."""
    key = (n, dim, scale_base, float(theta), str(device))
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
    """LayerNorm -> to_q/to_kv -> RoPE -> MQA flash attention -> to_out + residual(normed x)   (unet.py:125-141).
    aq / akv: LoRA / DoRA adapters on to_q / to_kv (trainer_peft.py:241), their parameters passed as autograd inputs."""

    @staticmethod
    def forward(ctx, x, nw, nb, wq, wkv, wo, bo, cache, heads, dim_head, scale_base, aq=None, akv=None,
                qa=None, qb=None, qm=None, kva=None, kvb=None, kvm=None, kv_heads=1, base=None, grad_mode=False):
        # kv_heads = G > 1 (grouped-query attention, unet.py:132-135): the caller hands wq with its head blocks in GROUP-MAJOR order
        # (heads g*H/G .. share K/V head g) and wo with its input columns permuted alike (modules/unet.py Attention._forward_gqa);
        # base = (to_q.weight, to_out.weight): the parameters those two views derive from -- what the pack cache must version on
        B, N, C = x.shape
        H, D, G = heads, dim_head, kv_heads
        dt = x.dtype
        xn, mr = ops.ln_fwd(x, nw, nb)
        wqkv = AttentionFn._qkv_packs(cache, dt, wq, wkv, aq, akv, base)[0]
        qkv = ops.gemm_nt(xn, wqkv, None, out_shape=(B, N, (H + 2 * G) * D))
        cos, sin = rope_tables(N, D, scale_base, x.device)
        scale = D ** -0.5
        # the softmax scale (x log2 e) rides the queries' one bf16 rounding where the kernels have the pre-scaled form: the attention backward's
        # generated loop then takes Qs K^T - lse2 straight as the exponent (64 fewer vector instructions per (head, query block) pair)
        qs = ops.q_prescale_ok(D, ops.ATTN_BWD_DEFAULT)
        # a backward will follow: the forward kernel zero-fills the dQ accumulator of this layer's fused backward sweep (private workspace, kept to then)
        # (grad_mode: the caller's torch.is_grad_enabled() -- needs_input_grad alone is also true under no_grad, e.g. in the sampler)
        will_bwd = grad_mode and any(ctx.needs_input_grad)
        dq_ws = ops.fused_bwd_workspace(B, N, H, D, dt, x.device) if (G == 1 and will_bwd) else None
        # (with the stored queries AND the zero fill behind its loop the kernel's tail grows by ~30-50 us whatever N is: measured against the 15 us a
        #  RoPE pass over N = 1024 rows saves, the training forward of short sequences keeps the separate pass -- tools/time_fwd_rope.py)
        if ops.fwd_rope_ok(qkv, D, G, qs) and (N >= 2048 or not will_bwd):
            # the attention kernel rotates / scales / rounds its own query tiles (and stores them for the backward): only K | V take the RoPE + cast pass
            qkv_r, o, lse = ops.mqa_fwd_rope(qkv, cos, sin, B, N, H, D, dt, scale, write_q=will_bwd, zero_dq=dq_ws)
        else:
            qkv_r = ops.rope_cast(qkv, cos, sin, N, H + G, H + 2 * G, D, q_mul=scale * ops.LOG2E if qs else 1.0, n_q_heads=H)   # rotate q and k heads; cast v
            o, lse = ops.mqa_fwd(qkv_r, B, N, H, D, dt, scale, kv_heads=G, qs=qs, zero_dq=dq_ws)
        # DoRA adapters on to_q / to_kv: their magnitude gradient needs sum_m dy * y over the PRE-RoPE projections.  Kept from here (604 MB per N = 4096
        # layer at B = 64: 23 GB over the UNet, affordable on 288 GB) instead of running the q|kv GEMM a second time in the backward (39 GEMMs of
        # ~260 us per DoRA step); OSUF_ATTN_RECOMPUTE_QKV=1 restores the recompute
        keep_raw = will_bwd and ((aq is not None and aq.m is not None) or (akv is not None and akv.m is not None)) and \
            os.environ.get("OSUF_ATTN_RECOMPUTE_QKV") != "1"
        ctx.qkv_raw = qkv if keep_raw else None
        del qkv
        ctx.dq_ws = dq_ws
        wpo = cache.packs(("po", dt), (wo,) if base is None else (base[1],), wo, "same", dt)[0]
        out = ops.gemm_nt(o, wpo, bo, residual=xn, out_shape=x.shape)
        ctx.base = base
        ctx.save_for_backward(x, nw, mr, xn, wq, wkv, wo, qkv_r, o, lse)
        ctx.cache, ctx.geom = cache, (H, D, scale_base, scale, G)
        ctx.nb, ctx.bo = nb, bo
        ctx.aq, ctx.akv = aq, akv
        ctx.qs = qs
        return out

    @staticmethod
    def _qkv_packs(cache, dt, wq, wkv, aq, akv, base=None):
        """Stacked (q | kv) projection operands; adapted halves use their effective weight."""
        if aq is None and akv is None:
            return cache.packs(("qkv", dt), (wq, wkv) if base is None else (base[0], wkv), (wq, wkv), "same", dt)
        ws = (aq.effective()[0] if aq is not None else wq, akv.effective()[0] if akv is not None else wkv)
        params = (wq, wkv) + (aq.params if aq is not None else ()) + (akv.params if akv is not None else ())
        return cache.packs(("qkv_dora", dt), params, ws, "same", dt)

    @staticmethod
    def backward(ctx, dout):
        with scope("Attention (backward)"):
            return AttentionFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        x, nw, mr, xn, wq, wkv, wo, qkv_r, o, lse = ctx.saved_tensors
        H, D, scale_base, scale, G = ctx.geom
        cache = ctx.cache
        need = ctx.needs_input_grad
        B, N, C = x.shape
        dt = x.dtype
        dout = _rc(dout)
        # to_out
        dwo, dbo = conv_wgrad_bias(dout, o, wo, "same", ctx.bo, need[5], need[6])
        wdo = cache.packs(("po", dt), (wo,) if ctx.base is None else (ctx.base[1],), wo, "same", dt)[1]
        if ops.FUSE_ROWDOT and G == 1 and D == 64:                             # (the epilogue sums 64-column heads)
            do, delta = ops.gemm_nt_rowdot(dout, wdo, o, N, H)                   # dO and sum_d dO * O from one epilogue
        else:
            do, delta = ops.gemm_nt(dout, wdo, None, out_shape=o.shape), None
        do16 = ops.cast_rows(do, torch.bfloat16)                                 # SDPA backward runs in bf16 (attention.py:101)
        # attention + rope
        cos, sin = rope_tables(N, D, scale_base, x.device)
        variant = ops.ATTN_BWD_DEFAULT
        if ctx.qs and variant not in ops._FUSED_DQ_MODE:                         # the default changed between forward and backward (tests do): the
            variant = ops.ATTN_FUSED                                             # saved queries are pre-scaled, only the fused sweeps read those
        dqkv = ops.mqa_bwd(qkv_r, o, do16, lse, B, N, H, D, scale, dt, cos, sin, variant=variant, delta=delta,
                           kv_heads=G, qs=ctx.qs, workspace=ctx.dq_ws)           # RoPE transpose + cast ride the kernels' epilogues
        ctx.dq_ws = None                                                         # (a second backward through this node falls back to the memset)
        # to_q / to_kv
        dwq = conv_wgrad(dqkv[..., : H * D], xn, wq, "same") if need[3] else None
        dwkv = conv_wgrad(dqkv[..., H * D:], xn, wkv, "same") if need[4] else None
        aq, akv = ctx.aq, ctx.akv
        packs = AttentionFn._qkv_packs(cache, dt, wq, wkv, aq, akv, ctx.base)
        gq = gkv = (None, None, None)
        if (aq is not None and any(need[13:16])) or (akv is not None and any(need[16:19])):
            qkv = ctx.qkv_raw                                                     # pre-RoPE projections (for d magnitude): kept by the forward, or again
            if qkv is None:
                qkv = ops.gemm_nt(xn, packs[0], None, out_shape=(B, N, (H + 2 * G) * D))
            ctx.qkv_raw = None
            if aq is not None:
                gq = adapter_grads(aq, dqkv[..., : H * D], xn, qkv[..., : H * D], None, "same", cache)
            if akv is not None:
                gkv = adapter_grads(akv, dqkv[..., H * D:], xn, qkv[..., H * D:], None, "same", cache)
            del qkv
        dxn = ops.gemm_nt(dqkv, packs[1], None, residual=dout, out_shape=x.shape)   # + residual path (x + to_out(..), x = normed)
        tg, tb = grad_target(nw), grad_target(ctx.nb)
        direct_norm = tg is not None and tb is not None
        dx, dnw, dnb = ops.ln_bwd(dxn, x, mr, nw, tg if direct_norm else None, tb if direct_norm else None)
        if direct_norm:
            grad_done(nw); grad_done(ctx.nb)
            dnw = dnb = None
        return (dx, dnw if need[1] else None, dnb if need[2] else None, dwq, dwkv, dwo, dbo, None, None, None, None, None, None,
                *gq, *gkv, None, None, None)


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
