"""Compute-dtype policy and layout helpers shared by the modules.

Activations inside the network are channels-last "rows" (B, L, C).  The reference's API is (B, C, L); a rows
tensor is exposed to callers as its zero-copy permuted view, and an incoming permuted view is unwrapped again
without a copy, so a chain of modules never transposes anything (unet.py:180,183 become no-ops).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import functional as Fn
from . import ops

_FORCED: Optional[torch.dtype] = None


def set_compute_dtype(dtype: Optional[torch.dtype]) -> None:
    """None: follow torch autocast (bf16 autocast -> bf16 kernels, otherwise fp32); or force bf16 / fp32."""
    global _FORCED
    assert dtype in (None, torch.float32, torch.bfloat16)
    _FORCED = dtype


class forced_compute_dtype:
    """Context manager: force the compute dtype inside (used by OsuFusion.set_full_bf16 and the tests)."""

    def __init__(self, dtype: Optional[torch.dtype]) -> None:
        self.dtype = dtype

    def __enter__(self):
        global _FORCED
        self.prev = _FORCED
        if self.dtype is not None:
            _FORCED = self.dtype
        return self

    def __exit__(self, *exc):
        global _FORCED
        _FORCED = self.prev
        return False


def compute_dtype(param_dtype: torch.dtype = torch.float32) -> torch.dtype:
    if _FORCED is not None:
        return _FORCED
    if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return torch.bfloat16
    return torch.bfloat16 if param_dtype == torch.bfloat16 else torch.float32


def require_gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError("osufusion_amd runs on MI355X only: its HIP kernels have no CPU / eager-PyTorch fallback "
                           "(move the module and its inputs to cuda)")


def is_rows_view(x: torch.Tensor) -> bool:
    return x.dim() == 3 and x.stride(1) == 1 and x.shape[1] > 0 and (x.stride(2) >= x.shape[1] or x.shape[2] == 1)


def to_rows(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """logical (B, C, L) -> rows (B, L, C) in `dtype` (zero-copy when x already is a rows view of that dtype)."""
    require_gpu(x)
    if is_rows_view(x) and x.shape[1] % 8 == 0:
        r = x.permute(0, 2, 1)
        return r if r.dtype == dtype else _CastFn.apply(r, dtype)
    C = x.shape[1]
    if C % 8:
        raise ValueError(f"channel count {C} must be a multiple of 8 for the HIP kernels")
    return Fn.RowsFromNCLFn.apply(x, dtype, C, 1)


def to_logical(rows: torch.Tensor) -> torch.Tensor:
    """rows (B, L, C) -> logical (B, C, L) view."""
    return rows.permute(0, 2, 1)


class _CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dt = x.dtype
        return ops.cast_rows(x.contiguous(), dtype)

    @staticmethod
    def backward(ctx, g):
        return ops.cast_rows(g.contiguous(), ctx.dt), None


def cast_rows(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    return x if x.dtype == dtype else _CastFn.apply(x, dtype)


_CAT_MEMO = [None, None, None, None, None]


def _version_of(v: Optional[torch.Tensor]) -> int:
    if v is None:
        return -1
    try:
        return v._version
    except RuntimeError:                                   # inference tensors keep no version counter (sample() runs under
        return -2                                          # inference_mode; denoise_rows drops the memo after every forward)


def shared_cat(t: Optional[torch.Tensor], c: Optional[torch.Tensor]) -> torch.Tensor:
    """cat((t, c), -1) as fp32, memoised on the identity of (t, c) and on the autograd mode: every conditioned ResidualBlock of one
    UNet forward gets the same tensor (residual.py:126-127 re-concatenates per block), so autograd sums their gradients into one
    node.  The grad mode is part of the key: under reentrant activation checkpointing the first pass runs without grad, and a
    tensor memoised there must not be handed to the blocks outside the checkpointed region (it carries no graph)."""
    mode = torch.is_grad_enabled()
    ver = (_version_of(t), _version_of(c))                  # an in-place update of t / c (static buffers) is a new value
    if _CAT_MEMO[2] is not None and _CAT_MEMO[0] is t and _CAT_MEMO[1] is c and _CAT_MEMO[3] == mode and _CAT_MEMO[4] == ver:
        return _CAT_MEMO[2]
    e = torch.cat([v for v in (t, c) if v is not None], dim=-1).float()
    _CAT_MEMO[0], _CAT_MEMO[1], _CAT_MEMO[2], _CAT_MEMO[3], _CAT_MEMO[4] = t, c, e, mode, ver
    return e


def clear_shared_cat() -> None:
    _CAT_MEMO[0] = _CAT_MEMO[1] = _CAT_MEMO[2] = _CAT_MEMO[3] = _CAT_MEMO[4] = None
    film_clear()


_FILM_MEMO = [None, None]                                  # [the embedding the group ran on, {id(weight): output}]


def film_prepare(emb: torch.Tensor, linears) -> bool:
    """Run the FiLM projections Sequential(SiLU, Linear) of `linears` on the shared embedding `emb` (= shared_cat(t, c)) in one launch
    and park the outputs for film_take().  Skipped (-> False: every block then runs its own small_linear) when the group kernels'
    conditions do not hold, under hipGraph capture (the descriptor table is uploaded per call) and for non-fp32 masters."""
    _FILM_MEMO[0] = _FILM_MEMO[1] = None
    if not linears or not emb.is_cuda or torch.cuda.is_current_stream_capturing():
        return False
    ws = [l.weight for l in linears]
    x = emb.float().contiguous()
    if any(w.dtype != torch.float32 for w in ws) or not ops.skinny_group_ok(x, ws):
        return False
    _FILM_MEMO[0], _FILM_MEMO[1] = emb, Fn.film_group(x, linears, compute_dtype(ws[0].dtype), ops.ACT_SILU)
    return True


def film_take(emb: torch.Tensor, lin) -> Optional[torch.Tensor]:
    """The parked output of film_prepare for this Linear, if the group ran on this very embedding tensor.  The entry stays parked: a
    ResidualBlock applied twice in one forward takes a fresh tap per use (film_tap use-counts the weight, so its gradient is
    reported complete once, after the last use -- a second use falling back to small_linear on the attached weight would report
    it a second time through its AccumulateGrad hook)."""
    if _FILM_MEMO[1] is None or _FILM_MEMO[0] is not emb:
        return None
    entry = _FILM_MEMO[1].get(id(lin.weight))
    return None if entry is None else Fn.film_tap(entry)


def film_clear() -> None:
    _FILM_MEMO[0] = _FILM_MEMO[1] = None


def small_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], in_act: int = 0, out_act: int = 0) -> torch.Tensor:
    """out_act(in_act(x) @ w(N, K[, 1])^T + b) for the embedding-sized MLPs (time / cond / FiLM / GlobalContext): fp32 rows in, fp32
    out, the fp32 master weight read in place by the skinny-linear kernels (csrc/skinny.hip).  in_act: ops.ACT_SILU fuses the
    SiLU the reference applies to the input (nn.Sequential(SiLU, Linear)); out_act: ops.ACT_SIGMOID fuses GlobalContext's sigmoid.

    The products run in the compute dtype, as the reference's do: fp32 run -> exact-f32 MFMA; bf16 autocast -> operands rounded to
    bf16, fp32 accumulation (torch autocast casts nn.Linear / 1x1 Conv1d to bf16 too)."""
    require_gpu(x)
    dt = compute_dtype(w.dtype)
    if w.dtype != torch.float32:                           # a model cast with .to(bfloat16): the kernels read fp32 masters
        w, b = w.float(), (b.float() if b is not None else None)
    return Fn.SkinnyLinearFn.apply(x.float(), w, b, dt, in_act, out_act)
