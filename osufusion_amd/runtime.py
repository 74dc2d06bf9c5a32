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


def small_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], cache: Fn.PackCache, tag: str,
                 vparams: Optional[tuple] = None) -> torch.Tensor:
    """(B, K) @ w(N, K)^T + b for the embedding-sized MLPs (time / cond / FiLM / GlobalContext), fp32 in, fp32 out.

    The GEMM runs in the compute dtype, as the reference does: fp32 run -> exact-f32 MFMA; bf16 autocast -> bf16 operands with
    fp32 accumulation (torch autocast casts nn.Linear / 1x1 Conv1d to bf16 too).  The f32 MFMA runs at 1/16 of the bf16 rate,
    which made these 280 M=32 GEMMs cost 18 ms of a 340 ms bf16 step when they were pinned to fp32."""
    require_gpu(x)
    dt = compute_dtype(w.dtype)
    K = x.shape[-1]
    x = x.float()
    w2 = w.reshape(w.shape[0], -1).float()
    if K % 8:                                    # e.g. cond_mlp.0: Linear(5, E)
        padk = 8 - K % 8
        x = torch.nn.functional.pad(x, (0, padk))
        w2 = torch.nn.functional.pad(w2, (0, padk))
    N = w2.shape[0]
    padn = (-N) % 8
    if padn:
        w2 = torch.nn.functional.pad(w2, (0, 0, 0, padn))
        b = torch.nn.functional.pad(b, (0, padn)) if b is not None else None
    xr = cast_rows(x.contiguous().unsqueeze(0), dt)
    y = Fn.ConvFn.apply(xr, w2, b.float() if b is not None else None, cache, "same", (tag, *(vparams if vparams is not None else (w,))))
    y = cast_rows(y, torch.float32).squeeze(0)
    return y[:, :N] if padn else y
