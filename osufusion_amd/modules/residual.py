"""HIP-backed mirror of osu_fusion/modules/residual.py: same classes, ctor signatures and state_dict keys.

The nn.Conv1d / nn.GroupNorm / nn.Linear children are parameter containers only (identical names, shapes and
default init as the reference); their torch forward is never called -- compute goes through the gfx950 kernels.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: N812

from .. import functional as Fn
from .. import ops
from .. import runtime as rt
from ..tracing import scope


class GlobalContext(nn.Module):
    """residual.py:14-37.  forward(x: (B,C,L)) -> gate (B, C_out, 1)."""

    def __init__(self, dim_in: int, dim_out: int, reduction: int = 2, dim_min: int = 8) -> None:
        super().__init__()
        self.to_k = nn.Conv1d(dim_in, 1, 1)
        inner_dim = max(dim_min, dim_out // reduction)
        self.layers = nn.Sequential(
            nn.Conv1d(dim_in, inner_dim, 1),
            nn.SiLU(),
            nn.Conv1d(inner_dim, dim_out, 1),
            nn.Sigmoid(),
        )

    def gate_from_rows(self, h: torch.Tensor, link=None) -> torch.Tensor:
        """rows (B, L, C) -> gate fp32 (B, C_out).  link: functional.GateLink shared with the gate * h consumer."""
        with scope("GlobalContext"):                       # the reference's record_function scope (residual.py:35)
            pooled = Fn.GCAPoolFn.apply(h, self.to_k.weight, self.to_k.bias, link)
            l0, l2 = self.layers[0], self.layers[2]
            z = rt.small_linear(pooled, l0.weight, l0.bias)
            return rt.small_linear(z, l2.weight, l2.bias, in_act=ops.ACT_SILU, out_act=ops.ACT_SIGMOID)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = rt.to_rows(x, rt.compute_dtype(self.to_k.weight.dtype))
        return self.gate_from_rows(h).unsqueeze(-1)


class SqueezeExcite(nn.Module):
    """residual.py:40-59: mean over the sequence -> Conv1d(1x1) -> SiLU -> Conv1d(1x1) -> sigmoid.  forward(x: (B,C,L)) -> (B, C_out, 1).
    The gate ResidualBlock(use_gca=False) uses (residual.py:116); same state-dict keys (`layers.0`, `layers.2`; the pool has none)."""

    def __init__(self, dim: int, dim_out: int, reduction: int = 2, dim_minimum: int = 8) -> None:
        super().__init__()
        inner_dim = max(dim_minimum, dim_out // reduction)
        self.global_avg_pool = nn.AdaptiveAvgPool1d(1)
        self.layers = nn.Sequential(nn.Conv1d(dim, inner_dim, 1), nn.SiLU(), nn.Conv1d(inner_dim, dim_out, 1), nn.Sigmoid())

    def gate_from_rows(self, h: torch.Tensor, link=None) -> torch.Tensor:
        with scope("SqueezeExcite"):                       # residual.py:57
            pooled = Fn.MeanPoolFn.apply(h, link)
            l0, l2 = self.layers[0], self.layers[2]
            z = rt.small_linear(pooled, l0.weight, l0.bias)
            return rt.small_linear(z, l2.weight, l2.bias, in_act=ops.ACT_SILU, out_act=ops.ACT_SIGMOID)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = rt.to_rows(x, rt.compute_dtype(self.layers[0].weight.dtype))
        return self.gate_from_rows(h).unsqueeze(-1)


class Block(nn.Module):
    """residual.py:62-88: Conv1d(k3) -> GroupNorm(1, C) -> FiLM -> SiLU."""

    def __init__(self, dim_in: int, dim_out: int, norm: bool = True) -> None:
        super().__init__()
        self.proj = nn.Conv1d(dim_in, dim_out, 3, padding=1)
        self.norm = nn.GroupNorm(1, dim_out) if norm else nn.Identity()
        self.activation = nn.SiLU()
        self._cache = Fn.PackCache()

    def forward_rows(self, x: torch.Tensor, ss: Optional[torch.Tensor], reslink=None) -> torch.Tensor:
        extra = self.proj.adapter_inputs() if hasattr(self.proj, "adapter_inputs") else (None, None, None, None)  # lora_layers.LoraConv1d
        with scope("Residual's Block"):                    # residual.py:86
            gamma, beta = (self.norm.weight, self.norm.bias) if isinstance(self.norm, nn.GroupNorm) else (None, None)
            return Fn.BlockFn.apply(x, self.proj.weight, self.proj.bias, gamma, beta, ss, self._cache, *extra, reslink)

    def forward(self, x: torch.Tensor, scale_shift: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
        rows = rt.to_rows(x, rt.compute_dtype(self.proj.weight.dtype))
        ss = None
        if scale_shift is not None:
            scale, shift = scale_shift
            ss = torch.cat([scale.reshape(scale.shape[0], -1), shift.reshape(shift.shape[0], -1)], dim=1).float()
        return rt.to_logical(self.forward_rows(rows, ss))


class ResidualBlock(nn.Module):
    """residual.py:91-137."""

    def __init__(self, dim_in: int, dim_out: int, dim_time: Optional[int] = None, dim_cond: Optional[int] = None,
                 use_gca: bool = True) -> None:
        super().__init__()
        self.has_time_cond = dim_time is not None
        self.has_cond = dim_cond is not None
        self.mlp = (
            nn.Sequential(nn.SiLU(), nn.Linear(int(dim_time) + int(dim_cond), dim_out * 2))
            if dim_time or dim_cond
            else None
        )
        self.block1 = Block(dim_in, dim_out)
        self.block2 = Block(dim_out, dim_out)
        self.res_conv = nn.Conv1d(dim_in, dim_out, 1) if dim_in != dim_out else nn.Identity()
        self.se = GlobalContext(dim_out, dim_out) if use_gca else SqueezeExcite(dim_out, dim_out)
        self._cr = Fn.PackCache()

    def forward_rows(self, x: torch.Tensor, t: Optional[torch.Tensor], c: Optional[torch.Tensor]) -> torch.Tensor:
        ss = None
        if self.mlp is not None and (self.has_time_cond or self.has_cond):
            emb = rt.shared_cat(t, c)                      # cat(t, c): one tensor for all blocks of a forward
            lin = self.mlp[1]
            ss = rt.film_take(emb, lin)                    # (B, 2C): scale | shift -- from the UNet's one grouped launch, or
            if ss is None:                                 # (stand-alone block, recomputation, capture) its own
                ss = rt.small_linear(emb, lin.weight, lin.bias, in_act=ops.ACT_SILU)
        rlink = Fn.ResLink() if torch.is_grad_enabled() and x.requires_grad else None      # residual-path gradient -> block1's dgrad
        h = self.block1.forward_rows(x, ss, rlink)
        h = self.block2.forward_rows(h, None)
        link = Fn.GateLink() if torch.is_grad_enabled() and h.requires_grad else None
        gate = self.se.gate_from_rows(h, link)
        if isinstance(self.res_conv, nn.Identity):
            return Fn.GateResFn.apply(h, gate, x, link, rlink)
        return Fn.GateResConvFn.apply(h, gate, x, self.res_conv.weight, self.res_conv.bias, self._cr, link, rlink)

    def forward(self, x: torch.Tensor, t: Optional[torch.Tensor] = None, c: Optional[torch.Tensor] = None) -> torch.Tensor:
        rows = rt.to_rows(x, rt.compute_dtype(self.block1.proj.weight.dtype))
        try:
            return rt.to_logical(self.forward_rows(rows, t, c))
        finally:
            rt.clear_shared_cat()                          # stand-alone use: nothing else will drop the memoised cat(t, c)
