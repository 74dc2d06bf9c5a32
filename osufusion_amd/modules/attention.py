"""HIP-backed mirror of osu_fusion/modules/attention.py.

``RotaryPositionEmbedding`` and ``Attend`` keep the reference's call signatures on (B, H, N, D) tensors; the UNet's
own Attention block does not go through them (it uses the fused LN -> QKV -> RoPE -> flash path of functional.py),
they exist so code written against the reference's attention.py keeps working on the HIP kernels.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops
from .. import runtime as rt


class RotaryPositionEmbedding(nn.Module):
    """attention.py:15-58: positions rescaled by scale_base / seq_len, half-split rotation."""

    def __init__(self, dim: int, theta: int = 10000, scale_base: int = 4096) -> None:
        super().__init__()
        self.dim, self.theta, self.scale_base = dim, theta, scale_base
        inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)

    def forward(self, q: torch.Tensor, k: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        rt.require_gpu(q)
        if self.dim % 16:
            raise NotImplementedError("the HIP RoPE kernel rotates head dims that are multiples of 16")
        cos, sin = Fn.rope_tables(q.shape[-2], self.dim, self.scale_base, q.device, float(self.theta))
        return _rope_bhnd(q, cos, sin), _rope_bhnd(k, cos, sin)


def _rope_bhnd(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    B, H, N, D = x.shape
    rows = x.permute(0, 2, 1, 3).reshape(B, N, H * D).contiguous()
    if rows.dtype not in (torch.float32, torch.bfloat16):
        rows = rows.float()
    out = ops.rope_cast(rows, cos, sin, N, H, H, D)                     # bf16, as Attend would cast it next
    return out.view(B, N, H, D).permute(0, 2, 1, 3).to(x.dtype)


class Attend(nn.Module):
    """attention.py:61-101: q, k, v -> bf16, softmax(q k^T / sqrt(d)) v, back to the input dtype.  Inference only here."""

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        rt.require_gpu(q)
        B, H, N, D = q.shape
        if D not in (16, 32, 64, 128):
            raise NotImplementedError("the HIP attention kernels cover head dims 16, 32, 64 and 128")
        if k.shape[1] not in (1, H) or v.shape[1] != k.shape[1]:
            raise ValueError(f"k / v must carry 1 or {H} heads (got {k.shape[1]} / {v.shape[1]})")
        G = k.shape[1]
        rows = lambda t: t.permute(0, 2, 1, 3).reshape(B, N, t.shape[1] * D)
        if attn_mask is not None:
            # attention.py:90-98: the mask is cast to the q/k/v dtype (bf16) and goes to SDPA as an ADDITIVE bias, whatever its dtype was
            m4 = attn_mask.to(torch.bfloat16)
            while m4.dim() < 4:
                m4 = m4.unsqueeze(0)
            m4 = m4.expand(B, H, N, N)                       # a view: broadcast dimensions keep stride 0
            outs = []
            for g in range(G):                               # per K/V head: its query heads are g, g + G, ... only when G == H or 1 here
                qs = q if G == 1 else q[:, g:g + 1]
                ms = m4 if G == 1 else m4[:, g:g + 1]
                Hq = qs.shape[1]
                qkv = torch.cat([rows(qs), rows(k[:, g:g + 1]), rows(v[:, g:g + 1])], dim=-1).to(torch.bfloat16).contiguous()
                o = ops.mqa_fwd_masked(qkv, ms, B, N, Hq, D, torch.bfloat16, D ** -0.5)
                outs.append(o.view(B, N, Hq, D))
            return torch.cat(outs, dim=2).permute(0, 2, 1, 3).to(v.dtype)
        if G != 1 and torch.equal(k[:, :1].expand_as(k), k) and torch.equal(v[:, :1].expand_as(v), v):
            G, k, v = 1, k[:, :1], v[:, :1]                 # one K/V head repeated (what the UNet's Attention hands over): one launch
        qkv = torch.cat([rows(q), rows(k), rows(v)], dim=-1).to(torch.bfloat16).contiguous()
        o, _ = ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5, kv_heads=G)   # G = H: every query head has its own K/V head
        return o.view(B, N, H, D).permute(0, 2, 1, 3).to(v.dtype)
