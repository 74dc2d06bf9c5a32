"""ORACLE (test infrastructure, NOT product code) -- rectified-flow wrapper restatement.

Follows ``/root/reference/osu_fusion/models/rectified_flow.py``:
  :15-16   cosmap                                        -> cosmap
  :81-111  OsuFusion.forward (flow-matching MSE)         -> training_loss   (RNG draws passed in)
  :57-79   OsuFusion.sample (odeint, method="midpoint")  -> sample

The ODE solver lives in **torchdiffeq==0.2.4** (requirements.txt:13), absent from /root/reference and from this image; its
published fixed-grid midpoint rule (y1 = y0 + dt * f(t0 + dt/2, y0 + dt/2 * f(t0, y0)) over the given time grid; rtol/atol are
ignored by fixed-grid solvers) is restated.  The reference holds no tests or vectors for it -> **parity unpinned** at that
boundary (the UNet underneath is pinned by the goldens).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .unet_oracle import UNetConfig, unet_forward

Tensor = torch.Tensor


def cosmap(t: Tensor) -> Tensor:
    return 1.0 - (1.0 / (torch.tan(math.pi / 2 * t) + 1))


def training_loss(p: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, a: Tensor, c: Tensor, noise: Tensor, times: Tensor,
                  cond_mask: Optional[Tensor] = None, cond_drop_prob: float = 0.5, orig_len: Optional[Tensor] = None,
                  mode: str = "fp32", prefix: str = "unet.") -> Tensor:
    assert x.shape[-1] == a.shape[-1], "x and a must have the same number of sequence length"
    t = cosmap(times[:, None, None])
    x_noisy = t * x + (1 - t) * noise
    flow = x - noise
    pred = unet_forward(p, cfg, x_noisy, a, times, c, cond_drop_prob=cond_drop_prob, cond_mask=cond_mask, mode=mode, prefix=prefix)
    loss = (pred - flow) ** 2
    if orig_len is not None:
        b, d, n = x.shape
        mask = (torch.arange(n)[None, :] < orig_len[:, None]).to(loss.dtype)[:, None, :].expand(b, d, n)
        return (loss * mask).sum() / mask.sum()
    return loss.mean()


@torch.no_grad()
def sample(p: Dict[str, Tensor], cfg: UNetConfig, a: Tensor, c: Tensor, x: Tensor, sampling_steps: int = 16, cond_scale: float = 2.0,
           mode: str = "fp32", prefix: str = "unet.") -> Tensor:
    b = a.shape[0]

    def f(t: float, y: Tensor) -> Tensor:
        tb = torch.full((b,), t, dtype=torch.float32)
        out = unet_forward(p, cfg, y, a, tb, c, cond_drop_prob=0.0, mode=mode, prefix=prefix)
        if cond_scale != 1.0:
            null = unet_forward(p, cfg, y, a, tb, c, cond_drop_prob=1.0, mode=mode, prefix=prefix)
            out = null + (out - null) * cond_scale
        return out

    times = torch.linspace(0.0, 1.0, sampling_steps)
    for t0, t1 in zip(times[:-1].tolist(), times[1:].tolist()):
        dt = t1 - t0
        k1 = f(t0, x)
        k2 = f(t0 + 0.5 * dt, x + 0.5 * dt * k1)
        x = x + dt * k2
    return x
