"""ORACLE (test infrastructure, NOT product code) -- DDPM/DDIM wrapper restatement.

Follows ``/root/reference/osu_fusion/models/diffusion.py``:
  :48-51   DDIMScheduler(num_train_timesteps=1000, beta_schedule="linear")  -> ddim_alphas_cumprod
  :96      scheduler.add_noise                                              -> add_noise
  :71      scheduler.set_timesteps                                          -> ddim_timesteps
  :75      scheduler.step(...).prev_sample                                  -> ddim_step
  :79-111  OsuFusion.forward (eps-prediction MSE, optional length mask)     -> training_loss
  :59-77   OsuFusion.sample (DDIM loop with classifier-free guidance)       -> sample

The scheduler arithmetic lives in a third-party dependency that is absent from /root/reference
and from this image: **diffusers==0.29.2** (requirements.txt:3).  Its published DDIM algorithm
(eta=0, epsilon prediction, "leading" timestep spacing, clip_sample=True, set_alpha_to_one=True)
is restated here; the reference holds no tests or golden vectors for it, so this boundary is
**parity unpinned** beyond the known-answer constants recorded in SURVEY.md §8a row 15
(checked in tests/test_oracle_golden.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .unet_oracle import UNetConfig, unet_forward

Tensor = torch.Tensor


def ddim_alphas_cumprod(train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02) -> Tensor:
    betas = torch.linspace(beta_start, beta_end, train_timesteps, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(x: Tensor, noise: Tensor, timesteps: Tensor, acp: Tensor) -> Tensor:
    sa = acp[timesteps].sqrt().view(-1, *([1] * (x.ndim - 1)))
    sb = (1.0 - acp[timesteps]).sqrt().view(-1, *([1] * (x.ndim - 1)))
    return sa * x + sb * noise


def ddim_timesteps(sampling_steps: int, train_timesteps: int = 1000) -> Tensor:
    ratio = train_timesteps // sampling_steps                          # "leading" spacing, steps_offset=0
    return (torch.arange(sampling_steps) * ratio).flip(0).to(torch.int64)


def ddim_step(pred_eps: Tensor, t: int, x: Tensor, acp: Tensor, sampling_steps: int, train_timesteps: int = 1000) -> Tensor:
    prev_t = t - train_timesteps // sampling_steps
    a_t = acp[t]
    a_prev = acp[prev_t] if prev_t >= 0 else torch.tensor(1.0)       # set_alpha_to_one=True
    x0 = (x - (1.0 - a_t).sqrt() * pred_eps) / a_t.sqrt()
    x0 = x0.clamp(-1.0, 1.0)                                           # clip_sample=True, range 1.0
    return a_prev.sqrt() * x0 + (1.0 - a_prev).sqrt() * pred_eps       # eta = 0, un-clipped eps direction


def training_loss(p: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, a: Tensor, c: Tensor, noise: Tensor,
                  timesteps: Tensor, cond_mask: Optional[Tensor] = None, cond_drop_prob: float = 0.5,
                  orig_len: Optional[Tensor] = None, mode: str = "fp32", prefix: str = "unet.") -> Tensor:
    """diffusion.py:79-111 with the RNG draws (noise, timesteps, cond mask) passed in explicitly."""
    assert x.shape[-1] == a.shape[-1], "x and a must have the same number of sequence length"
    acp = ddim_alphas_cumprod()
    x_noisy = add_noise(x, noise, timesteps, acp)
    pred = unet_forward(p, cfg, x_noisy, a, timesteps, c, cond_drop_prob=cond_drop_prob, cond_mask=cond_mask,
                        mode=mode, prefix=prefix)
    loss = (pred - noise) ** 2
    if orig_len is not None:
        b, d, n = x.shape
        mask = (torch.arange(n)[None, :] < orig_len[:, None]).to(loss.dtype)[:, None, :].expand(b, d, n)
        return (loss * mask).sum() / mask.sum()
    return loss.mean()


@torch.no_grad()
def sample(p: Dict[str, Tensor], cfg: UNetConfig, a: Tensor, c: Tensor, x: Tensor, sampling_steps: int = 35,
           cond_scale: float = 7.0, mode: str = "fp32", prefix: str = "unet.") -> Tensor:
    acp = ddim_alphas_cumprod()
    b = a.shape[0]
    for t in ddim_timesteps(sampling_steps).tolist():
        tb = torch.full((b,), t, dtype=torch.int64)
        pred = unet_forward(p, cfg, x, a, tb, c, cond_drop_prob=0.0, mode=mode, prefix=prefix)
        if cond_scale != 1.0:                                          # unet.py:458-465
            null = unet_forward(p, cfg, x, a, tb, c, cond_drop_prob=1.0, mode=mode, prefix=prefix)
            pred = null + (pred - null) * cond_scale
        x = ddim_step(pred, t, x, acp, sampling_steps)
    return x
