"""HIP-backed mirror of osu_fusion/models/rectified_flow.py: ``OsuFusion`` (flow-matching variant) with the same ctor,
``forward`` (training loss), ``sample`` (fixed-grid midpoint ODE + classifier-free guidance), ``set_full_bf16`` and the
``unet`` / ``sample_timesteps`` attributes.

torchdiffeq==0.2.4's ``odeint(..., method="midpoint")`` (absent from this image) is restated: over ``linspace(0, 1, S)`` each
interval does ``k1 = f(t0, y); k2 = f(t0 + dt/2, y + dt/2 * k1); y += dt * k2`` (rtol / atol are ignored by fixed-grid solvers).
"""
import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: N812

from .. import ops
from .. import runtime as rt
from ..modules.unet import UNet
from .diffusion import AUDIO_DIM, CONTEXT_DIM, TOTAL_DIM, _MSEFn


def cosmap(t: torch.Tensor) -> torch.Tensor:
    """rectified_flow.py:15-16."""
    return 1.0 - (1.0 / (torch.tan(math.pi / 2 * t) + 1))


class OsuFusion(nn.Module):
    def __init__(self, dim_h: int, dim_h_mult: Tuple[int] = (1, 2, 3, 4), num_layer_blocks: Tuple[int] = (3, 3, 3, 3),
                 num_middle_transformers: int = 3, cross_embed_kernel_sizes: Tuple[int] = (3, 7, 15), attn_dim_head: int = 64,
                 attn_heads: int = 16, attn_kv_heads: int = 1, attn_context_len: int = 4096, cond_drop_prob: float = 0.5,
                 sampling_timesteps: int = 16) -> None:
        super().__init__()
        self.unet = UNet(dim_in_x=TOTAL_DIM, dim_in_a=AUDIO_DIM, dim_in_c=CONTEXT_DIM, dim_h=dim_h, dim_h_mult=dim_h_mult,
                         num_layer_blocks=num_layer_blocks, num_middle_transformers=num_middle_transformers,
                         cross_embed_kernel_sizes=cross_embed_kernel_sizes, attn_dim_head=attn_dim_head, attn_heads=attn_heads,
                         attn_kv_heads=attn_kv_heads, attn_context_len=attn_context_len)
        self.sample_timesteps = sampling_timesteps
        self.cond_drop_prob = cond_drop_prob
        self._full_bf16 = False

    def set_full_bf16(self) -> None:
        """Keeps fp32 master weights; all kernels compute in bf16 (see models/diffusion.py)."""
        self._full_bf16 = True

    def _dtype_ctx(self):
        return rt.forced_compute_dtype(torch.bfloat16 if self._full_bf16 else None)

    @torch.inference_mode()
    def sample(self, a: torch.Tensor, c: torch.Tensor, x: Optional[torch.Tensor] = None, cond_scale: float = 2.0) -> torch.Tensor:
        """rectified_flow.py:57-79.  Same restructuring as the DDIM sampler: audio code computed once, conditional + null
        branches as one batch of 2B; 2 * (S - 1) UNet evaluations.  Bit-reproducible (ops.reproducible_mode), as the DDIM sampler."""
        with ops.reproducible_mode(True):
            return self._sample(a, c, x, cond_scale)

    def _sample(self, a, c, x, cond_scale):
        (b, _, n), device = a.shape, a.device
        rt.require_gpu(a)
        if x is None:
            x = torch.randn((b, TOTAL_DIM, n), device=device)
        x = x.float().contiguous()
        unet = self.unet
        depth = len(unet.down_layers)
        pad_len = (2 ** depth - (n % (2 ** depth))) % (2 ** depth)
        cfg = cond_scale != 1.0
        ones = torch.ones(b, dtype=torch.float32, device=device)
        with self._dtype_ctx():
            dtype = rt.compute_dtype(unet.final_conv.weight.dtype)
            a_rows = unet.encode_audio(a, pad_len, dtype)
            keep = torch.ones(b, dtype=torch.bool, device=device)
            if cfg:
                a_rows = torch.cat([a_rows, a_rows], 0)
                ce = unet.embed_cond(torch.cat([c, c], 0), torch.cat([keep, ~keep], 0))
            else:
                ce = unet.embed_cond(c, keep)

            def f(t: float, y: torch.Tensor) -> torch.Tensor:
                yin = torch.cat([y, y], 0) if cfg else y
                tb = torch.full((yin.shape[0],), t, dtype=torch.float32, device=device)
                y_rows = unet.init_x.forward_rows(F.pad(yin, (0, pad_len), value=-1.0), dtype)
                out = unet.denoise_rows(y_rows, a_rows, unet.embed_time(tb), ce)[:, :, :n].contiguous()
                if cfg:                                    # null + (cond - null) * s  ==  (1 - s) * null + s * cond
                    out = ops.axpby_rows(out[b:].contiguous(), out[:b].contiguous(), ones * (1.0 - cond_scale), ones * cond_scale)
                return out

            times = torch.linspace(0.0, 1.0, self.sample_timesteps).tolist()
            for t0, t1 in zip(times[:-1], times[1:]):
                dt = t1 - t0
                k1 = f(t0, x)
                k2 = f(t0 + 0.5 * dt, ops.axpby_rows(x, k1, ones, ones * (0.5 * dt)))
                x = ops.axpby_rows(x, k2, ones, ones * dt)
        return x

    def forward(self, x: torch.Tensor, a: torch.Tensor, c: torch.Tensor, orig_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert x.shape[-1] == a.shape[-1], "x and a must have the same number of sequence length"
        rt.require_gpu(x)
        noise = torch.randn_like(x, device=x.device)
        times = torch.rand(x.shape[0], device=x.device)
        return self.loss_with(x, a, c, noise, times, orig_len)

    def loss_with(self, x, a, c, noise, times, orig_len=None, cond_drop_prob: Optional[float] = None) -> torch.Tensor:
        """forward() with the RNG draws passed in (parity tests, benchmarks)."""
        p = self.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
        x, noise = x.float().contiguous(), noise.float().contiguous()
        t = cosmap(times.float())
        ones = torch.ones_like(t)
        with self._dtype_ctx():
            x_noisy = ops.axpby_rows(x, noise, t.contiguous(), (1 - t).contiguous())
            flow = ops.axpby_rows(x, noise, ones, -ones)
            pred = self.unet(x_noisy, a, times.float(), c, cond_drop_prob=p)
        return _MSEFn.apply(pred, flow, orig_len)
