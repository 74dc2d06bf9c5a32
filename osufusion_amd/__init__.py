"""osufusion_amd: MI355X-native (gfx950) implementation of OsuFusion's diffusion-denoising hot path.

Drop-in for ``osu_fusion.modules.{unet,residual,attention,utils}`` and ``osu_fusion.models.diffusion`` (same class names,
constructor signatures, tensor API and state_dict keys); compute runs in hand-written HIP kernels (libosuf_hip.so).
"""
from .runtime import compute_dtype, forced_compute_dtype, set_compute_dtype  # noqa: F401
from .ops import set_f32_matmul  # noqa: F401

__all__ = ["set_compute_dtype", "compute_dtype", "forced_compute_dtype", "set_f32_matmul"]
