"""HIP-backed mirror of osu_fusion/modules/unet.py: same classes, ctor signatures, state_dict keys and quirks.

Public ``forward`` methods keep the reference's (B, C, L) / (B, N, C) tensor API.  Internally every module also has a
``forward_rows`` that works on channels-last rows (B, L, C) so a chain of modules never changes layout.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: N812
from torch.utils.checkpoint import checkpoint

from .. import functional as Fn
from .. import ops
from .. import runtime as rt
from ..tracing import scope
from .residual import ResidualBlock
from .utils import prob_mask_like


def zero_init(module: nn.Module) -> nn.Module:
    """unet.py:18-23: weight and bias of `module` set to zero (the reference starts final_conv this way, unet.py:354)."""
    with torch.no_grad():
        for tensor in (module.weight, module.bias):
            if tensor is not None:
                tensor.zero_()
    return module


class SinusoidalPositionEmbedding(nn.Module):
    """unet.py:26-39 (tiny: B x dim; plain torch on device)."""

    def __init__(self, dim: int, theta: int = 10000) -> None:
        super().__init__()
        self.dim = dim
        self.theta = theta

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,) timesteps -> (B, dim) = [sin(t f_i) | cos(t f_i)], f_i = theta^(-i / (dim/2 - 1)): the divisor is half_dim - 1
        (unet.py:35), and an int64 t is promoted to fp32 by the product."""
        n_freq = self.dim // 2
        freqs = torch.exp(torch.arange(n_freq, device=x.device) * (-math.log(self.theta) / (n_freq - 1)))
        angles = x.unsqueeze(1) * freqs.unsqueeze(0)
        return torch.cat((torch.sin(angles), torch.cos(angles)), dim=1)


class CrossEmbedLayer(nn.Module):
    """unet.py:42-58: multi-kernel conv stem, outputs concatenated on channels.

    All kernel sizes are merged into ONE tap-GEMM with zero-padded taps; for few input channels (the 6-channel
    beatmap signal) the taps are folded into K by an im2col layout change so the GEMM has K = 96 instead of 15 x 8.
    """

    def __init__(self, dim: int, dim_out: int, kernel_sizes: Tuple[int]) -> None:
        super().__init__()
        kernel_sizes = sorted(kernel_sizes)
        num_scales = len(kernel_sizes)
        dim_scales = [int(dim / (2 ** i)) for i in range(1, num_scales)]
        dim_scales = [*dim_scales, dim_out - sum(dim_scales)]
        self.convs = nn.ModuleList(
            [nn.Conv1d(dim, dim_scale, kernel, padding=kernel // 2) for kernel, dim_scale in zip(kernel_sizes, dim_scales)])
        self.kernel_sizes = kernel_sizes
        self.dim, self.dim_out = dim, dim_out
        self._cache = Fn.PackCache()

    def _merged(self) -> Tuple[torch.Tensor, torch.Tensor]:
        kmax = self.kernel_sizes[-1]
        ws = []
        for conv in self.convs:
            k = conv.kernel_size[0]
            off = kmax // 2 - k // 2
            ws.append(F.pad(conv.weight, (off, kmax - k - off)))
        return torch.cat(ws, 0), torch.cat([conv.bias for conv in self.convs], 0)      # (dim_out, dim, kmax), (dim_out,)

    def forward_rows(self, x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """x: (B, dim, L) fp32 channel-major -> rows (B, L, dim_out)."""
        rt.require_gpu(x)
        kmax = self.kernel_sizes[-1]
        w, b = self._merged()
        vp = ("stem", *[c.weight for c in self.convs])
        if self.dim * kmax <= 128:
            width = (self.dim * kmax + 7) // 8 * 8
            rows = Fn.RowsFromNCLFn.apply(x, dtype, width, kmax)
            w2 = F.pad(w.permute(0, 2, 1).reshape(self.dim_out, kmax * self.dim), (0, width - kmax * self.dim))
            return Fn.ConvFn.apply(rows, w2, b, self._cache, "same", vp)
        if self.dim % 8:
            raise ValueError("CrossEmbedLayer: input channels must be a multiple of 8 (or dim * kmax <= 128)")
        rows = rt.to_rows(x, dtype)
        return Fn.ConvFn.apply(rows, w, b, self._cache, "same", vp)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return rt.to_logical(self.forward_rows(x, rt.compute_dtype(self.convs[0].weight.dtype)))


class Upsample(nn.Module):
    """unet.py:61-74: nearest x2 then Conv1d k3 (the upsample is folded into the GEMM's row map)."""

    def __init__(self, dim_in: int, dim_out: int) -> None:
        super().__init__()
        self.conv = nn.Conv1d(dim_in, dim_out, 3, padding=1)
        self._cache = Fn.PackCache()

    def forward_rows(self, x: torch.Tensor) -> torch.Tensor:
        with scope("Upsample"):                            # the reference's record_function scope (unet.py:72), as a roctx range
            return Fn.ConvFn.apply(x, self.conv.weight, self.conv.bias, self._cache, "up")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return rt.to_logical(self.forward_rows(rt.to_rows(x, rt.compute_dtype(self.conv.weight.dtype))))


class Downsample(nn.Module):
    """unet.py:77-92: reflect-pad right by 1, Conv1d k3 stride 2 (pad folded into the GEMM's row map)."""

    def __init__(self, dim_in: int, dim_out: int) -> None:
        super().__init__()
        self.conv = nn.Conv1d(dim_in, dim_out, 3, stride=2, padding=0)
        self._cache = Fn.PackCache()

    def forward_rows(self, x: torch.Tensor) -> torch.Tensor:
        with scope("Downsample"):                          # unet.py:90
            return Fn.ConvFn.apply(x, self.conv.weight, self.conv.bias, self._cache, "down")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return rt.to_logical(self.forward_rows(rt.to_rows(x, rt.compute_dtype(self.conv.weight.dtype))))


class Parallel(nn.Module):
    """unet.py:95-101, as the UNet uses it: Conv1d(k3, pad 1) + Conv1d(k1) summed == one k3 conv with the k1 weights
    added to its centre tap (exact up to fp32 summation order)."""

    def __init__(self, *fns: nn.Module) -> None:
        super().__init__()
        self.fns = nn.ModuleList(fns)
        ok = (len(fns) == 2 and all(isinstance(f, nn.Conv1d) for f in fns) and fns[0].kernel_size == (3,) and fns[1].kernel_size == (1,)
              and fns[0].padding == (1,) and fns[0].stride == (1,))
        if not ok:
            raise NotImplementedError("Parallel is implemented for (Conv1d k3 pad1, Conv1d k1), the only form the UNet builds")
        self._cache = Fn.PackCache()

    def forward_rows(self, x: torch.Tensor) -> torch.Tensor:
        c3, c1 = self.fns[0], self.fns[1]
        w = c3.weight + F.pad(c1.weight, (1, 1))
        return Fn.ConvFn.apply(x, w, c3.bias + c1.bias, self._cache, "same", ("par", c3.weight, c1.weight))

    def forward(self, x: torch.Tensor, *args: List, **kwargs: Dict) -> torch.Tensor:
        return rt.to_logical(self.forward_rows(rt.to_rows(x, rt.compute_dtype(self.fns[0].weight.dtype))))


class Attention(nn.Module):
    """unet.py:104-146.  forward(x: (B, N, C)) -> (B, N, C); the residual is the LayerNorm'ed x (reference quirk)."""

    def __init__(self, dim_in: int, dim_head: int, heads: int, kv_heads: int, context_len: int = 4096) -> None:
        super().__init__()
        self.heads, self.kv_heads, self.dim_head, self.context_len = heads, kv_heads, dim_head, context_len
        self.norm = nn.LayerNorm(dim_in)
        self.to_q = nn.Linear(dim_in, dim_head * heads, bias=False)
        self.to_kv = nn.Linear(dim_in, dim_head * kv_heads * 2, bias=False)
        self.to_out = nn.Linear(dim_head * heads, dim_in)
        self._cache = Fn.PackCache()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rt.require_gpu(x)
        if self.dim_head not in (16, 32, 64, 128):
            raise NotImplementedError(f"the HIP attention kernels cover head dims 16, 32, 64 and 128 (got dim_head={self.dim_head})")
        if self.heads % self.kv_heads:
            raise ValueError(f"heads ({self.heads}) must be a multiple of kv_heads ({self.kv_heads})")
        x = rt.cast_rows(x.contiguous(), rt.compute_dtype(self.to_q.weight.dtype))
        if self.kv_heads != 1:
            return self._forward_gqa(x)
        extra = (None,) * 8
        if hasattr(self.to_q, "adapter_inputs") or hasattr(self.to_kv, "adapter_inputs"):                # lora_layers.LoraLinear
            none4 = (None, None, None, None)
            aq, qa, qb, qm = self.to_q.adapter_inputs() if hasattr(self.to_q, "adapter_inputs") else none4
            akv, ka, kb, km = self.to_kv.adapter_inputs() if hasattr(self.to_kv, "adapter_inputs") else none4
            extra = (aq, akv, qa, qb, qm, ka, kb, km)
        with scope("Attention"):                           # unet.py:144
            return Fn.AttentionFn.apply(x, self.norm.weight, self.norm.bias, self.to_q.weight, self.to_kv.weight, self.to_out.weight,
                                        self.to_out.bias, self._cache, self.heads, self.dim_head, self.context_len, *extra, 1, None,
                                        torch.is_grad_enabled())       # (grad mode is off inside Function.forward: tell it whether a backward can follow)


    def _forward_gqa(self, x: torch.Tensor) -> torch.Tensor:
        """kv_heads = G > 1 (unet.py:132-135: `repeat(t, "b h n d -> b (r h) n d")`, i.e. query head j reads K/V head j mod G).  The
        kernels take one K/V head per launch, so the query heads are regrouped GROUP-MAJOR -- heads {g, g + G, g + 2G, ...} become one
        contiguous block next to "their" K/V head -- by permuting the head blocks of to_q's rows and of to_out's input columns (views
        of the parameters: autograd carries the gradients back through the permutation), and AttentionFn runs G launches per pass."""
        if hasattr(self.to_q, "adapter_inputs") or hasattr(self.to_kv, "adapter_inputs"):
            raise NotImplementedError("LoRA / DoRA adapters on a grouped-query Attention (kv_heads > 1) have no HIP path")
        H, G, D = self.heads, self.kv_heads, self.dim_head
        r, C = H // G, self.to_q.weight.shape[1]
        wq = self.to_q.weight.view(r, G, D, C).permute(1, 0, 2, 3).reshape(H * D, C)
        wo = self.to_out.weight.view(-1, r, G, D).permute(0, 2, 1, 3).reshape(-1, H * D)
        with scope("Attention"):                           # unet.py:144
            return Fn.AttentionFn.apply(x, self.norm.weight, self.norm.bias, wq, self.to_kv.weight, wo, self.to_out.bias, self._cache,
                                        H, D, self.context_len, None, None, None, None, None, None, None, None, G,
                                        (self.to_q.weight, self.to_out.weight), torch.is_grad_enabled())


class FeedForward(nn.Sequential):
    """unet.py:149-156 (parameter container; TransformerBlock fuses `ff(x) + x`)."""

    def __init__(self, dim: int, dim_mult: int = 2) -> None:
        inner_dim = dim * dim_mult
        super().__init__(nn.Linear(dim, inner_dim), nn.SiLU(), nn.Linear(inner_dim, dim))


class TransformerBlock(nn.Module):
    """unet.py:159-183."""

    def __init__(self, dim: int, ff_mult: int = 2, attn_dim_head: int = 64, attn_heads: int = 16, attn_kv_heads: int = 1,
                 attn_context_len: int = 4096) -> None:
        super().__init__()
        self.attn = Attention(dim, attn_dim_head, attn_heads, attn_kv_heads, attn_context_len)
        self.ff = FeedForward(dim, ff_mult)
        self._cache = Fn.PackCache()

    def forward_rows(self, x: torch.Tensor) -> torch.Tensor:
        x = self.attn(x)
        return Fn.FeedForwardFn.apply(x, self.ff[0].weight, self.ff[0].bias, self.ff[2].weight, self.ff[2].bias, self._cache)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return rt.to_logical(self.forward_rows(rt.to_rows(x, rt.compute_dtype(self.ff[0].weight.dtype))))


class UNetBlock(nn.Module):
    """unet.py:186-263."""

    def __init__(self, dim_in: int, dim_out: int, dim_time: Optional[int], dim_cond: Optional[int], layer_idx: int, num_layers: int,
                 num_blocks: int, down_block: bool, attn_dim_head: int, attn_heads: int, attn_kv_heads: int, attn_context_len: int) -> None:
        super().__init__()
        self.init_resnet = ResidualBlock(dim_in if down_block else dim_in + dim_out, dim_in, dim_time, dim_cond)
        self.resnets = nn.ModuleList([ResidualBlock(dim_in, dim_in, dim_time, dim_cond) for _ in range(num_blocks)])
        self.transformers = nn.ModuleList([
            TransformerBlock(dim_in, attn_dim_head=attn_dim_head, attn_heads=attn_heads, attn_kv_heads=attn_kv_heads,
                             attn_context_len=attn_context_len) for _ in range(num_blocks)])
        last = layer_idx >= (num_layers - 1)
        if last:
            self.sampler = Parallel(nn.Conv1d(dim_in, dim_out, 3, padding=1), nn.Conv1d(dim_in, dim_out, 1))
        else:
            self.sampler = Downsample(dim_in, dim_out) if down_block else Upsample(dim_in, dim_out)
        self.gradient_checkpointing = False

    def forward_body(self, x: torch.Tensor, t: Optional[torch.Tensor] = None, c: Optional[torch.Tensor] = None):
        """rows in, (sampled rows, pre-sample rows) out."""
        x = self.init_resnet.forward_rows(x, t, c)
        for resnet, transformer in zip(self.resnets, self.transformers):
            x = resnet.forward_rows(x, t, c)
            x = transformer.forward_rows(x)
        return self.sampler.forward_rows(x), x

    def forward_rows(self, x, t=None, c=None):
        if self.training and self.gradient_checkpointing:
            return checkpoint(self.forward_body, x, t, c, use_reentrant=True)
        return self.forward_body(x, t, c)

    def forward(self, x: torch.Tensor, t: Optional[torch.Tensor] = None, c: Optional[torch.Tensor] = None):
        rows = rt.to_rows(x, rt.compute_dtype(self.init_resnet.block1.proj.weight.dtype))
        y, skip = self.forward_rows(rows, t, c)
        return rt.to_logical(y), rt.to_logical(skip)


class AudioEncoder(nn.Module):
    """unet.py:266-318."""

    def __init__(self, dim_in: int, dim_h: int, dim_h_mult: Tuple[int] = (1, 2, 3, 4), num_layer_blocks: Tuple[int] = (3, 3, 3, 3),
                 cross_embed_kernel_sizes: Tuple[int] = (3, 7, 15), attn_dim_head: int = 64, attn_heads: int = 16,
                 attn_kv_heads: int = 1, attn_context_len: int = 4096) -> None:
        super().__init__()
        self.dim_h = dim_h
        self.dim_emb = dim_h * 4
        self.attn_context_len = attn_context_len
        self.init_conv = CrossEmbedLayer(dim_in, dim_h, cross_embed_kernel_sizes)
        dims_h = (dim_h, *[dim_h * mult for mult in dim_h_mult])
        in_out = tuple(zip(dims_h[:-1], dims_h[1:]))
        n_layers = len(in_out)
        self.layers = nn.ModuleList([
            UNetBlock(di, do, None, None, i, n_layers, num_layer_blocks[i], True, attn_dim_head, attn_heads, attn_kv_heads,
                      attn_context_len // (2 ** i)) for i, (di, do) in enumerate(in_out)])

    def forward_rows(self, a: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        x = self.init_conv.forward_rows(a, dtype)
        for layer in self.layers:
            x, _ = layer.forward_rows(x)
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return rt.to_logical(self.forward_rows(x, rt.compute_dtype(self.init_conv.convs[0].weight.dtype)))


class UNet(nn.Module):
    """unet.py:321-513."""

    def __init__(self, dim_in_x: int, dim_in_a: int, dim_in_c: int, dim_h: int, dim_h_mult: Tuple[int] = (1, 2, 3, 4),
                 num_layer_blocks: Tuple[int] = (3, 3, 3, 3), num_middle_transformers: int = 3,
                 cross_embed_kernel_sizes: Tuple[int] = (3, 7, 15), attn_dim_head: int = 64, attn_heads: int = 16,
                 attn_kv_heads: int = 1, attn_context_len: int = 4096) -> None:
        super().__init__()
        self.dim_h = dim_h
        self.dim_emb = dim_h * 4
        self.dim_in_x = dim_in_x
        self.attn_context_len = attn_context_len

        self.init_x = CrossEmbedLayer(dim_in_x, dim_h, cross_embed_kernel_sizes)
        self.audio_encoder = AudioEncoder(dim_in_a, dim_h, dim_h_mult=dim_h_mult, num_layer_blocks=num_layer_blocks,
                                          cross_embed_kernel_sizes=cross_embed_kernel_sizes, attn_dim_head=attn_dim_head,
                                          attn_heads=attn_heads, attn_kv_heads=attn_kv_heads)
        self.final_resnet = ResidualBlock(dim_h * 2, dim_h, self.dim_emb, self.dim_emb)
        self.final_conv = zero_init(nn.Conv1d(dim_h, dim_in_x, 1))

        self.time_mlp = nn.Sequential(SinusoidalPositionEmbedding(self.dim_emb), nn.Linear(self.dim_emb, self.dim_emb), nn.SiLU(),
                                      nn.Linear(self.dim_emb, self.dim_emb))
        self.cond_mlp = nn.Sequential(nn.Linear(dim_in_c, self.dim_emb), nn.SiLU(), nn.Linear(self.dim_emb, self.dim_emb))
        self.null_cond = nn.Parameter(torch.randn(self.dim_emb))

        dims_h = (dim_h, *[dim_h * mult for mult in dim_h_mult])
        in_out = tuple(zip(dims_h[:-1], dims_h[1:]))
        n_layers = len(in_out)
        self.down_layers = nn.ModuleList([
            UNetBlock(di, do, self.dim_emb, self.dim_emb, i, n_layers, num_layer_blocks[i], True, attn_dim_head, attn_heads,
                      attn_kv_heads, attn_context_len // (2 ** i)) for i, (di, do) in enumerate(in_out)])

        self.middle_resnet1 = ResidualBlock(dims_h[-1] * 2, dims_h[-1], self.dim_emb, self.dim_emb)
        self.middle_transformer = nn.ModuleList([
            TransformerBlock(dims_h[-1], attn_dim_head=attn_dim_head, attn_heads=attn_heads, attn_kv_heads=attn_kv_heads,
                             attn_context_len=attn_context_len // (2 ** (n_layers - 1))) for _ in range(num_middle_transformers)])
        self.middle_resnet2 = ResidualBlock(dims_h[-1], dims_h[-1], self.dim_emb, self.dim_emb)

        rev = tuple(reversed(in_out))
        rblocks = tuple(reversed(num_layer_blocks))
        self.up_layers = nn.ModuleList([
            UNetBlock(di, do, self.dim_emb, self.dim_emb, i, n_layers, rblocks[i], False, attn_dim_head, attn_heads, attn_kv_heads,
                      attn_context_len // (2 ** (n_layers - i - 1))) for i, (do, di) in enumerate(rev)])
        self._cf = Fn.PackCache()
        self._film = None                                   # cached list of the FiLM Linears (see _film_linears)

    def set_gradient_checkpointing(self, value: bool) -> None:
        """unet.py:452-456: switch reentrant activation checkpointing of every UNetBlock (the reference prints one line per
        block; the same line is kept so logs of the two trainers match)."""
        blocks = [(n, m) for n, m in self.named_modules() if isinstance(m, UNetBlock)]
        for name, block in blocks:
            block.gradient_checkpointing = value
            print(f"Set gradient checkpointing to {value} for {name}")

    def forward_with_cond_scale(self, *args: List, cond_scale: float = 1.0, **kwargs: Dict) -> torch.Tensor:
        """unet.py:458-465, classifier-free guidance: null + (cond - null) * cond_scale with the null branch evaluated at
        cond_drop_prob=1 (a second full forward, as in the reference; OsuFusion.sample batches the two instead)."""
        cond = self.forward(*args, **kwargs)
        if cond_scale == 1.0:
            return cond
        null = self.forward(*args, **{**kwargs, "cond_drop_prob": 1.0})
        return torch.lerp(null, cond, float(cond_scale))

    # -- pieces reused by the sampler (audio code cached across DDIM steps) ------------------------------------
    def encode_audio(self, a: torch.Tensor, pad_len: int, dtype: torch.dtype) -> torch.Tensor:
        a = F.pad(a.float(), (0, pad_len), value=-23.0)
        return self.audio_encoder.forward_rows(a, dtype)

    def embed_time(self, t: torch.Tensor) -> torch.Tensor:
        e = self.time_mlp[0](t)
        e = rt.small_linear(e, self.time_mlp[1].weight, self.time_mlp[1].bias)
        return rt.small_linear(e, self.time_mlp[3].weight, self.time_mlp[3].bias, in_act=ops.ACT_SILU)

    def embed_cond(self, c: torch.Tensor, cond_mask: torch.Tensor) -> torch.Tensor:
        e = rt.small_linear(c.float(), self.cond_mlp[0].weight, self.cond_mlp[0].bias)
        e = rt.small_linear(e, self.cond_mlp[2].weight, self.cond_mlp[2].bias, in_act=ops.ACT_SILU)
        null = self.null_cond.float()[None, :].expand(e.shape[0], -1)
        return torch.where(cond_mask[:, None], e, null)

    def denoise_rows(self, x_rows: torch.Tensor, a_rows: torch.Tensor, t: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
        """x_rows: stem output (B, L, dim_h); a_rows: audio code (B, L/2^(depth-1), 4*dim_h) -> (B, dim_in_x, L) fp32."""
        try:
            return self._denoise_rows(x_rows, a_rows, t, c)
        finally:
            rt.clear_shared_cat()                          # the 35 FiLM projections shared one cat(t, c)

    def _film_linears(self):
        """The Linear of every conditioned ResidualBlock's Sequential(SiLU, Linear), in module order (35 at the headline config)."""
        if self._film is None:
            self._film = [m.mlp[1] for m in self.modules() if isinstance(m, ResidualBlock) and m.mlp is not None and isinstance(m.mlp[1], nn.Linear)]
        return self._film

    def _denoise_rows(self, x_rows: torch.Tensor, a_rows: torch.Tensor, t: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
        # every block projects the same cat(t, c): run all the projections now, in one launch (blocks pick theirs up by weight).
        # Not under reentrant activation checkpointing -- its recomputation runs each block's own projection anyway.
        if not (self.training and any(getattr(l, "gradient_checkpointing", False) for l in (*self.down_layers, *self.up_layers))):
            rt.film_prepare(rt.shared_cat(t, c), self._film_linears())
        r = x_rows
        x = x_rows
        skips = []
        for down_layer in self.down_layers:
            x, skip = down_layer.forward_rows(x, t, c)
            skips.append(skip)
        x = torch.cat([x, a_rows], dim=-1)
        x = self.middle_resnet1.forward_rows(x, t, c)
        for blk in self.middle_transformer:
            x = blk.forward_rows(x)
        x = self.middle_resnet2.forward_rows(x, t, c)
        for up_layer, skip in zip(self.up_layers, reversed(skips)):
            x = torch.cat([x, skip], dim=-1)
            x, _ = up_layer.forward_rows(x, t, c)
        x = torch.cat([x, r], dim=-1)
        x = self.final_resnet.forward_rows(x, t, c)
        nx = self.dim_in_x
        npad = (nx + 7) // 8 * 8
        w = F.pad(self.final_conv.weight[:, :, 0], (0, 0, 0, npad - nx))
        b = F.pad(self.final_conv.bias, (0, npad - nx))
        y = Fn.ConvFn.apply(x, w, b, self._cf, "same", ("final", self.final_conv.weight))
        return Fn.NCLFromRowsFn.apply(y, nx)

    def forward(self, x: torch.Tensor, a: torch.Tensor, t: torch.Tensor, c: torch.Tensor, cond_drop_prob: float = 0.0) -> torch.Tensor:
        rt.require_gpu(x)
        n = x.shape[-1]
        depth = len(self.down_layers)
        pad_len = (2 ** depth - (n % (2 ** depth))) % (2 ** depth)
        dtype = rt.compute_dtype(self.final_conv.weight.dtype)
        x = F.pad(x.float(), (0, pad_len), value=-1.0)
        x_rows = self.init_x.forward_rows(x, dtype)
        a_rows = self.encode_audio(a, pad_len, dtype)
        te = self.embed_time(t)
        cond_mask = prob_mask_like((x.shape[0],), 1.0 - cond_drop_prob, device=x.device)
        ce = self.embed_cond(c, cond_mask)
        return self.denoise_rows(x_rows, a_rows, te, ce)[:, :, :n]
