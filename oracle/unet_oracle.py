"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the OsuFusion denoiser.

This is a from-scratch functional restatement (plain PyTorch CPU ops over a flat
``{state_dict_key: tensor}`` dict) of the reference's UNet hot path.  It exists only to check
the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Parity pinning: ``tests/golden/*.npz`` were produced by importing the
reference's own modules in the build container (``tests/golden/make_golden.py``); the oracle
is checked against them in ``tests/test_oracle_golden.py`` (fp32, rtol 1e-5).

Reference citations (``/root/reference/osu_fusion/...``):
  modules/unet.py:26-39    SinusoidalPositionEmbedding   -> sinusoidal_embedding
  modules/unet.py:42-58    CrossEmbedLayer               -> cross_embed
  modules/unet.py:61-92    Upsample / Downsample         -> upsample / downsample
  modules/unet.py:95-101   Parallel (k3 + k1)            -> parallel_conv
  modules/unet.py:104-146  Attention (MQA + RoPE)        -> attention
  modules/unet.py:149-183  FeedForward/TransformerBlock  -> transformer_block
  modules/unet.py:186-263  UNetBlock                     -> unet_block
  modules/unet.py:266-318  AudioEncoder                  -> audio_encoder
  modules/unet.py:321-513  UNet                          -> unet_forward
  modules/residual.py:14-37   GlobalContext              -> global_context
  modules/residual.py:62-88   Block                      -> block
  modules/residual.py:91-137  ResidualBlock              -> residual_block
  modules/attention.py:15-58  RotaryPositionEmbedding    -> rope_tables / apply_rope
  modules/attention.py:61-101 Attend (bf16 SDPA)         -> attend
  modules/utils.py:15-21      prob_mask_like             -> cond mask handling in unet_forward

``mode``:
  "fp32"  the reference's numerics on an fp32 run: fp32 everywhere except q,k,v -> bf16 SDPA
          (attention.py:87-101).  This is what the goldens pin.
  "bf16"  emulation of the HIP path's bf16-autocast numerics: every activation that the HIP
          path stores in HBM as bf16 is rounded to bf16 at the same point, and GEMM/conv
          operands (activations and weights) are bf16-rounded with fp32 accumulation.
          Used to check the bf16 kernels tightly; its distance to "fp32" is the price of bf16.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class UNetConfig:
    dim_in_x: int = 6
    dim_in_a: int = 96
    dim_in_c: int = 5
    dim_h: int = 256
    dim_h_mult: Tuple[int, ...] = (1, 2, 3, 4)
    num_layer_blocks: Tuple[int, ...] = (3, 3, 3, 3)
    num_middle_transformers: int = 3
    cross_embed_kernel_sizes: Tuple[int, ...] = (3, 7, 15)
    attn_dim_head: int = 64
    attn_heads: int = 16
    attn_kv_heads: int = 1
    attn_context_len: int = 4096

    @property
    def dim_emb(self) -> int:
        return self.dim_h * 4

    @property
    def dims_h(self) -> Tuple[int, ...]:
        return (self.dim_h, *[self.dim_h * m for m in self.dim_h_mult])


# --------------------------------------------------------------------------------------
# parameter inventory (state_dict names/shapes; SURVEY §8b "State / ownership")
# --------------------------------------------------------------------------------------

def _cross_embed_shapes(pre: str, dim: int, dim_out: int, ks: Tuple[int, ...]) -> List[Tuple[str, Tuple[int, ...]]]:
    ks = sorted(ks)
    scales = [int(dim / (2 ** i)) for i in range(1, len(ks))]        # unet.py:48 (derives from dim *in*)
    scales = [*scales, dim_out - sum(scales)]
    out = []
    for i, (k, d) in enumerate(zip(ks, scales)):
        out += [(f"{pre}.convs.{i}.weight", (d, dim, k)), (f"{pre}.convs.{i}.bias", (d,))]
    return out


def _resblock_shapes(pre: str, cin: int, cout: int, dim_cond: Optional[int]) -> List[Tuple[str, Tuple[int, ...]]]:
    out = []
    if dim_cond:
        out += [(f"{pre}.mlp.1.weight", (cout * 2, dim_cond)), (f"{pre}.mlp.1.bias", (cout * 2,))]
    for b, ci in (("block1", cin), ("block2", cout)):
        out += [(f"{pre}.{b}.proj.weight", (cout, ci, 3)), (f"{pre}.{b}.proj.bias", (cout,)),
                (f"{pre}.{b}.norm.weight", (cout,)), (f"{pre}.{b}.norm.bias", (cout,))]
    if cin != cout:
        out += [(f"{pre}.res_conv.weight", (cout, cin, 1)), (f"{pre}.res_conv.bias", (cout,))]
    inner = max(8, cout // 2)                                          # residual.py:20
    out += [(f"{pre}.se.to_k.weight", (1, cout, 1)), (f"{pre}.se.to_k.bias", (1,)),
            (f"{pre}.se.layers.0.weight", (inner, cout, 1)), (f"{pre}.se.layers.0.bias", (inner,)),
            (f"{pre}.se.layers.2.weight", (cout, inner, 1)), (f"{pre}.se.layers.2.bias", (cout,))]
    return out


def _transformer_shapes(pre: str, dim: int, cfg: UNetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    hd = cfg.attn_dim_head
    return [
        (f"{pre}.attn.norm.weight", (dim,)), (f"{pre}.attn.norm.bias", (dim,)),
        (f"{pre}.attn.to_q.weight", (hd * cfg.attn_heads, dim)),
        (f"{pre}.attn.to_kv.weight", (hd * cfg.attn_kv_heads * 2, dim)),
        (f"{pre}.attn.to_out.weight", (dim, hd * cfg.attn_heads)), (f"{pre}.attn.to_out.bias", (dim,)),
        (f"{pre}.ff.0.weight", (dim * 2, dim)), (f"{pre}.ff.0.bias", (dim * 2,)),
        (f"{pre}.ff.2.weight", (dim, dim * 2)), (f"{pre}.ff.2.bias", (dim,)),
    ]


def _unet_block_shapes(pre, dim_in, dim_out, dim_cond, layer_idx, n_layers, n_blocks, down, cfg):
    out = _resblock_shapes(f"{pre}.init_resnet", dim_in if down else dim_in + dim_out, dim_in, dim_cond)
    for i in range(n_blocks):
        out += _resblock_shapes(f"{pre}.resnets.{i}", dim_in, dim_in, dim_cond)
    for i in range(n_blocks):
        out += _transformer_shapes(f"{pre}.transformers.{i}", dim_in, cfg)
    if layer_idx < n_layers - 1:
        out += [(f"{pre}.sampler.conv.weight", (dim_out, dim_in, 3)), (f"{pre}.sampler.conv.bias", (dim_out,))]
    else:
        out += [(f"{pre}.sampler.fns.0.weight", (dim_out, dim_in, 3)), (f"{pre}.sampler.fns.0.bias", (dim_out,)),
                (f"{pre}.sampler.fns.1.weight", (dim_out, dim_in, 1)), (f"{pre}.sampler.fns.1.bias", (dim_out,))]
    return out


def param_shapes(cfg: UNetConfig, prefix: str = "") -> List[Tuple[str, Tuple[int, ...]]]:
    """All UNet state_dict entries (name, shape).  1,239 entries at the default dim_h=256 config."""
    P = prefix
    E = cfg.dim_emb
    dims = cfg.dims_h
    in_out = list(zip(dims[:-1], dims[1:]))
    n = len(in_out)
    out: List[Tuple[str, Tuple[int, ...]]] = [(f"{P}null_cond", (E,))]
    out += _cross_embed_shapes(f"{P}init_x", cfg.dim_in_x, cfg.dim_h, cfg.cross_embed_kernel_sizes)
    out += _cross_embed_shapes(f"{P}audio_encoder.init_conv", cfg.dim_in_a, cfg.dim_h, cfg.cross_embed_kernel_sizes)
    for i, (di, do) in enumerate(in_out):
        out += _unet_block_shapes(f"{P}audio_encoder.layers.{i}", di, do, None, i, n, cfg.num_layer_blocks[i], True, cfg)
    out += _resblock_shapes(f"{P}final_resnet", cfg.dim_h * 2, cfg.dim_h, 2 * E)
    out += [(f"{P}final_conv.weight", (cfg.dim_in_x, cfg.dim_h, 1)), (f"{P}final_conv.bias", (cfg.dim_in_x,))]
    out += [(f"{P}time_mlp.1.weight", (E, E)), (f"{P}time_mlp.1.bias", (E,)),
            (f"{P}time_mlp.3.weight", (E, E)), (f"{P}time_mlp.3.bias", (E,))]
    out += [(f"{P}cond_mlp.0.weight", (E, cfg.dim_in_c)), (f"{P}cond_mlp.0.bias", (E,)),
            (f"{P}cond_mlp.2.weight", (E, E)), (f"{P}cond_mlp.2.bias", (E,))]
    for i, (di, do) in enumerate(in_out):
        out += _unet_block_shapes(f"{P}down_layers.{i}", di, do, 2 * E, i, n, cfg.num_layer_blocks[i], True, cfg)
    out += _resblock_shapes(f"{P}middle_resnet1", dims[-1] * 2, dims[-1], 2 * E)
    for i in range(cfg.num_middle_transformers):
        out += _transformer_shapes(f"{P}middle_transformer.{i}", dims[-1], cfg)
    out += _resblock_shapes(f"{P}middle_resnet2", dims[-1], dims[-1], 2 * E)
    rev = list(reversed(in_out))
    rblocks = list(reversed(cfg.num_layer_blocks))
    for i, (do, di) in enumerate(rev):                                  # unet.py:430 (layer_dim_out, layer_dim_in)
        out += _unet_block_shapes(f"{P}up_layers.{i}", di, do, 2 * E, i, n, rblocks[i], False, cfg)
    return out


# --------------------------------------------------------------------------------------
# rounding model
# --------------------------------------------------------------------------------------

class _RoundBF16(torch.autograd.Function):
    """bf16 round-trip in forward AND backward (activations and their grads live in HBM as bf16)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class Numerics:
    def __init__(self, mode: str = "fp32"):
        assert mode in ("fp32", "bf16")
        self.mode = mode

    def act(self, x: Tensor) -> Tensor:
        """An activation tensor written to HBM by the HIP path."""
        return _RoundBF16.apply(x) if self.mode == "bf16" else x

    def lin(self, x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
        """An embedding-sized Linear / 1x1 conv on (B, K): fp32 in the fp32 run; bf16 operands, fp32 accumulate and a
        bf16-stored result in bf16 mode (what torch autocast does to nn.Linear, and what the HIP path does)."""
        w2 = w.reshape(w.shape[0], -1)
        if self.mode == "bf16":
            return self.act(F.linear(self.act(x), self.w(w2), b))
        if w.dim() == 3:                                   # fp32 run: the reference's own op (1x1 Conv1d on (B, C, 1))
            return F.conv1d(x[..., None], w, b)[..., 0]
        return F.linear(x, w2, b)

    def w(self, w: Tensor) -> Tensor:
        """A GEMM/conv weight operand (bf16 pack of the fp32 master in bf16 mode)."""
        if self.mode == "bf16":
            return w + (w.to(torch.bfloat16).to(torch.float32) - w).detach()   # straight-through
        return w


# --------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------

def sinusoidal_embedding(t: Tensor, dim: int, theta: float = 10000.0) -> Tensor:
    half = dim // 2
    step = math.log(theta) / (half - 1)                                 # unet.py:35: divisor half_dim-1
    freqs = torch.exp(torch.arange(half) * -step)
    arg = t[:, None] * freqs[None, :]                                   # int64 t promotes to fp32
    return torch.cat([arg.sin(), arg.cos()], dim=-1)


def cross_embed(p: Dict[str, Tensor], pre: str, x: Tensor, ks: Tuple[int, ...], nm: Numerics) -> Tensor:
    outs = []
    for i, k in enumerate(sorted(ks)):
        outs.append(F.conv1d(x, nm.w(p[f"{pre}.convs.{i}.weight"]), p[f"{pre}.convs.{i}.bias"], padding=k // 2))
    return nm.act(torch.cat(outs, dim=1))


def global_context(p: Dict[str, Tensor], pre: str, h: Tensor, nm: Numerics) -> Tensor:
    """residual.py:29-32.  h: (B,C,N) -> gate (B,C,1).  Pooling in fp32 in both modes."""
    logits = F.conv1d(h, p[f"{pre}.to_k.weight"], p[f"{pre}.to_k.bias"])         # (B,1,N)
    w = logits.softmax(dim=-1)
    pooled = torch.einsum("bcn,bjn->bcj", h, w)[..., 0]                           # (B,C)
    g = nm.lin(pooled, p[f"{pre}.layers.0.weight"], p[f"{pre}.layers.0.bias"])
    g = nm.lin(F.silu(g), p[f"{pre}.layers.2.weight"], p[f"{pre}.layers.2.bias"])
    return torch.sigmoid(g)[..., None]


def squeeze_excite(p: Dict[str, Tensor], pre: str, h: Tensor, nm: Numerics) -> Tensor:
    """residual.py:40-59 (the gate ResidualBlock(use_gca=False) builds, residual.py:116): mean over the sequence -> 1x1 MLP -> sigmoid."""
    pooled = h.mean(dim=-1)                                                        # AdaptiveAvgPool1d(1)
    g = nm.lin(pooled, p[f"{pre}.layers.0.weight"], p[f"{pre}.layers.0.bias"])
    g = nm.lin(F.silu(g), p[f"{pre}.layers.2.weight"], p[f"{pre}.layers.2.bias"])
    return torch.sigmoid(g)[..., None]


def block(p, pre, x, scale_shift, nm: Numerics) -> Tensor:
    """residual.py:75-84: conv3 -> GroupNorm(1,C) -> FiLM -> SiLU.  Without `norm.*` parameters: Block(norm=False), whose norm is
    nn.Identity (residual.py:71)."""
    y = F.conv1d(x, nm.w(p[f"{pre}.proj.weight"]), p[f"{pre}.proj.bias"], padding=1)
    y = nm.act(y)                                                      # HIP path: conv output stored, stats on stored values
    if f"{pre}.norm.weight" in p:
        y = F.group_norm(y, 1, p[f"{pre}.norm.weight"], p[f"{pre}.norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        y = y * (scale + 1) + shift
    return nm.act(F.silu(y))


def residual_block(p, pre, x, t, c, nm: Numerics) -> Tensor:
    """residual.py:118-137."""
    scale_shift = None
    if f"{pre}.mlp.1.weight" in p:
        emb = torch.cat([e for e in (t, c) if e is not None], dim=-1)
        emb = nm.lin(F.silu(emb), p[f"{pre}.mlp.1.weight"], p[f"{pre}.mlp.1.bias"])     # SiLU first (residual.py:106)
        emb = emb[:, :, None]
        scale_shift = emb.chunk(2, dim=1)
    h = block(p, f"{pre}.block1", x, scale_shift, nm)
    h = block(p, f"{pre}.block2", h, None, nm)
    gate = global_context(p, f"{pre}.se", h, nm) if f"{pre}.se.to_k.weight" in p else squeeze_excite(p, f"{pre}.se", h, nm)
    if f"{pre}.res_conv.weight" in p:
        res = F.conv1d(x, nm.w(p[f"{pre}.res_conv.weight"]), p[f"{pre}.res_conv.bias"])
    else:
        res = x
    return nm.act(h * gate + res)


def rope_tables(n: int, dim: int, scale_base: int, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """attention.py:24-47 in fp32: positions rescaled by scale_base / n; half-split layout."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim))
    pos = torch.arange(n, dtype=torch.float32)
    pos = pos * (scale_base / n)
    freqs = torch.einsum("i,j->ij", pos, inv_freq)
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    x1, x2 = x.chunk(2, dim=-1)
    return x * cos + torch.cat((-x2, x1), dim=-1) * sin               # utils.py:25-32


def attend(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """attention.py:84-101 on a gfx9/sm80+ device: q,k,v -> bf16, SDPA, back to input dtype."""
    dt = v.dtype
    q, k, v = (t.to(torch.bfloat16).contiguous() for t in (q, k, v))
    return F.scaled_dot_product_attention(q, k, v).to(dt)


def attention(p, pre, x: Tensor, cfg: UNetConfig, scale_base: int, nm: Numerics) -> Tensor:
    """unet.py:125-141.  x: (B,N,C).  Residual is taken from the *normed* x (quirk)."""
    B, N, C = x.shape
    H, KV, D = cfg.attn_heads, cfg.attn_kv_heads, cfg.attn_dim_head
    xn = nm.act(F.layer_norm(x, (C,), p[f"{pre}.norm.weight"], p[f"{pre}.norm.bias"], eps=1e-5))
    q = F.linear(xn, nm.w(p[f"{pre}.to_q.weight"])).view(B, N, H, D).transpose(1, 2)
    kv = F.linear(xn, nm.w(p[f"{pre}.to_kv.weight"]))
    k, v = kv.chunk(2, dim=-1)
    k = k.reshape(B, N, KV, D).transpose(1, 2)
    v = v.reshape(B, N, KV, D).transpose(1, 2)
    rep = H // KV
    k = k[:, None].expand(B, rep, KV, N, D).reshape(B, H, N, D)        # "b h n d -> b (r h) n d"
    v = v[:, None].expand(B, rep, KV, N, D).reshape(B, H, N, D)
    cos, sin = rope_tables(N, D, scale_base)
    q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    o = attend(q, k, v)                                                # bf16 in both modes
    o = o.transpose(1, 2).reshape(B, N, H * D)
    return nm.act(xn + F.linear(o, nm.w(p[f"{pre}.to_out.weight"]), p[f"{pre}.to_out.bias"]))


def transformer_block(p, pre, x: Tensor, cfg: UNetConfig, scale_base: int, nm: Numerics) -> Tensor:
    """unet.py:179-183.  x: (B,C,N) in, (B,C,N) out; FF has no pre-norm."""
    x = x.transpose(1, 2)
    x = attention(p, f"{pre}.attn", x, cfg, scale_base, nm)
    h = nm.act(F.silu(F.linear(x, nm.w(p[f"{pre}.ff.0.weight"]), p[f"{pre}.ff.0.bias"])))
    x = nm.act(F.linear(h, nm.w(p[f"{pre}.ff.2.weight"]), p[f"{pre}.ff.2.bias"]) + x)
    return x.transpose(1, 2)


def downsample(p, pre, x: Tensor, nm: Numerics) -> Tensor:
    x = F.pad(x, (0, 1), mode="reflect")                               # right only (unet.py:84-85)
    return nm.act(F.conv1d(x, nm.w(p[f"{pre}.conv.weight"]), p[f"{pre}.conv.bias"], stride=2))


def upsample(p, pre, x: Tensor, nm: Numerics) -> Tensor:
    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return nm.act(F.conv1d(x, nm.w(p[f"{pre}.conv.weight"]), p[f"{pre}.conv.bias"], padding=1))


def parallel_conv(p, pre, x: Tensor, nm: Numerics) -> Tensor:
    y3 = F.conv1d(x, nm.w(p[f"{pre}.fns.0.weight"]), p[f"{pre}.fns.0.bias"], padding=1)
    y1 = F.conv1d(x, nm.w(p[f"{pre}.fns.1.weight"]), p[f"{pre}.fns.1.bias"])
    return nm.act(y3 + y1)


def unet_block(p, pre, x, t, c, cfg: UNetConfig, n_blocks: int, scale_base: int, down: bool, nm: Numerics):
    """unet.py:240-252 -> (sampled, pre-sample skip)."""
    x = residual_block(p, f"{pre}.init_resnet", x, t, c, nm)
    for i in range(n_blocks):
        x = residual_block(p, f"{pre}.resnets.{i}", x, t, c, nm)
        x = transformer_block(p, f"{pre}.transformers.{i}", x, cfg, scale_base, nm)
    if f"{pre}.sampler.conv.weight" in p:
        y = downsample(p, f"{pre}.sampler", x, nm) if down else upsample(p, f"{pre}.sampler", x, nm)
    else:
        y = parallel_conv(p, f"{pre}.sampler", x, nm)
    return y, x


def audio_encoder(p, pre, a: Tensor, cfg: UNetConfig, nm: Numerics) -> Tensor:
    a = cross_embed(p, f"{pre}.init_conv", a, cfg.cross_embed_kernel_sizes, nm)
    for i in range(len(cfg.dim_h_mult)):
        # AudioEncoder is built without attn_context_len (unet.py:343-352) => always 4096-based
        a, _ = unet_block(p, f"{pre}.layers.{i}", a, None, None, cfg, cfg.num_layer_blocks[i], 4096 // (2 ** i), True, nm)
    return a


def unet_forward(p: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, a: Tensor, t: Tensor, c: Tensor,
                 cond_drop_prob: float = 0.0, cond_mask: Optional[Tensor] = None, mode: str = "fp32",
                 prefix: str = "") -> Tensor:
    """unet.py:467-513.  ``cond_mask`` (B,) bool overrides the RNG draw of prob_mask_like for 0<p<1."""
    nm = Numerics(mode)
    P = prefix
    n = x.shape[-1]
    depth = len(cfg.dim_h_mult)
    pad = (2 ** depth - (n % (2 ** depth))) % (2 ** depth)
    x = F.pad(x, (0, pad), value=-1.0)
    a = F.pad(a, (0, pad), value=-23.0)

    x = cross_embed(p, f"{P}init_x", x, cfg.cross_embed_kernel_sizes, nm)
    a = audio_encoder(p, f"{P}audio_encoder", a, cfg, nm)
    E = cfg.dim_emb
    te = sinusoidal_embedding(t, E)
    te = nm.lin(te, p[f"{P}time_mlp.1.weight"], p[f"{P}time_mlp.1.bias"])
    te = nm.lin(F.silu(te), p[f"{P}time_mlp.3.weight"], p[f"{P}time_mlp.3.bias"])
    r = x

    B = x.shape[0]
    if cond_mask is None:
        keep = 1.0 - cond_drop_prob
        if keep == 0.0:
            cond_mask = torch.zeros(B, dtype=torch.bool)
        elif keep == 1.0:
            cond_mask = torch.ones(B, dtype=torch.bool)
        else:
            cond_mask = torch.zeros(B).uniform_(0.0, 1.0) < keep
    ce = nm.lin(c, p[f"{P}cond_mlp.0.weight"], p[f"{P}cond_mlp.0.bias"])
    ce = nm.lin(F.silu(ce), p[f"{P}cond_mlp.2.weight"], p[f"{P}cond_mlp.2.bias"])
    ce = torch.where(cond_mask[:, None], ce, p[f"{P}null_cond"][None, :].expand(B, E))

    L = len(cfg.dim_h_mult)
    skips = []
    for i in range(L):
        x, skip = unet_block(p, f"{P}down_layers.{i}", x, te, ce, cfg, cfg.num_layer_blocks[i],
                             cfg.attn_context_len // (2 ** i), True, nm)
        skips.append(skip)
    x = torch.cat([x, a], dim=1)
    x = residual_block(p, f"{P}middle_resnet1", x, te, ce, nm)
    for i in range(cfg.num_middle_transformers):
        x = transformer_block(p, f"{P}middle_transformer.{i}", x, cfg, cfg.attn_context_len // (2 ** (L - 1)), nm)
    x = residual_block(p, f"{P}middle_resnet2", x, te, ce, nm)
    rblocks = list(reversed(cfg.num_layer_blocks))
    for i in range(L):
        x = torch.cat([x, skips[L - 1 - i]], dim=1)
        x, _ = unet_block(p, f"{P}up_layers.{i}", x, te, ce, cfg, rblocks[i],
                          cfg.attn_context_len // (2 ** (L - i - 1)), False, nm)
    x = torch.cat([x, r], dim=1)
    x = residual_block(p, f"{P}final_resnet", x, te, ce, nm)
    y = F.conv1d(x, nm.w(p[f"{P}final_conv.weight"]), p[f"{P}final_conv.bias"])
    return y[:, :, :n]


def make_params(cfg: UNetConfig, requires_grad: bool = False, prefix: str = "") -> Dict[str, Tensor]:
    """State dict filled with the closed-form pattern of osufusion_amd.pattern (shared with the goldens)."""
    from osufusion_amd.pattern import param_pattern
    out = {}
    for name, shape in param_shapes(cfg, prefix):
        t = torch.from_numpy(param_pattern(name[len(prefix):] if prefix else name, shape).copy())
        out[name] = t.requires_grad_(requires_grad)
    return out
