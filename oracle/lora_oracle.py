"""ORACLE (test infrastructure, NOT product code) -- LoRA / DoRA adapter restatement for the fine-tune path.

Follows ``/root/reference/osu_fusion/modules/lora_layers.py`` and the wiring at ``/root/reference/trainer_peft.py:236-244``
(r=32, lora_alpha=32, use_dora=True; targets attn.to_q, attn.to_kv, block1.proj, block2.proj):
  :16-26    DoraConv1dLayer.get_weight_norm        -> weight_norm
  :60-92    DoraConv1dLayer.forward                -> dora_extra_conv1d   (the literal three-convolution formula)
  :284-310  LoraConv1d.get_delta_weight            -> delta_weight
  :312-328  LoraConv1d.forward                     -> lora_conv1d
  :199-246  LoraConv1d.merge (DoRA branch)         -> merged_weight
The Linear targets go through **peft==0.12.0** (requirements.txt:8) `lora.Linear` + `DoraLinearLayer`, absent from /root/reference
and from this image (and lora_layers.py itself imports peft, so it cannot be imported here either): their published formula --
identical to the conv one with k = 1 and a magnitude of shape (out,) -- is restated in lora_linear.  The reference holds no tests
or vectors for any of this -> **parity unpinned** for the adapter path; what IS checked: the product's merged-weight design equals
the literal formula (tests/test_oracle_golden.py), and the full adapted UNet = the golden-pinned UNet oracle run on
effective weights.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple

import torch
import torch.nn.functional as F  # noqa: N812

Tensor = torch.Tensor

TARGETS = ("attn.to_q", "attn.to_kv", "attn.linear", "block1.proj", "block2.proj")      # trainer_peft.py:241


def delta_weight(a: Tensor, b: Tensor, like: Tensor) -> Tensor:
    """B A as a weight shaped like the base (lora_layers.py:69-71; :284-298 computes the same via a conv)."""
    return (b.flatten(1) @ a.flatten(1)).reshape(like.shape)


def weight_norm(weight: Tensor, lora_weight: Tensor, scaling: float) -> Tensor:
    """Per-out-channel L2 norm of W + s*BA over all remaining dims, shape (O,)   (lora_layers.py:16-26)."""
    w = weight + scaling * lora_weight
    return w.reshape(w.shape[0], -1).norm(p=2, dim=1)


def dora_extra_conv1d(x: Tensor, w: Tensor, a: Tensor, b: Tensor, mag: Tensor, scaling: float, stride: int, padding: int) -> Tensor:
    """What DoraConv1dLayer.forward returns (added on top of the base layer output), lora_layers.py:72-92."""
    lw = delta_weight(a, b, w)
    norm = weight_norm(w, lw.detach(), scaling).detach()
    g = (mag.reshape(-1) / norm)[None, :, None]
    lora = F.conv1d(F.conv1d(x, a, None, stride=stride, padding=padding), b)
    return (g - 1) * F.conv1d(x, w, None, stride=stride, padding=padding) + g * lora * scaling


def lora_conv1d(x: Tensor, w: Tensor, bias: Optional[Tensor], a: Tensor, b: Tensor, mag: Optional[Tensor], scaling: float,
                stride: int = 1, padding: int = 1) -> Tensor:
    """LoraConv1d.forward, one active adapter, dropout 0 (lora_layers.py:312-328).  mag None = plain LoRA."""
    base = F.conv1d(x, w, bias, stride=stride, padding=padding)
    if mag is None:
        return base + F.conv1d(F.conv1d(x, a, None, stride=stride, padding=padding), b) * scaling
    return base + dora_extra_conv1d(x, w, a, b, mag, scaling, stride, padding)


def lora_linear(x: Tensor, w: Tensor, bias: Optional[Tensor], a: Tensor, b: Tensor, mag: Optional[Tensor], scaling: float) -> Tensor:
    """peft 0.12 lora.Linear.forward (+ DoraLinearLayer.forward): same algebra, k = 1."""
    base = F.linear(x, w, bias)
    lora = F.linear(F.linear(x, a), b) * scaling
    if mag is None:
        return base + lora
    norm = weight_norm(w, (b @ a).detach(), scaling).detach()
    g = mag.reshape(-1) / norm
    return base + (g - 1) * F.linear(x, w) + g * lora


def effective_weight(w: Tensor, a: Tensor, b: Tensor, mag: Optional[Tensor], scaling: float) -> Tensor:
    """g * (W + s*BA) with the norm detached: conv(x, effective) + bias == lora_conv1d(...) identically (the merged form,
    lora_layers.py:236-241 with the live adapter), and it is differentiable in (a, b, mag) with the same gradients."""
    v = w + scaling * delta_weight(a, b, w)
    if mag is None:
        return v
    g = mag.reshape(-1) / weight_norm(w, delta_weight(a, b, w).detach(), scaling).detach()
    return g.reshape(-1, *([1] * (w.dim() - 1))) * v


def merged_weight(w: Tensor, a: Tensor, b: Tensor, mag: Optional[Tensor], scaling: float) -> Tensor:
    """base_layer.weight after LoraConv1d.merge (lora_layers.py:199-246)."""
    return effective_weight(w, a, b, mag, scaling).detach()


def target_names(param_names: Iterable[str], targets: Tuple[str, ...] = TARGETS):
    """Module paths (without '.weight') that peft's suffix rule selects."""
    out = []
    for n in param_names:
        if n.endswith(".weight"):
            mod = n[: -len(".weight")]
            if any(mod == t or mod.endswith("." + t) for t in targets):
                out.append(mod)
    return out


def adapter_shapes(w_shape: Tuple[int, ...], r: int):
    """(lora_A, lora_B, magnitude) shapes for a base weight (lora_layers.py:151-160,25; peft Linear)."""
    if len(w_shape) == 3:
        o, i, k = w_shape
        return (r, i, k), (o, r, 1), (1, o, 1)
    o, i = w_shape
    return (r, i), (o, r), (o,)


def effective_params(p: Dict[str, Tensor], adapters: Dict[str, Tuple[Tensor, Tensor, Optional[Tensor]]], scaling: float) -> Dict[str, Tensor]:
    """Parameter dict for unet_oracle.unet_forward with every adapted weight replaced by its effective weight."""
    q = dict(p)
    for mod, (a, b, mag) in adapters.items():
        q[mod + ".weight"] = effective_weight(p[mod + ".weight"], a, b, mag, scaling)
    return q
