"""Host-side helpers with the names and semantics of osu_fusion/modules/utils.py (the UNet itself only needs
``prob_mask_like``; the rest exists so code written against the reference module keeps importing).  Pure tensor-shape
utilities on whatever device the caller's tensors live on -- none of them is on the hot path: the rotary embedding
of the UNet runs inside osuf_rope_cast / the attention-backward epilogues."""
from contextlib import nullcontext

import torch


def right_pad_dims_to(x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """utils.py:7-11: append singleton axes to t until it has as many axes as x."""
    missing = x.dim() - t.dim()
    return t if missing <= 0 else t[(...,) + (None,) * missing]


def prob_mask_like(shape, prob: float, device) -> torch.Tensor:
    """utils.py:15-21, the classifier-free-guidance keep mask: prob is P(True) per entry.  The end points are exact
    (prob == 1 -> all kept, prob == 0 -> none) and draw nothing from the RNG; in between one uniform draw per entry."""
    if prob in (0.0, 1.0):
        return torch.full(shape, bool(prob), dtype=torch.bool, device=device)
    return torch.rand(shape, device=device) < prob


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """utils.py:25-27: (x1 | x2) -> (-x2 | x1) over the two halves of the last axis."""
    half = x.shape[-1] // 2
    return torch.cat((x[..., half:].neg(), x[..., :half]), dim=-1)


def apply_rotary_pos_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """utils.py:30-32 (half-split rotation): y1 = x1 cos - x2 sin, y2 = x2 cos + x1 sin."""
    return torch.addcmul(x * cos, rotate_half(x), sin)


def dummy_context_manager():
    """utils.py:36-38: the no-op stand-in the reference swaps in for record_function when DEBUG is unset."""
    return nullcontext()
