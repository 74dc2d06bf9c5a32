"""Mirror of osu_fusion/modules/utils.py (same names and semantics)."""
from contextlib import contextmanager
from typing import Generator

import torch


def right_pad_dims_to(x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    padding_dims = x.ndim - t.ndim
    if padding_dims <= 0:
        return t
    return t.view(*t.shape, *((1,) * padding_dims))


def prob_mask_like(shape, prob: float, device) -> torch.Tensor:
    """Classifier-free-guidance keep mask (utils.py:15-21): all-False at 0, all-True at 1, else uniform < prob."""
    if prob == 0.0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    if prob == 1.0:
        return torch.ones(shape, device=device, dtype=torch.bool)
    return torch.zeros(shape, device=device).uniform_(0.0, 1.0) < prob


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    return (x * cos) + (rotate_half(x) * sin)


@contextmanager
def dummy_context_manager() -> Generator[None, None, None]:
    yield
