"""Data formats either side of the hot path: the dataset files the reference's `dataset_creator.py` writes, the
padding collate of `trainer.py`, the context normalisation, and model checkpoints.  Host-side only -- nothing here touches a kernel.

  <name>.map.npz : x (6, L) float32 beatmap signals in [-1, 1], c (5,) normalised context, spec_path (relative path of spec.npz)
                   (scripts/dataset_creator.py:180)
  spec.npz       : a (96, L) float32 log-VQT of the song, shared by all difficulties of a set (scripts/dataset_creator.py:114)
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F  # noqa: N812

PAD_X, PAD_A = -1.0, -23.0          # trainer.py:84-85; the UNet pads with the same values (modules/unet.py:475-480)


def load_tensor(map_file: Path) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """library/dataset.py:25-37: (x (6, L), a (96, L), c (5,)) as float32; ValueError on NaNs."""
    map_file = Path(map_file)
    map_data = np.load(map_file)
    audio_data = np.load(map_file.parent / map_data["spec_path"].tolist())
    x = torch.tensor(map_data["x"], dtype=torch.float32)
    c = torch.tensor(map_data["c"], dtype=torch.float32)
    a = torch.tensor(audio_data["a"], dtype=torch.float32)
    if torch.isnan(x).any() or torch.isnan(a).any() or torch.isnan(c).any():
        raise ValueError("Invalid values in map file")
    return x, a, c


def save_tensor(map_file: Path, x: np.ndarray, c: np.ndarray, a: np.ndarray, spec_name: str = "spec.npz") -> None:
    """Write one (map, spec) pair in the layout above (scripts/dataset_creator.py:114,180)."""
    map_file = Path(map_file)
    spec = map_file.parent / spec_name
    if not spec.exists():
        np.savez_compressed(spec, a=np.asarray(a, dtype=np.float32))
    np.savez_compressed(map_file, x=np.asarray(x, dtype=np.float32), c=np.asarray(c, dtype=np.float32), spec_path=spec_name)


def collate_fn(batch: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]):
    """trainer.py:74-95: right-pad every sample to the longest one (x with -1.0, a with -23.0) and return the original lengths --
    `OsuFusion.forward(x, a, c, orig_len)` masks the loss with them (models/diffusion.py:104-110)."""
    max_length = max(x.shape[1] for x, _, _ in batch)
    xs, as_, cs, orig = [], [], [], []
    for x, a, c in batch:
        orig.append(x.shape[1])
        xs.append(F.pad(x, (0, max_length - x.shape[1]), mode="constant", value=PAD_X))
        as_.append(F.pad(a, (0, max_length - a.shape[1]), mode="constant", value=PAD_A))
        cs.append(c)
    return torch.stack(xs), torch.stack(as_), torch.stack(cs), torch.tensor(orig)


def normalize_context(context: np.ndarray) -> np.ndarray:
    """scripts/dataset_creator.py:57-66 (in place): CS, AR, OD, HP from [0, 10] and star rating from [0, 20] to [-1, 1]."""
    context[:4] = context[:4] / 5 - 1
    context[4] = context[4] / 10 - 1
    return context


def unnormalize_context(context):
    """scripts/dataset_creator.py:69-78 (in place), numpy array or tensor."""
    context[:4] = (context[:4] + 1) * 5
    context[4] = (context[4] + 1) * 10
    return context


def filter_dataset(map_files: Sequence[Path], max_length: int) -> List[Path]:
    """Keep maps of at most max_length frames (trainer.py's `filter_dataset`); reads only the npz headers' x array."""
    keep = []
    for f in map_files:
        with np.load(f) as d:
            if d["x"].shape[1] <= max_length:
                keep.append(Path(f))
    return keep


def save_model_sd(model: torch.nn.Module, path: Path) -> None:
    """trainer.py:143-145: `model.safetensors` with the module's own key names."""
    from safetensors.torch import save_file
    save_file({k: v.detach().contiguous().cpu() for k, v in model.state_dict().items()}, str(path))


def load_model_sd(model: torch.nn.Module, path: Path, strict: bool = True) -> Dict[str, List[str]]:
    """inference_gradio.py:33-41 / trainer_peft.py:233: load `model.safetensors` or a `checkpoint.pt` (its "model_state_dict")."""
    path = Path(path)
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file
        sd = load_file(str(path))
    else:
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("model_state_dict", sd)
    res = model.load_state_dict(sd, strict=strict)
    from . import functional as Fn
    Fn.bump_weight_epoch()
    return {"missing": list(res.missing_keys), "unexpected": list(res.unexpected_keys)}
