"""Deterministic, platform-independent tensor patterns (integer hash -> uniform floats).

Used to fill weights and inputs identically in three places without shipping weight blobs:
the golden-vector generator (runs next to the reference), the oracle tests, and the HIP
path's parity tests / bench.  Pure integer arithmetic (splitmix64) so the values are
bit-identical on every machine; no dependence on numpy/torch RNG streams.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def hash_uniform(tag: str, n: int) -> np.ndarray:
    """n float64 values in [0, 1), a pure function of (tag, index)."""
    seed = np.uint64(zlib.crc32(tag.encode("utf-8")) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * _GOLD + seed * _M1
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform_pm(tag: str, shape: Tuple[int, ...], amp: float) -> np.ndarray:
    """float32 array uniform in [-amp, amp)."""
    n = int(np.prod(shape)) if len(shape) else 1
    return ((hash_uniform(tag, n) * 2.0 - 1.0) * amp).astype(np.float32).reshape(shape)


def param_pattern(name: str, shape: Tuple[int, ...]) -> np.ndarray:
    """Weight fill rule keyed on the state_dict name.

    * norm weights      -> 1 + U(-0.2, 0.2)
    * biases, null_cond -> U(-0.1, 0.1)  (null_cond U(-1, 1))
    * conv / linear     -> U(-a, a), a = 1 / sqrt(fan_in)  (the bound torch's default Conv1d/Linear init uses,
                           i.e. the reference's own initialisation scale)
    final_conv is *not* zero here: the reference zero-inits it (unet.py:354), which makes
    every other gradient exactly zero and a backward parity test vacuous.
    """
    leaf = name.split(".")[-1]
    if name.endswith("null_cond"):
        return uniform_pm(name, shape, 1.0)
    if ".norm." in name or name.split(".")[-2] == "norm":
        if leaf == "weight":
            return (1.0 + uniform_pm(name, shape, 0.2)).astype(np.float32)
        return uniform_pm(name, shape, 0.1)
    if leaf == "bias":
        return uniform_pm(name, shape, 0.1)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return uniform_pm(name, shape, float(np.sqrt(1.0 / max(fan_in, 1))))


def fill_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]]) -> Dict[str, np.ndarray]:
    return {k: param_pattern(k, tuple(s)) for k, s in shapes}


def synth_inputs(tag: str, batch: int, length: int, dim_x: int = 6, dim_a: int = 96, dim_c: int = 5):
    """Synthetic (x, a, c, t, noise) following SURVEY §8(d): x in [-1,1], a log-VQT-like, c in [-1,1]."""
    x = uniform_pm(tag + "/x", (batch, dim_x, length), 1.0)
    a = (uniform_pm(tag + "/a", (batch, dim_a, length), 5.0) - 10.0).astype(np.float32)
    c = uniform_pm(tag + "/c", (batch, dim_c), 1.0)
    t = (hash_uniform(tag + "/t", batch) * 1000.0).astype(np.int64)
    # noise: sum of 4 uniforms, rescaled to unit variance (bounded, gaussian-ish)
    u = sum(hash_uniform(f"{tag}/n{i}", batch * dim_x * length) for i in range(4))
    noise = ((u - 2.0) * np.sqrt(3.0)).astype(np.float32).reshape(batch, dim_x, length)
    return x, a, c, t, noise
