"""Profiler scopes with the reference's names, as roctx ranges.

The reference wraps its blocks in ``torch.profiler.record_function`` scopes -- "Upsample", "Downsample", "Attention"
(osu_fusion/modules/unet.py:72,90,144), "GlobalContext", "Residual's Block" (residual.py:35,86) -- and swaps in a no-op
context manager when the DEBUG environment variable is set (unet.py:15, residual.py:10).  Here the same names become roctx
ranges (libroctx64: ``rocprofv3 --marker-trace`` shows them around the HIP kernels of each block, forward and backward
[autograd Function backward scopes get a " (backward)" suffix]) plus ``record_function`` scopes for torch.profiler users.

They cost a library call per block, so they are OFF unless asked for: ``OSUF_TRACE=1`` in the environment or
``tracing.enable(True)``; ``DEBUG`` set keeps them off, as in the reference.  Off, ``scope()`` returns one shared
no-op context (nothing allocated per call).
"""
from __future__ import annotations

import ctypes
import os
from contextlib import nullcontext

_NOOP = nullcontext()
_ENABLED = bool(os.environ.get("OSUF_TRACE")) and not os.environ.get("DEBUG")
_roctx = None

SCOPES = ("Upsample", "Downsample", "Attention", "GlobalContext", "SqueezeExcite", "Residual's Block")      # the reference's scope names


def _lib():
    global _roctx
    if _roctx is None:
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _roctx = lib
                break
            except (OSError, AttributeError):
                continue
        else:
            _roctx = False
    return _roctx


def enable(flag: bool = True) -> None:
    global _ENABLED
    _ENABLED = bool(flag)


def enabled() -> bool:
    return _ENABLED


class _Range:
    __slots__ = ("name", "rf")

    def __init__(self, name: str) -> None:
        self.name, self.rf = name, None

    def __enter__(self):
        lib = _lib()
        if lib:
            lib.roctxRangePushA(self.name.encode())
        import torch
        self.rf = torch.profiler.record_function(self.name)
        self.rf.__enter__()
        return self

    def __exit__(self, *exc):
        self.rf.__exit__(*exc)
        lib = _lib()
        if lib:
            lib.roctxRangePop()
        return False


def scope(name: str):
    """Context manager: a roctx range + record_function scope called `name` when tracing is enabled, else a shared no-op."""
    return _Range(name) if _ENABLED else _NOOP
