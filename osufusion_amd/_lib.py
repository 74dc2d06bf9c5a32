"""ctypes binding of libosuf_hip.so (C ABI declared in include/osufusion_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the library cannot be loaded the first kernel call
raises ``RuntimeError`` (loudly), it never silently computes something else.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_long, c_void_p
from pathlib import Path

_CSRC = Path(__file__).resolve().parent / "csrc"
# OSUF_HIP_LIB: load another build of the same C ABI (same-box A/B timing of two kernel revisions; tools/build_base.sh)
LIB_PATH = Path(os.environ["OSUF_HIP_LIB"]).resolve() if os.environ.get("OSUF_HIP_LIB") else _CSRC / "libosuf_hip.so"

P, L, I, F = c_void_p, c_long, c_int, c_float

# name -> argtypes (all return int).  Must list every symbol of include/osufusion_hip.h (tests/test_host_logic.py::test_capi_* check).
SIGNATURES = {
    "osuf_version": [],
    "osuf_gemm_nt": [I, P, L, P, L, L, P, L, P, L, P, L, P, L, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "osuf_gemm_nt_rowdot": [I, P, L, P, L, P, L, P, L, P, I, I, I, I, I, P],
    "osuf_gemm_tn": [I, P, L, P, L, P, L, L, I, I, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "osuf_gemm_tn_bias": [I, P, L, P, L, P, L, L, I, I, I, I, I, I, I, I, I, I, I, I, P, L, P, P],
    "osuf_gemm_tn_workspace_bytes": [I, I, I, I, I],
    "osuf_colsum": [I, P, L, I, I, P, P],
    "osuf_gn_finalize": [P, P, I, L, P],
    "osuf_gn_apply_fwd": [I, P, L, P, L, P, P, P, P, I, I, I, P],
    "osuf_gn_apply_fwd_stats": [I, P, L, P, L, P, L, P, P, P, P, I, I, I, P],
    "osuf_gn_bwd": [I, P, L, P, L, P, L, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "osuf_ln_fwd": [I, P, L, P, L, P, P, P, I, I, P],
    "osuf_ln_bwd": [I, P, L, P, L, P, L, P, P, P, P, I, I, P],
    "osuf_rowdot": [I, P, L, P, L, P, P, I, I, I, P],
    "osuf_softmax_rows": [P, I, I, P],
    "osuf_wcolsum": [I, P, L, P, L, P, P, I, I, I, P, P],
    "osuf_gn_stats": [I, P, L, P, P, I, I, I, P],
    "osuf_gca_pool": [I, P, L, P, P, P, P, P, I, I, I, P],
    "osuf_gca_pool_workspace_bytes": [I, I, I],
    "osuf_gn_stats_parts": [I, P, L, P, I, I, I, P],
    "osuf_gn_apply_fwd_parts": [I, P, L, P, L, P, P, P, P, P, I, I, I, P],
    "osuf_gn_stats_workspace_bytes": [I, I, I],
    "osuf_gate_residual": [I, P, L, P, P, L, P, L, I, I, I, P],
    "osuf_gca_bwd_apply": [I, P, L, P, L, P, L, P, P, P, P, P, P, I, I, I, P, P, P, L, P],
    "osuf_gca_bwd_apply_workspace_bytes": [I, I],
    "osuf_rope_cast": [I, P, L, P, L, P, P, I, I, I, I, I, P],
    "osuf_rope_cast_qs": [I, P, L, P, L, P, P, I, I, I, I, I, F, I, P],
    "osuf_rope_bwd": [I, P, L, P, L, P, P, I, I, I, I, I, P],
    "osuf_mqa_fwd": [P, L, P, L, P, L, P, L, I, P, I, I, I, I, F, P],
    "osuf_mqa_fwd_qs": [P, L, P, L, P, L, P, L, I, P, I, I, I, I, F, P],
    "osuf_mqa_fwd_zdq": [P, L, P, L, P, L, P, L, I, P, I, I, I, I, F, I, P, P],
    "osuf_mqa_fwd_rope": [P, L, P, L, P, L, P, L, I, P, I, I, I, I, F, P, P, F, P, L, P, P],
    "osuf_mqa_fwd_masked": [P, L, P, L, P, L, P, L, I, P, P, L, L, L, L, I, I, I, I, F, P],
    "osuf_attn_delta": [P, L, P, L, I, P, I, I, I, I, P],
    "osuf_mqa_bwd_dq": [P, L, P, L, P, L, P, L, P, P, P, L, I, I, I, I, F, I, P, P, I, P],
    "osuf_mqa_bwd_dkv": [P, L, P, L, P, L, P, L, P, P, P, P, L, I, I, I, I, F, I, P, P, P, L, I, I, P],
    "osuf_mqa_bwd_dkv_workspace_bytes": [I, I, I],
    "osuf_mqa_bwd_fused": [P, L, P, L, P, L, P, L, P, P, P, L, P, P, L, I, I, I, I, F, I, P, P, P, L, I, I, P],
    "osuf_mqa_bwd_fused_qs": [P, L, P, L, P, L, P, L, P, P, P, L, P, P, L, I, I, I, I, F, I, P, P, P, L, I, I, P],
    "osuf_mqa_bwd_fused_workspace_bytes": [I, I, I, I, I, I],
    "osuf_ncl_to_rows": [I, P, P, L, I, I, I, I, I, P],
    "osuf_rows_to_ncl": [I, P, L, P, I, I, I, P],
    "osuf_copy2d": [I, P, L, I, P, L, I, I, P],
    "osuf_add2d": [I, P, L, P, L, P, L, I, I, P],
    "osuf_axpby_rows": [P, P, P, P, P, I, L, P],
    "osuf_ddim_step": [P, P, P, F, P, P, I, L, P],
    "osuf_mse": [P, P, P, P, P, I, I, I, P],
    "osuf_sqnorm": [P, L, P, P],
    "osuf_adamw": [P, P, P, P, L, F, F, F, F, F, I, P, P],
    "osuf_clip_coef": [P, F, F, P, P, P],
    "osuf_cast_f32_bf16": [P, P, L, P],
    "osuf_pack_weight": [P, I, I, I, I, P, L, L, P, L, L, I, P],
    "osuf_pack_weight_group": [P, I, I, I, P],
    "osuf_dora_effective": [P, P, P, P, I, I, I, F, P, P, P],
    "osuf_dora_gain": [P, P, P, P, I, I, I, I, F, P, P, P, P, P],
    "osuf_adapter_finish": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "osuf_pack_weight_adapted": [P, P, P, P, F, I, I, I, I, I, P, L, L, P, L, L, I, P],
    "osuf_clock_probe": [I, I, I, P, P],
    "osuf_log_vqt": [P, L, P, I, I, I, P, F, P, P, L, L, P],
    "osuf_vqt_logmag": [P, L, P, L, P, I, L, F, P],
    "osuf_fir_decimate2": [P, L, P, I, P, L, P],
    "osuf_frame_rows": [P, L, I, I, P, L, P],
    "osuf_skinny_fwd": [I, P, L, P, P, P, L, I, I, I, I, I, P],
    "osuf_skinny_bwd": [I, P, L, P, L, P, L, P, P, L, P, P, I, I, I, I, I, I, P],
    "osuf_skinny_fwd_group": [I, P, L, P, I, I, I, I, I, P],
    "osuf_skinny_dx_group": [I, P, I, I, P, L, P, L, I, I, I, P],
}

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load (building in-tree with hipcc if it is absent and a compiler exists) the HIP library."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists() and build_if_missing:
        try:
            from .csrc import build as _build
            _build.build()
        except Exception as e:  # noqa: BLE001
            raise HipExtensionMissing(f"libosuf_hip.so is missing and could not be built: {e}") from e
    if not LIB_PATH.exists():
        raise HipExtensionMissing(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        lib = ctypes.CDLL(str(LIB_PATH), mode=os.RTLD_NOW | os.RTLD_LOCAL)
    except OSError as e:
        raise HipExtensionMissing(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipExtensionMissing(f"{LIB_PATH} does not export {name} (stale or partial build of the C ABI)") from e
        fn.argtypes = argtypes
        fn.restype = c_long if name.endswith("_bytes") else c_int
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        kind = {-1: "invalid argument", -2: "unsupported configuration"}.get(status, f"hipError {status}")
        raise RuntimeError(f"{what} failed: {kind}")
