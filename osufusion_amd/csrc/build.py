#!/usr/bin/env python3
"""Build libosuf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output:
osufusion_amd/csrc/libosuf_hip.so -- git-ignored, but it travels to the GPU box with the snapshot."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
SOURCES = ["gemm.hip", "norm.hip", "attn.hip", "elementwise.hip", "skinny.hip", "audio.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
LIB = HERE / "libosuf_hip.so"


ATTN_SOURCES = ["attn.hip", "attn_bwd512_asm.inc", "attn_bwd512qs_asm.inc", "attn_generic.hpp", "common.hpp"]


def source_hash(names=ATTN_SOURCES) -> str:
    """sha256 over the text of the named kernel sources: what a committed counter summary is stamped with (profiles/pmc_manifest.json),
    so that bench.py can tell whether the kernels it times are still the kernels those counters were collected on."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        h.update(n.encode() + b"\0" + (HERE / n).read_bytes() + b"\0")
    return h.hexdigest()


def _stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [HERE / s for s in SOURCES] + [HERE / "common.hpp", HERE / "attn_generic.hpp", HERE / "attn_bwd512_asm.inc", HERE / "attn_bwd512qs_asm.inc", Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = HERE / "build"
    objdir.mkdir(exist_ok=True)

    def compile_one(src: str) -> Path:
        obj = objdir / (src + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(HERE / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
