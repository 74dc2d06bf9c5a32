#!/bin/bash
# SQ counter passes (where do the wave cycles go?) over a stand-alone timing script (GPU box, repo root):
#   tools/pmc_sq.sh <tag> <kernel-name regex> <script.py> [args]  ->  gpurun_out/<tag>_sq_{a,b}.csv  (rows of the matching kernels only)
set -e
TAG=$1; RE=$2; shift 2
ROOT=$(pwd)
SCRIPT=$ROOT/$1; shift
ARGS=("$@")
cd /tmp && export TMPDIR=/tmp
for pass in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "b SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE"; do
  set -- $pass
  name=$1; shift
  out="$ROOT/gpurun_out/pmc_${TAG}_sq_$name"
  mkdir -p "$out"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -- python3 "$SCRIPT" "${ARGS[@]}" > "$out/run.log" 2> "$out/run.err" || { tail -5 "$out/run.err"; exit 1; }
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  (head -1 "$f"; grep -E "$RE" "$f") > "$ROOT/gpurun_out/${TAG}_sq_${name}.csv"
  wc -l "$ROOT/gpurun_out/${TAG}_sq_${name}.csv"
done
