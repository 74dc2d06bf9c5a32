#!/bin/bash
# SQ counter passes over the stand-alone attention-backward timing (GPU box, repo root): where do the 512-key sweep's wave cycles go?
#   tools/pmc_attn_sq.sh <tag>  ->  gpurun_out/<tag>_sq_{a,b}.csv  (rows of the fused kernels only)
set -e
TAG=$1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pass in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "b SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE"; do
  set -- $pass
  name=$1; shift
  out="$ROOT/gpurun_out/pmc_${TAG}_sq_$name"
  mkdir -p "$out"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -- python3 "$ROOT/tools/bench_attn_bwd.py" 4096 > "$out/run.log" 2> "$out/run.err" || { tail -5 "$out/run.err"; exit 1; }
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  (head -1 "$f"; grep -E "mqa_bwd_fused" "$f") > "$ROOT/gpurun_out/${TAG}_sq_${name}.csv"
  wc -l "$ROOT/gpurun_out/${TAG}_sq_${name}.csv"
done
