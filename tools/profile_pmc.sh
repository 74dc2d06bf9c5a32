#!/bin/bash
# rocprofv3 counter passes over the headline benchmark (GPU box, repo root): one pass per counter set, each in its own run
# (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only.
#   tools/profile_pmc.sh <tag>   ->  gpurun_out/<tag>_{fetch,write,mfma}_counter_collection.csv
set -e
TAG=$1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  set -- $pass
  name=$1; shift
  out="$ROOT/gpurun_out/pmc_${TAG}_$name"
  mkdir -p "$out"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-sampler --no-config5 > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  # keep only the attention / GEMM kernels' rows: the full file has one row per dispatch and counter
  (head -1 "$f"; grep -E "mqa_|gemm_" "$f") > "$ROOT/gpurun_out/${TAG}_${name}_counter_collection.csv"
  wc -l "$ROOT/gpurun_out/${TAG}_${name}_counter_collection.csv"
done
