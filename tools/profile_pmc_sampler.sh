#!/bin/bash
# Counter evidence for BASELINE config 4 (the DDIM sampler at B=16, L=8192): kernel statistics + FETCH / WRITE / MFMA-busy passes over
# tools/sampler_short.py (S=3), each in its own run, the program directly after `--`.   tools/profile_pmc_sampler.sh <tag>
set -e
TAG=$1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
out="$ROOT/gpurun_out/pmc_${TAG}_sampler"
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$ROOT/tools/sampler_short.py" > "$out/stats_run.json" 2> "$out/stats_run.err" || { tail -5 "$out/stats_run.err"; exit 1; }
cp "$(find "$out/stats" -name "*kernel_stats.csv" | head -1)" "$out/kernel_stats.csv"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  set -- $pass
  name=$1; shift
  mkdir -p "$out/$name"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- python3 "$ROOT/tools/sampler_short.py" > "$out/${name}_run.json" 2> "$out/${name}_run.err" || { tail -5 "$out/${name}_run.err"; exit 1; }
  f=$(find "$out/$name" -name "*counter_collection.csv" | head -1)
  (head -1 "$f"; grep -E "mqa_|gemm_|gn_|ln_|rope|gate|wcolsum|softmax|rowdot" "$f") > "$out/${name}_counter_collection.csv"
  rm -rf "$out/$name"
  wc -l "$out/${name}_counter_collection.csv"
done
rm -rf "$out/stats"
python3 "$ROOT/tools/pmc_sampler_summary.py" "$out" "$TAG" > "$out/summary.txt"
cat "$out/summary.txt"
