#!/bin/bash
# rocprofv3 kernel statistics of the headline benchmark (run on the GPU box from the repo root):
#   tools/profile_bench.sh <tag> [bench.py args]   ->  gpurun_out/<tag>_kernel_stats.csv (+ the bench line, <tag>_bench.json)
# (cd /tmp + TMPDIR=/tmp as the pool's guide asks; the program after `--` is python itself, no wrapper in between.)
set -e
TAG=$1; shift
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out/prof_$TAG"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$TAG" -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-sampler --no-config5 "$@" > "$ROOT/gpurun_out/${TAG}_bench.json" 2> "$ROOT/gpurun_out/${TAG}_bench.err" || { tail -5 "$ROOT/gpurun_out/${TAG}_bench.err"; exit 1; }
F=$(find "$ROOT/gpurun_out/prof_$TAG" -name "*kernel_stats.csv" | head -1)
cp "$F" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
head -25 "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" | cut -c1-160
