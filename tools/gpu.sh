#!/bin/bash
# build the HIP library (if stale) and run a command on the MI355X box:  tools/gpu.sh [--timeout S] '<command>'
set -e
cd /root/repo
python osufusion_amd/csrc/build.py >/dev/null
T=900
if [ "$1" == "--timeout" ]; then T=$2; shift 2; fi
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$1"
