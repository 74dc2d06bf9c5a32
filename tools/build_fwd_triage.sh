#!/bin/bash
# Timing-only triage builds of the forward attention kernel (results are WRONG by construction): tools/build_fwd_triage.sh NAME=MACRO[,MACRO] ...
#   MACROs: OSUF_FWD_TRIAGE_NOLOAD (K / V tile never re-staged), OSUF_FWD_TRIAGE_NOBARRIER, OSUF_FWD_TRIAGE_NOEXP  -> libosuf_hip_<NAME>.so
set -e
cd /root/repo
python osufusion_amd/csrc/build.py > /dev/null
for spec in "$@"; do
  name=${spec%%=*}; defs=""; IFS=',' read -ra M <<< "${spec#*=}"; for m in "${M[@]}"; do defs="$defs -D$m"; done
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $defs -c osufusion_amd/csrc/attn.hip -o /tmp/attn_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "osufusion_amd/csrc/libosuf_hip_$name.so" /tmp/attn_$name.o $(ls osufusion_amd/csrc/build/*.o | grep -v attn.hip.o) &&
    echo "built libosuf_hip_$name.so ($defs)" ) &
done
wait
