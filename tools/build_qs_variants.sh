#!/bin/bash
# Placement-knob variants of the PRE-SCALED-QUERY attention-backward loop (legitimate streams: same results, other schedules):
#   tools/build_qs_variants.sh name="--lat 4 --look 3" ...   ->  osufusion_amd/csrc/libosuf_hip_<name>.so  (select with OSUF_HIP_LIB)
set -e
cd /root/repo
python osufusion_amd/csrc/build.py > /dev/null
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  D=$(mktemp -d /tmp/osuf_qsv.XXXX)
  cp osufusion_amd/csrc/*.hip osufusion_amd/csrc/*.hpp osufusion_amd/csrc/*.inc "$D/"
  python tools/gen_attn_bwd512.py --qs $flags --out "$D/attn_bwd512qs_asm.inc" > /dev/null
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c "$D/attn.hip" -o "$D/attn.o" 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "osufusion_amd/csrc/libosuf_hip_$name.so" "$D/attn.o" $(ls osufusion_amd/csrc/build/*.o | grep -v attn.hip.o) &&
    echo "built libosuf_hip_$name.so ($flags)"; rm -rf "$D" ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
