#!/bin/bash
# Timing-only triage builds of the hand-placed attention backward: tools/build_variants.sh name=drop,list ...
#   -> osufusion_amd/csrc/libosuf_hip_<name>.so (select with OSUF_HIP_LIB; results are WRONG by construction, never shipped)
set -e
cd /root/repo
python osufusion_amd/csrc/build.py > /dev/null
for spec in "$@"; do
  name=${spec%%=*}; drop=${spec#*=}
  D=$(mktemp -d /tmp/osuf_var.XXXX)
  cp osufusion_amd/csrc/*.hip osufusion_amd/csrc/*.hpp osufusion_amd/csrc/*.inc "$D/"
  if [[ "$drop" == @* ]]; then python tools/gen_attn_bwd512.py ${drop#@} --out "$D/attn_bwd512_asm.inc" > /dev/null   # name=@--flag: generator flags instead of a drop list
  else python tools/gen_attn_bwd512.py --drop "$drop" --out "$D/attn_bwd512_asm.inc" > /dev/null; fi
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c "$D/attn.hip" -o "$D/attn.o" 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "osufusion_amd/csrc/libosuf_hip_$name.so" "$D/attn.o" $(ls osufusion_amd/csrc/build/*.o | grep -v attn.hip.o) &&
    echo "built libosuf_hip_$name.so (drop: $drop)"; rm -rf "$D" ) &
done
wait
