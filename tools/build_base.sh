#!/bin/bash
# Build the kernels of a git revision (default HEAD) into osufusion_amd/csrc/libosuf_hip_base.so for same-box A/B timing:
#   tools/build_base.sh [rev];  then on the GPU box:  OSUF_HIP_LIB=osufusion_amd/csrc/libosuf_hip_base.so python bench.py ...
set -e
cd /root/repo
REV=${1:-HEAD}
D=$(mktemp -d /tmp/osuf_base.XXXX)
# every source the revision has (the list follows the tree, not this script: an omitted file means missing symbols at load time)
for f in $(git ls-tree --name-only "$REV" osufusion_amd/csrc/ | grep -E "\.(hip|hpp|inc)$" | xargs -n1 basename); do git show "$REV:osufusion_amd/csrc/$f" > "$D/$f"; done
for f in $(cd "$D" && ls *.hip | sed s/.hip//); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$D/$f.hip" -o "$D/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o osufusion_amd/csrc/libosuf_hip_base.so "$D"/*.o
rm -rf "$D"
ls -la osufusion_amd/csrc/libosuf_hip_base.so
