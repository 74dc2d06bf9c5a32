#!/bin/bash
# Same-box A/B of the whole tree (Python + kernels): snapshot a git revision (default HEAD) into _ab_base/ (git-ignored, but it
# travels to the GPU box), build its library there; then on the box:  (cd _ab_base && python bench.py ...)  vs  python bench.py ...
set -e
cd /root/repo
REV=${1:-HEAD}
rm -rf _ab_base && mkdir _ab_base
git archive "$REV" osufusion_amd oracle bench.py | tar -x -C _ab_base
(cd _ab_base && python osufusion_amd/csrc/build.py > /dev/null)
ls -la _ab_base/osufusion_amd/csrc/libosuf_hip.so
