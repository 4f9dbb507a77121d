#!/bin/bash
# device listing of libcovermhip's kernels + per-kernel register / spill summary (round 6's instruction diet works from these)
#   tools/r06/isa.sh [kernel-name-substring ...]   -> /tmp/isa/dev.s, /tmp/isa/<substring>.s
mkdir -p /tmp/isa && cd /tmp/isa || exit 1
R=/root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include --cuda-device-only -S -o dev.s -x hip $R/coverm_amd/csrc/covermhip.hip -Wall -Wno-unused-function -Wno-pass-failed 2>&1 | grep -v "hip-link" | head -30
for k in "$@"; do
  python $R/tools/r06/isa_fn.py dev.s $k --dump $k.s
  awk -v k="$k" '$0 ~ "\\.name:.*"k {f=1} f&&/(\.sgpr_count|sgpr_spill|\.vgpr_count|vgpr_spill|private_segment_fixed)/{printf "%s ", $0} f&&/vgpr_spill_count/{print ""; exit}' dev.s
done
