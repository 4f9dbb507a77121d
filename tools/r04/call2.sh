#!/bin/bash
# Second GPU call of round 4 (≈ 3 min): end to end at 200 M reads, k_inflate against k_inflate_wave with windows of 81920 .. 10240 blocks.
#   gpurun --timeout 420 -- tools/r04/call2.sh [stores]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call2; mkdir -p $OUT
cd $R
SWEEP_STORES=${1:-2} timeout 400 python tools/r04/window_sweep.py 200000000 16 > $OUT/window_sweep.log 2>&1
cat $OUT/window_sweep.log
