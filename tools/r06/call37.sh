#!/bin/bash
# round 6, call 37: blocks per round again, now that the feed runs at the link and the kernels are faster than in round 4's sweep
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call37
FEED_AB_ROUNDS=1 timeout 1500 python $R/tools/r06/feed_ab.py 200000000 3 $R/gpurun_out/r06_call37/rounds_200M.json 2>&1 | tee $R/gpurun_out/r06_call37/rounds_200M.log | grep -v "^{'mode'" | tail -30
