#!/bin/bash
# round 6, call 24: where the time outside main() goes (spawn -> main, exit -> reaped), 100 M reads, two pairs
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call24
timeout 900 python $R/tools/r06/feed_ab.py 100000000 2 $R/gpurun_out/r06_call24/feed_ab.json 2>&1 | tee $R/gpurun_out/r06_call24/feed_ab.log | tail -14
