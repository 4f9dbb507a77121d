#!/bin/bash
# round 6, call 34: hunting the slow-feed runs (feed calls 0.18 s instead of 0.06 s): default against one process, 200 M reads, five rounds, with the
# ingest's own breakdown (drain / launches / upload calls)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call34
FEED_AB_EXIT=1 timeout 1500 python $R/tools/r06/feed_ab.py 200000000 5 $R/gpurun_out/r06_call34/exit_ab_200M.json 2>&1 | tee $R/gpurun_out/r06_call34/exit_ab_200M.log | grep -v "^{'mode'" | tail -16
