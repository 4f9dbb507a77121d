#!/bin/bash
# round 6, call 29: which part of the feed calls is slow in the slow runs (200 M reads, default against pread, five rounds)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call29
timeout 1500 python $R/tools/r06/feed_ab.py 200000000 5 $R/gpurun_out/r06_call29/feed_ab_200M.json 2>&1 | tee $R/gpurun_out/r06_call29/feed_ab_200M.log | tail -30
