#!/bin/bash
# round 6, call 25: the ways of zapping the mapping against pread, end to end: 100 M then 200 M reads, three rounds each
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call25
FEED_AB_ZAPS=1 timeout 900 python $R/tools/r06/feed_ab.py 100000000 3 $R/gpurun_out/r06_call25/feed_ab_100M.json 2>&1 | tee $R/gpurun_out/r06_call25/feed_ab_100M.log | grep -v "^{'mode'" | tail -32
FEED_AB_ZAPS=1 timeout 1200 python $R/tools/r06/feed_ab.py 200000000 3 $R/gpurun_out/r06_call25/feed_ab_200M.json 2>&1 | tee $R/gpurun_out/r06_call25/feed_ab_200M.log | grep -v "^{'mode'" | tail -32
