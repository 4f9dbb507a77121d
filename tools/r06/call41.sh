#!/bin/bash
# round 6, call 41: COV_WANT_INGEST against the streams created inside cov_ingest_begin, one box, 200 M reads, five rounds
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call41
FEED_AB_PREPARE=1 timeout 1500 python $R/tools/r06/feed_ab.py 200000000 5 $R/gpurun_out/r06_call41/prepare_ab_200M.json 2>&1 | tee $R/gpurun_out/r06_call41/prepare_ab_200M.log | grep -v "^{'mode'" | tail -16
