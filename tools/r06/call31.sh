#!/bin/bash
# round 6, call 31: bytes per staging piece (= per H2D copy) now that the link is what the reader waits for: 200 M reads, three rounds
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call31
FEED_AB_PIECES=1 timeout 1500 python $R/tools/r06/feed_ab.py 200000000 3 $R/gpurun_out/r06_call31/pieces_200M.json 2>&1 | tee $R/gpurun_out/r06_call31/pieces_200M.log | grep -v "^{'mode'" | tail -30
