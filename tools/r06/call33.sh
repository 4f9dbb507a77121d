#!/bin/bash
# round 6, call 33: two upload streams taking the staging pieces in turn — ingest tests with the knob, then 200 M reads alternating
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call33
COVERM_KNOBS=ingest_copy_streams=2 timeout 900 python -m pytest $R/tests/test_gpu_ingest.py -q -m gpu -x -k "not windows" 2>&1 | tail -4 > $R/gpurun_out/r06_call33/pytest.log
cat $R/gpurun_out/r06_call33/pytest.log
FEED_AB_COPY_STREAMS=1 timeout 1500 python $R/tools/r06/feed_ab.py 200000000 4 $R/gpurun_out/r06_call33/copy_streams_200M.json 2>&1 | tee $R/gpurun_out/r06_call33/copy_streams_200M.log | grep -v "^{'mode'" | tail -20
