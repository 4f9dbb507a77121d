#!/bin/bash
# round 6, call 35: short rounds over the file's last bytes (the run's tail), 200 M reads, four rounds
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call35
COVERM_KNOBS=ingest_tail_mb=3 timeout 900 python -m pytest $R/tests/test_gpu_ingest.py -q -m gpu -x -k "equals_cpu_reader or bit_exact or four_feeders or ragged" 2>&1 | tail -3
FEED_AB_TAIL=1 timeout 1500 python $R/tools/r06/feed_ab.py 200000000 4 $R/gpurun_out/r06_call35/tail_200M.json 2>&1 | tee $R/gpurun_out/r06_call35/tail_200M.log | grep -v "^{'mode'" | tail -30
