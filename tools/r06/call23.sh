#!/bin/bash
# round 6, call 23: the staging slots filled from a mapping with non-temporal stores — ingest tests, then end to end at 200 M reads, alternating
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call23
timeout 1200 python -m pytest $R/tests/test_gpu_ingest.py $R/tests/test_gpu_bounded_store.py $R/tests/test_cli_binary.py -q -m gpu -x 2>&1 | tail -6 > $R/gpurun_out/r06_call23/pytest.log
cat $R/gpurun_out/r06_call23/pytest.log
timeout 1200 python $R/tools/r06/feed_ab.py 200000000 4 $R/gpurun_out/r06_call23/feed_ab.json 2>&1 | tee $R/gpurun_out/r06_call23/feed_ab.log | tail -30
