#!/bin/bash
# round 6, call 21: the ragged-block CRC test (three kernel combinations)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call21
timeout 1200 python -m pytest $R/tests/test_gpu_ingest.py -q -m gpu -x -k "ragged or irregular" 2>&1 | tail -12 > $R/gpurun_out/r06_call21/pytest.log
cat $R/gpurun_out/r06_call21/pytest.log
