#!/bin/bash
# round 6, call 38: rounds of 61 440 blocks as the default — ingest / bounded store / CLI tests
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call38
timeout 1500 python -m pytest $R/tests/test_gpu_ingest.py $R/tests/test_gpu_bounded_store.py $R/tests/test_cli_binary.py $R/tests/test_gpu_configs.py $R/tests/test_gpu_pair_filter.py $R/tests/test_genes.py -q -m gpu 2>&1 | tail -12 | tee $R/gpurun_out/r06_call38/pytest.log
