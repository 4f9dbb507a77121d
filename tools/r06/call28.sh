#!/bin/bash
# round 6, call 28: the launcher / child split of coverm-amd — CLI tests, then wall times at 200 M reads against one process
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call28
timeout 1500 python -m pytest $R/tests/test_cli_binary.py $R/tests/test_gpu_configs.py $R/tests/test_genes.py $R/tests/test_filter_subcommand.py $R/tests/test_gpu_bounded_store.py -q -m gpu -x 2>&1 | tail -6 > $R/gpurun_out/r06_call28/pytest.log
cat $R/gpurun_out/r06_call28/pytest.log
FEED_AB_EXIT=1 timeout 1200 python $R/tools/r06/feed_ab.py 200000000 4 $R/gpurun_out/r06_call28/exit_ab_200M.json 2>&1 | tee $R/gpurun_out/r06_call28/exit_ab_200M.log | tail -32
