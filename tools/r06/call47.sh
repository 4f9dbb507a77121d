#!/bin/bash
# round 6, call 47: many-contig start-up and finish (header read, result staging obtained beside the ingest) — tests with many contigs, then the 2 M timing
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call47
timeout 1500 python -m pytest $R/tests/test_gpu_abi_parity.py $R/tests/test_gpu_estimates.py $R/tests/test_cli_binary.py $R/tests/test_gpu_configs.py $R/tests/test_gpu_bounded_store.py -q -m gpu -x 2>&1 | tail -9 | head -4
timeout 900 python $R/tools/r06/two_million_timing.py 2>&1 | tee $R/gpurun_out/r06_call47/timing.log | grep "wall\|open\|main:" | cut -c1-250
