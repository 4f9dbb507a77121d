#!/bin/bash
# round 6, call 40: COV_WANT_INGEST (the ingest's streams and events created from cov_create on) — tests, then 200 M reads: first upload after?
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call40
timeout 1500 python -m pytest $R/tests/test_gpu_ingest.py $R/tests/test_gpu_bounded_store.py $R/tests/test_cli_binary.py $R/tests/test_gpu_configs.py $R/tests/test_gpu_pair_filter.py $R/tests/test_genes.py -q -m gpu -x 2>&1 | tail -9 | head -4 | tee $R/gpurun_out/r06_call40/pytest.log
timeout 1200 python $R/tools/r06/feed_ab.py 200000000 3 $R/gpurun_out/r06_call40/feed_ab.json 2>&1 | tee $R/gpurun_out/r06_call40/feed_ab.log | grep "^{'mode'" | sed "s/'bytes_from.*//; s/'vm_hwm_mb.*'sessions_s'/'sessions_s'/" | cut -c1-360
