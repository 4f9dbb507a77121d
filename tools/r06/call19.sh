#!/bin/bash
# round 6, call 19: k_crc32_wave — the ingest tests, then the kernel trace of a 100 M-read run (compare with call 18's timeline)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call19
timeout 1200 python -m pytest $R/tests/test_gpu_ingest.py $R/tests/test_gpu_bounded_store.py -q -m gpu -x 2>&1 | tail -8 > $R/gpurun_out/r06_call19/pytest.log
cat $R/gpurun_out/r06_call19/pytest.log
bash $R/tools/prof_ingest.sh r06crc 100000000
python $R/tools/r06/ingest_timeline.py "$R/gpurun_out/prof_r06crc/trace/**/*kernel_trace.csv" $R/gpurun_out/prof_r06crc/timeline.json | head -45
