#!/bin/bash
# round 3, GPU call 2: device pair filter tests, ingest tests with the mmap path, e2e wall times per I/O mode (50 M and 200 M reads)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_call2; mkdir -p $OUT
cd $R
( time timeout 900 python -m pytest tests/test_gpu_pair_filter.py tests/test_gpu_ingest.py tests/test_cli_binary.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
for io in mmap pread; do
  COVERM_INGEST_IO=$io PROBE_REPS=3 PROBE_NO_CPU=1 PROBE_SLEEP=2 timeout 600 python tools/e2e_probe.py 50000000 16 > $OUT/e2e_50M_$io.log 2>&1
done
COVERM_INGEST_IO=mmap PROBE_REPS=3 PROBE_NO_CPU=1 PROBE_SLEEP=3 timeout 900 python tools/e2e_probe.py 200000000 16 > $OUT/e2e_200M_mmap.log 2>&1
grep -h "wall\|spawn\|device ingest:\|main:" $OUT/e2e_*.log
