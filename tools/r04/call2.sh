#!/bin/bash
# Second GPU call of round 4: the whole -m gpu suite with the wave inflate as the default (and the lane-per-block kernel as the second
# implementation in tests/test_gpu_ingest.py), replicated histogram bins in k_pileup_fast, end to end at 200 M reads over window sizes.
#   gpurun --timeout 900 -- tools/r04/call2.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call2; mkdir -p $OUT
cd $R
( timeout 420 python -m pytest tests -m gpu -x -q --timeout 180 2>&1 | tail -8 ) > $OUT/pytest_gpu.log 2>&1
for r in 1 2 4; do
  ( COVERM_PILEUP_HREP=$r timeout 120 python bench.py --no-cpu-baseline --no-e2e --no-binary-legs --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('HREP $r', d.get('ms_per_step'), d.get('roofline'), d.get('kernel_ms'))" ) > $OUT/bench_hrep$r.log 2>&1
done
timeout 300 python tools/r04/window_sweep.py 200000000 16 > $OUT/window_sweep.log 2>&1
cat $OUT/pytest_gpu.log $OUT/bench_hrep*.log $OUT/window_sweep.log
