#!/bin/bash
# Second half of the first measurements of round 4 (≈ 10 min): the ingest tests under COVERM_INFLATE_V=3, replicated histogram bins in k_pileup_fast.
#   gpurun --timeout 1000 -- tools/r04/call1b.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call1b; mkdir -p $OUT
cd $R
( COVERM_INFLATE_V=3 timeout 240 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --timeout 120 2>&1 | tail -5 ) > $OUT/pytest_v3.log 2>&1
# replicated histogram bins in k_pileup_fast: parity (the coverage parity tests) and time
for r in 2 4; do
  ( COVERM_PILEUP_HREP=$r timeout 200 python -m pytest tests/test_gpu_abi_parity.py -m gpu -x -q --timeout 120 2>&1 | tail -3 ) > $OUT/pytest_hrep$r.log 2>&1
  ( COVERM_PILEUP_HREP=$r timeout 120 python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>&1 | tail -2 ) > $OUT/bench_hrep$r.log 2>&1
done
cat $OUT/pytest_v3.log $OUT/pytest_hrep*.log $OUT/bench_hrep*.log
