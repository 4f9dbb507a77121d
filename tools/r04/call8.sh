#!/bin/bash
# the ingest tests (CG:B,I on the device), the driver's bench command, its kernel stats + counters, counters of the ingest kernels
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call8; mkdir -p $OUT; rm -f $OUT/*
cd $R
( timeout 200 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --timeout 90 2>&1 | tail -6 ) > $OUT/pytest_ingest.log 2>&1
( timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log ); echo "bench rc $?" >> $OUT/bench_err.log
timeout 400 tools/prof_bench.sh r04a --no-e2e --no-binary-legs > $OUT/prof_bench.log 2>&1
PROF_PMC=1 timeout 300 tools/prof_ingest.sh r04ing 20000000 > $OUT/prof_ingest.log 2>&1
cat $OUT/pytest_ingest.log; tail -c 1500 $OUT/bench_line.json; tail -3 $OUT/bench_err.log
