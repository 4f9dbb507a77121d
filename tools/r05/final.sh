#!/bin/bash
# round 5, the build as it ships: the whole -m gpu suite, the driver's bench command, kernel stats + counters of the coverage step (r05d) and of
# the ingest (r05ing3), smoke()
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_final; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests -q -m gpu --timeout 300 2>&1 | tail -12 ) > $OUT/pytest_gpu.log 2>&1
cat $OUT/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log ); echo "bench rc $?" >> $OUT/bench_err.log
tail -c 1600 $OUT/bench_line.json; tail -2 $OUT/bench_err.log
timeout 600 tools/prof_bench.sh r05d > $OUT/prof_bench.log 2>&1
PROF_PMC=1 timeout 900 tools/prof_ingest.sh r05ing3 20000000 > $OUT/prof_ingest.log 2>&1
tail -3 $OUT/prof_ingest.log
