#!/bin/bash
# the last build of round 4: the whole -m gpu suite, the driver's bench command, its kernel stats
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_final; mkdir -p $OUT; rm -f $OUT/*
cd $R
( timeout 500 python -m pytest tests -m gpu -x -q --timeout 180 2>&1 | tail -8 ) > $OUT/pytest_gpu.log 2>&1
( timeout 700 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log ); echo "bench rc $?" >> $OUT/bench_err.log
PROF_NO_PMC=1 timeout 200 tools/prof_cmd.sh r04f python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --no-e2e --no-binary-legs > $OUT/prof.log 2>&1
PROF_PMC= timeout 200 tools/prof_ingest.sh r04fing 20000000 > $OUT/prof_ingest.log 2>&1
cat $OUT/pytest_gpu.log; tail -c 1300 $OUT/bench_line.json; tail -2 $OUT/bench_err.log
