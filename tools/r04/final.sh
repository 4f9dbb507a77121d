#!/bin/bash
# the last build of round 4: the whole -m gpu suite, the driver's bench command, its kernel stats, the multi-device legs as a functional check
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_final; mkdir -p $OUT; rm -f $OUT/*
cd $R
( timeout 500 python -m pytest tests -m gpu -x -q --timeout 180 2>&1 | tail -8 ) > $OUT/pytest_gpu.log 2>&1
( timeout 700 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log ); echo "bench rc $?" >> $OUT/bench_err.log
PROF_NO_PMC=1 timeout 200 tools/prof_cmd.sh r04f python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --no-e2e --no-binary-legs > $OUT/prof.log 2>&1
( COVERM_BENCH_MULTI_DEVICE_CHECK=0,0 timeout 300 python bench.py --reads 2000000 --e2e-reads 4000000 --no-e2e --no-binary-legs --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps(d.get('multi_device_end_to_end_functional_check'), indent=1))" ) > $OUT/multi_device_check.json 2>&1
cat $OUT/pytest_gpu.log; tail -c 1200 $OUT/bench_line.json; tail -2 $OUT/bench_err.log; head -c 1500 $OUT/multi_device_check.json
