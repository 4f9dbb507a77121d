#!/bin/bash
# round 5, third GPU call: the suite on the build with the closed-form CIGAR walk and the merged launches, k_prep6 against k_prep, the
# driver's own bench command, kernel stats + counters of the coverage step and of the ingest
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call3; mkdir -p $OUT; rm -f $OUT/*; cd $R
( timeout 900 python -m pytest tests -q -m gpu --timeout 300 2>&1 | tail -15 ) > $OUT/pytest_gpu.log 2>&1
export COVERM_BENCH_CACHE=/dev/shm
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>$OUT/err_$tag.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$tag', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v, 4) for k, v in r['all_kernels_ms'].items()})" >> $OUT/ab.log 2>&1; }
for rep in 1 2; do
  run default X=1
  run prep_w5 COVERM_PREP_WAVES=5
done
cat $OUT/ab.log; cat $OUT/pytest_gpu.log
unset COVERM_BENCH_CACHE
( timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log ); echo "bench rc $?" >> $OUT/bench_err.log
tail -c 1500 $OUT/bench_line.json; tail -2 $OUT/bench_err.log
timeout 600 tools/prof_bench.sh r05b > $OUT/prof_bench.log 2>&1
PROF_PMC=1 timeout 900 tools/prof_ingest.sh r05ing 20000000 > $OUT/prof_ingest.log 2>&1
