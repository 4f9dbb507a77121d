#!/bin/bash
# round 5, first GPU call: the new tests, the whole -m gpu suite, k_prep / k_pileup_fast variants alternating on one box, counters of the defaults
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call1; mkdir -p $OUT; rm -f $OUT/*; cd $R
( timeout 700 python -m pytest tests/test_gpu_estimates.py tests/test_gpu_bounded_store.py tests/test_bench_launcher.py -x -q -m gpu --timeout 300 2>&1 | tail -25 ) > $OUT/pytest_new.log 2>&1
( timeout 700 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -12 ) > $OUT/pytest_gpu.log 2>&1
export COVERM_BENCH_CACHE=/dev/shm
run() { tag=$1; shift; extra=""; if [ "$1" = "--host-estimates" ]; then extra="--host-estimates"; shift; fi
  env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 $extra 2>$OUT/err_$tag.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$tag', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v, 4) for k, v in r['all_kernels_ms'].items()})" >> $OUT/ab.log 2>&1; }
for rep in 1 2; do
  run default X=1
  run host_estimates --host-estimates X=1
  run prep_v1 COVERM_PREP_V=1
  run prep_v1_w6 COVERM_PREP_V=1 COVERM_PREP_WAVES=6
  run prep_v2_w0 COVERM_PREP_WAVES=0
  run prep_v2_w6 COVERM_PREP_WAVES=6
  run prep_v2_w8 COVERM_PREP_WAVES=8
  run fast6 COVERM_FAST_WAVES=6
  run fast8 COVERM_FAST_WAVES=8
done
cat $OUT/ab.log
timeout 900 tools/prof_bench.sh r05a > $OUT/prof_bench.log 2>&1
cat $OUT/pytest_new.log; cat $OUT/pytest_gpu.log
