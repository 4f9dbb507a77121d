#!/bin/bash
# round 5, second GPU call: k_prep2 without a second path inside its loop, against k_prep, alternating; the whole suite
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call2; mkdir -p $OUT; rm -f $OUT/*; cd $R
( timeout 900 python -m pytest tests -q -m gpu --timeout 300 2>&1 | tail -25 ) > $OUT/pytest_gpu.log 2>&1
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
done
cat $OUT/ab.log; cat $OUT/pytest_gpu.log
