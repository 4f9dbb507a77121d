#!/bin/bash
# k_pileup_fast7 (384 LDS bins, registers capped for seven waves per SIMD) against k_pileup_fast on one box, then the whole -m gpu suite with it
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call20; mkdir -p $OUT; rm -f $OUT/*
cd $R
run() {
  local label=$1; shift
  ( env "$@" timeout 70 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --no-e2e --no-binary-legs 2> $OUT/err_$label.log | tail -1 > $OUT/line_$label.json )
  python - "$label" $OUT/line_$label.json <<'P' >> $OUT/sweep.log
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read())
    k = d["roofline"]["all_kernels_ms"]
    print("%-14s ms_per_step %.3f  k_pileup %.4f  k_prep %.4f" % (sys.argv[1], d["ms_per_step"], k.get("k_pileup", 0), k.get("k_prep", 0)))
except Exception as e:
    print(sys.argv[1], "failed", e)
P
}
run six_a COVERM_CHUNK=8
run seven_a COVERM_FAST_WAVES=7
run six_b COVERM_CHUNK=8
run seven_b COVERM_FAST_WAVES=7
cat $OUT/sweep.log
( COVERM_FAST_WAVES=7 timeout 160 python -m pytest tests -m gpu -x -q --timeout 120 2>&1 | tail -6 ) > $OUT/pytest_gpu_seven.log 2>&1
cat $OUT/pytest_gpu_seven.log
