#!/bin/bash
# chunk (tiles per wave between flushes) and grid sweep of k_pileup_fast after the flush changed: environment only, no rebuild
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call19; mkdir -p $OUT; rm -f $OUT/*
cd $R
run() {  # label, env...
  local label=$1; shift
  ( env "$@" timeout 100 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --no-e2e --no-binary-legs 2> $OUT/err_$label.log | tail -1 > $OUT/line_$label.json )
  python - "$label" $OUT/line_$label.json <<'P' >> $OUT/sweep.log
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read())
    k = d["roofline"]["all_kernels_ms"]
    print("%-14s ms_per_step %.3f  k_pileup %.4f  k_prep %.4f  k_ranges %.4f  k_hist %.4f" % (sys.argv[1], d["ms_per_step"], k.get("k_pileup", 0), k.get("k_prep", 0), k.get("k_ranges", 0), k.get("k_hist", 0)))
except Exception as e:
    print(sys.argv[1], "failed", e)
P
}
run chunk8 COVERM_CHUNK=8
run chunk16 COVERM_CHUNK=16
run chunk32 COVERM_CHUNK=32
run chunk4 COVERM_CHUNK=4
run chunk8_wg24 COVERM_CHUNK=8 COVERM_WG_PER_CU=24
run chunk8_wg96 COVERM_CHUNK=8 COVERM_WG_PER_CU=96
cat $OUT/sweep.log
