#!/bin/bash
# k_pileup_fast<true> reading the window statistics of interior tiles off the LDS histogram: the whole -m gpu suite, the bench line with its
# parity check (no end-to-end legs), kernel stats
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call18; mkdir -p $OUT; rm -f $OUT/*
cd $R
( timeout 400 python -m pytest tests -m gpu -x -q --timeout 180 2>&1 | tail -12 ) > $OUT/pytest_gpu.log 2>&1
( timeout 200 python bench.py --no-e2e --no-binary-legs > $OUT/bench_line.json 2> $OUT/bench_err.log ); echo "bench rc $?" >> $OUT/bench_err.log
PROF_NO_PMC=1 timeout 150 tools/prof_cmd.sh r04g python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --no-e2e --no-binary-legs > $OUT/prof.log 2>&1
cat $OUT/pytest_gpu.log; tail -c 1500 $OUT/bench_line.json; tail -2 $OUT/bench_err.log; grep -rh "k_pileup_fast\|k_prep<" $R/gpurun_out/prof_r04g/trace --include=*kernel_stats.csv 2>/dev/null | cut -c1-160 | head
