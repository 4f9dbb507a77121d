#!/bin/bash
# round 6, the build as it ships: the whole -m gpu suite, smoke(), the driver's bench command, kernel stats + counters of the coverage step
# usage: tools/r06/final.sh <tag>   (profiles get the tag: r06a, r06b, ...)
TAG=${1:-r06a}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${TAG}_final; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 1500 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -12 ) > $OUT/pytest_gpu.log 2>&1
cat $OUT/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( timeout 1400 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log ); echo "bench rc $?" >> $OUT/bench_err.log
tail -c 1800 $OUT/bench_line.json; tail -4 $OUT/bench_err.log
timeout 900 tools/prof_bench.sh $TAG > $OUT/prof_bench.log 2>&1; tail -3 $OUT/prof_bench.log
[ -n "$FINAL_SWEEP" ] && { timeout 1700 python tools/r06/contig_sweep.py --out $R/gpurun_out/r06_contig_sweep.json > $OUT/sweep.log 2> $OUT/sweep.err; tail -8 $OUT/sweep.log; tail -3 $OUT/sweep.err; }
