#!/bin/bash
# round 5, HEAD after k_bam_extract's four-record stores: the whole -m gpu suite, smoke(), the driver's bench command (r05e)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_final2; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests -q -m gpu --timeout 300 2>&1 | tail -5 ) > $OUT/pytest_gpu.log 2>&1
cat $OUT/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log ); echo "bench rc $?" >> $OUT/bench_err.log
tail -c 900 $OUT/bench_line.json; tail -1 $OUT/bench_err.log
