#!/bin/bash
# round 6, call 9: one mask per wave for the steps k_prep_lean leaves to k_prep_generic (list: 0.349 ms; flag per step: 0.404); tests; per-kernel
# trace; the contig sweep (device-resident steps + the binary over BAM files with 5 000 / 200 000 / 2 000 000 references)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call9; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests/test_gpu_abi_parity.py tests/test_gpu_estimates.py tests/test_gpu_bounded_store.py -x -q -m gpu --timeout 300 2>&1 | tail -6 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
timeout 600 python tools/r06/kernel_ab.py --variant lean8= --variant lean7=COVERM_PREP_KERNEL:17 > $OUT/ab_5k.log 2> $OUT/ab_5k.err; echo "ab exit $?" >> $OUT/ab_5k.log
tail -5 $OUT/ab_5k.log; tail -5 $OUT/ab_5k.err
PROF_SKIP_PMC=1 bash tools/r06/prof_ab.sh masks --variant default=
timeout 1700 python tools/r06/contig_sweep.py --out $R/gpurun_out/r06_contig_sweep.json > $OUT/sweep.log 2> $OUT/sweep.err; tail -30 $OUT/sweep.log; tail -5 $OUT/sweep.err
