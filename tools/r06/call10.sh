#!/bin/bash
# round 6, call 10: the steps k_prep_lean leaves go on k_prep_generic's list with one atomic per WORKGROUP; tests (with the reset-in-the-middle-of-an-ingest
# test); per-kernel trace at 5 000 contigs; the contig sweep
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call10; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests/test_gpu_abi_parity.py tests/test_gpu_estimates.py tests/test_gpu_bounded_store.py tests/test_gpu_configs.py -x -q -m gpu --timeout 300 2>&1 | tail -12 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
timeout 600 python tools/r06/kernel_ab.py --rounds 2 --variant default= --variant k_prep7s=COVERM_PREP_KERNEL:7 > $OUT/ab_5k.log 2> $OUT/ab_5k.err; echo "ab exit $?" >> $OUT/ab_5k.log
tail -5 $OUT/ab_5k.log; tail -5 $OUT/ab_5k.err
PROF_SKIP_PMC=1 bash tools/r06/prof_ab.sh wglist --variant default=
timeout 1700 python tools/r06/contig_sweep.py --out $R/gpurun_out/r06_contig_sweep.json > $OUT/sweep.log 2> $OUT/sweep.err; tail -12 $OUT/sweep.log; tail -5 $OUT/sweep.err
