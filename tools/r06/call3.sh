#!/bin/bash
# round 6, call 3: k_prep_generic with one set of atomics per run of a contig's records; the histogram layout over all CUs; k_pileup_fast with ONE
# table of biased deltas (d7 / d7b / d8) — parity tests, then bytes + times alternating at 5 000 / 200 000 / 2 000 000 contigs
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call3; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests/test_gpu_abi_parity.py tests/test_gpu_estimates.py tests/test_gpu_bounded_store.py tests/test_gpu_configs.py -x -q -m gpu --timeout 300 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
timeout 600 python tools/r06/kernel_ab.py --variant k_prep7s+fast7=COVERM_PREP_KERNEL:7 --variant lean+fast7= --variant lean+d7=COVERM_FAST_WAVES:71 --variant lean+d7b=COVERM_FAST_WAVES:72 --variant lean+d8=COVERM_FAST_WAVES:81 > $OUT/ab_5k.log 2> $OUT/ab_5k.err; echo "ab exit $?" >> $OUT/ab_5k.log
cat $OUT/ab_5k.log; tail -5 $OUT/ab_5k.err
for spec in "200000 1000000000" "2000000 2000000000"; do
  set -- $spec
  timeout 400 python tools/r06/kernel_ab.py --shapes 0 --rounds 2 --steps 5 --contigs $1 --bp $2 --min-len 1000 --variant k_prep7s+fast7=COVERM_PREP_KERNEL:7 --variant lean+fast7= --variant lean+d7=COVERM_FAST_WAVES:71 > $OUT/contigs_$1.log 2> $OUT/contigs_$1.err; echo "exit $?" >> $OUT/contigs_$1.log
  tail -7 $OUT/contigs_$1.log; tail -3 $OUT/contigs_$1.err
done
