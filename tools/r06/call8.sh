#!/bin/bash
# round 6, call 8: step flags instead of the list (k_prep_lean at 200 000 contigs), k_prep_lean at seven waves without scalar spills against eight
# with; tests; the contig sweep (device-resident steps + the binary over BAM files with 5 000 / 200 000 / 2 000 000 references)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call8; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests/test_gpu_abi_parity.py tests/test_gpu_estimates.py tests/test_gpu_bounded_store.py -x -q -m gpu --timeout 300 2>&1 | tail -6 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
timeout 600 python tools/r06/kernel_ab.py --variant lean8= --variant lean7=COVERM_PREP_KERNEL:17 > $OUT/ab_5k.log 2> $OUT/ab_5k.err; echo "ab exit $?" >> $OUT/ab_5k.log
tail -12 $OUT/ab_5k.log; tail -5 $OUT/ab_5k.err
timeout 400 python tools/r06/kernel_ab.py --shapes 0 --rounds 2 --steps 5 --contigs 200000 --bp 1000000000 --min-len 1000 --variant lean8= --variant lean7=COVERM_PREP_KERNEL:17 --variant k_prep7s=COVERM_PREP_KERNEL:7 > $OUT/contigs_200000.log 2> $OUT/contigs_200000.err
tail -6 $OUT/contigs_200000.log
timeout 1500 python tools/r06/contig_sweep.py --out $R/gpurun_out/r06_contig_sweep.json > $OUT/sweep.log 2> $OUT/sweep.err; tail -30 $OUT/sweep.log; tail -5 $OUT/sweep.err
