#!/bin/bash
# round 6, call 6: k_pileup_fast with 4 x deltas (the running value is the bin's LDS address), the clipped histogram loop for tiles at contig ends,
# DPP sums in the flush — tests with both table layouts, then times at 5 000 / 200 000 / 2 000 000 contigs
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call6; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests/test_gpu_abi_parity.py tests/test_gpu_estimates.py tests/test_gpu_configs.py tests/test_gpu_bounded_store.py tests/test_genes.py -x -q -m gpu --timeout 300 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
timeout 600 python tools/r06/kernel_ab.py --variant fast1= --variant fast2t=COVERM_FAST_TABLES:2 > $OUT/ab_5k.log 2> $OUT/ab_5k.err; echo "ab exit $?" >> $OUT/ab_5k.log
tail -14 $OUT/ab_5k.log; tail -5 $OUT/ab_5k.err
for spec in "200000 1000000000" "2000000 2000000000"; do
  set -- $spec
  timeout 400 python tools/r06/kernel_ab.py --shapes 0 --rounds 2 --steps 5 --contigs $1 --bp $2 --min-len 1000 --variant fast1= --variant fast2t=COVERM_FAST_TABLES:2 > $OUT/contigs_$1.log 2> $OUT/contigs_$1.err; echo "exit $?" >> $OUT/contigs_$1.log
  tail -6 $OUT/contigs_$1.log; tail -3 $OUT/contigs_$1.err
done
