#!/bin/bash
# round 5, last A/B: k_prep with tid / cigar_off one pass ahead in registers (k_prep5p, the default after this call's first run) against
# k_prep6 and k_prep — outputs byte for byte over four compile-time shapes at BASELINE config 2, kernel times alternating — then the parity
# files on the default
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call15; mkdir -p $OUT; rm -rf $OUT/*; cd $R
timeout 200 python tools/r05/prep_prefetch_ab.py > $OUT/ab.log 2> $OUT/ab.err; echo "ab exit $?" >> $OUT/ab.log
cat $OUT/ab.log; tail -5 $OUT/ab.err
( timeout 150 python -m pytest tests/test_gpu_abi_parity.py tests/test_gpu_configs.py tests/test_gpu_bounded_store.py tests/test_gpu_estimates.py tests/test_cli_fuzz.py tests/test_genes.py -x -q -m gpu --timeout 100 2>&1 | tail -4 ) > $OUT/pytest_default.log 2>&1
cat $OUT/pytest_default.log
