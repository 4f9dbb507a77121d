#!/bin/bash
# round 5, the last A/B (its fifth and final run): the four compilations of k_prep's body that ship (k_prep8s / 7s: one record per thread and
# pass, roots one pass ahead, eight / seven waves per SIMD; k_prep6; k_prep5p) and the default's choice by shape — outputs byte for byte over
# four compile-time shapes at BASELINE config 2, kernel times alternating — then the parity files (with the new test that forces each
# compilation on every shape) and the smoke run
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call15; mkdir -p $OUT; rm -rf $OUT/*; cd $R
timeout 120 python tools/r05/prep_prefetch_ab.py > $OUT/ab.log 2> $OUT/ab.err; echo "ab exit $?" >> $OUT/ab.log
cat $OUT/ab.log; tail -5 $OUT/ab.err
( timeout 170 python -m pytest tests/test_gpu_abi_parity.py tests/test_gpu_configs.py tests/test_gpu_bounded_store.py tests/test_gpu_estimates.py tests/test_cli_fuzz.py tests/test_genes.py tests/test_gpu_pair_filter.py -x -q -m gpu --timeout 100 2>&1 | tail -4 ) > $OUT/pytest_default.log 2>&1
cat $OUT/pytest_default.log
( timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $OUT/smoke.log 2>&1; cat $OUT/smoke.log
