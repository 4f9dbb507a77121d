#!/bin/bash
# round 6, call 1: k_prep_lean + k_prep_generic against k_prep7s — parity tests first, then bytes + times alternating at BASELINE config 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call1; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests/test_gpu_abi_parity.py -x -q -m gpu --timeout 300 2>&1 | tail -15 ) > $OUT/pytest_parity.log 2>&1; cat $OUT/pytest_parity.log
timeout 400 python tools/r06/kernel_ab.py --variant k_prep7s=COVERM_PREP_KERNEL:7 --variant lean= > $OUT/ab.log 2> $OUT/ab.err; echo "ab exit $?" >> $OUT/ab.log
cat $OUT/ab.log; tail -5 $OUT/ab.err
