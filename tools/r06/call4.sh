#!/bin/bash
# round 6, call 4: the whole -m gpu suite on the new defaults (k_prep_lean + k_prep_generic, one-table k_pileup_fast, parallel histogram layout),
# then their counters at BASELINE config 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call4; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 1500 python -m pytest tests -x -q -m gpu --timeout 600 2>&1 | tail -15 ) > $OUT/pytest_gpu.log 2>&1; cat $OUT/pytest_gpu.log
bash tools/r06/prof_ab.sh defaults --variant default=
