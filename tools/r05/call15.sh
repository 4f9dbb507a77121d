#!/bin/bash
# round 5, last A/B: k_prep with tid / cigar_off one pass ahead in registers (k_prep5p, the default after this call's first run) against
# k_prep6 and k_prep — outputs byte for byte over four compile-time shapes at BASELINE config 2, kernel times alternating — then the parity
# files on the default
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call15; mkdir -p $OUT; rm -rf $OUT/*; cd $R
timeout 200 python tools/r05/prep_prefetch_ab.py > $OUT/ab.log 2> $OUT/ab.err; echo "ab exit $?" >> $OUT/ab.log
cat $OUT/ab.log; tail -5 $OUT/ab.err
