#!/bin/bash
# round 6, call 2: counters of k_prep_lean (and of the unchanged pileup beside it) at BASELINE config 2; the many-contig regime, first look
R=$GRAFT_REPO_ROOT; cd $R
bash tools/r06/prof_ab.sh lean --variant lean=
OUT=$R/gpurun_out/r06_call2; mkdir -p $OUT; rm -rf $OUT/*
for spec in "200000 1000000000" "2000000 2000000000"; do
  set -- $spec
  timeout 300 python tools/r06/kernel_ab.py --shapes 0 --rounds 2 --steps 5 --contigs $1 --bp $2 --min-len 1000 --variant k_prep7s=COVERM_PREP_KERNEL:7 --variant lean= > $OUT/contigs_$1.log 2> $OUT/contigs_$1.err; echo "exit $?" >> $OUT/contigs_$1.log
  tail -6 $OUT/contigs_$1.log; tail -3 $OUT/contigs_$1.err
done
