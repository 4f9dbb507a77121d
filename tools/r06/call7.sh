#!/bin/bash
# round 6, call 7: per-kernel times (rocprofv3 kernel trace) at 200 000 and 2 000 000 contigs; counters of k_prep_generic at 2 M contigs; two samples in flight
R=$GRAFT_REPO_ROOT; cd $R
PROF_SKIP_PMC=1 bash tools/r06/prof_ab.sh c200k --contigs 200000 --bp 1000000000 --min-len 1000 --variant default=
bash tools/r06/prof_ab.sh c2M --contigs 2000000 --bp 2000000000 --min-len 1000 --variant default=
OUT=$R/gpurun_out/r06_call7; mkdir -p $OUT; rm -rf $OUT/*
timeout 300 python tools/r06/two_samples.py > $OUT/two_samples.json 2> $OUT/two_samples.err; tail -30 $OUT/two_samples.json; tail -3 $OUT/two_samples.err
