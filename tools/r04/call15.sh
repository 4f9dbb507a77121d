#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call15; mkdir -p $OUT; rm -f $OUT/*
cd $R
timeout 120 python tools/make_bam.py /dev/shm/e2e.bam 100000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/e2e.bam -m mean trimmed_mean covered_fraction covered_bases variance length count reads_per_base anir rpkm tpm --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/e2e.tsv"
COVERM_CLI_TIMING=1 timeout 60 $CMD > $OUT/run1.out 2> $OUT/run1.err; echo "rc $?" >> $OUT/run1.err
CMD2="$R/coverm_amd/coverm-amd contig -b /dev/shm/e2e.bam -m mean variance count --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/e2e2.tsv"
COVERM_CLI_TIMING=1 timeout 60 $CMD2 > $OUT/run2.out 2> $OUT/run2.err; echo "rc $?" >> $OUT/run2.err
tail -30 $OUT/run1.err; echo ------; tail -12 $OUT/run2.err
rm -f /dev/shm/e2e.bam /dev/shm/e2e*.tsv
