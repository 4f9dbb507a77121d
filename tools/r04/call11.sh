#!/bin/bash
# short first windows against full ones, same box, alternating
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call11; mkdir -p $OUT; rm -f $OUT/*
cd $R
timeout 240 python tools/make_bam.py /dev/shm/e2e.bam 200000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/e2e.bam -m mean trimmed_mean covered_fraction variance count --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/e2e.tsv"
timeout 30 $CMD 2> /dev/null      # warm
for rep in 1 2 3 4 5; do
  for ramp in 1 0; do
    rm -f /dev/shm/e2e.tsv
    s=$(date +%s%N); COVERM_INGEST_RAMP=$ramp COVERM_CLI_TIMING=1 timeout 30 $CMD 2> /tmp/err.log; e=$(date +%s%N)
    echo "ramp $ramp rep $rep: wall $(( (e - s) / 1000000 )) ms | $(grep -h 'device ingest: buffers' /tmp/err.log | sed 's/.*file read/file read/') | $(grep -h 'main:' /tmp/err.log | sed 's/.*main: //' | cut -c1-60)" >> $OUT/ab.log
  done
done
cat $OUT/ab.log
rm -f /dev/shm/e2e.bam /dev/shm/e2e.tsv
