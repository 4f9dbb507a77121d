#!/bin/bash
# the GPU suite on the last commit (the Python harness moved to tests/), and the spread of the end-to-end time: ten warm runs per level
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call13; mkdir -p $OUT; rm -f $OUT/*
cd $R
( timeout 500 python -m pytest tests -m gpu -x -q --timeout 180 2>&1 | tail -6 ) > $OUT/pytest_gpu.log 2>&1
timeout 240 python tools/make_bam.py /dev/shm/e2e.bam 200000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/e2e.bam -m mean trimmed_mean covered_fraction covered_bases variance length count reads_per_base anir rpkm tpm --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/e2e.tsv"
for rep in 1 2 3 4 5 6 7 8 9 10 11; do
  s=$(date +%s%N); COVERM_CLI_TIMING=1 timeout 30 $CMD 2> /tmp/err.log; e=$(date +%s%N)
  echo "level 1 run $rep: wall $(( (e - s) / 1000000 )) ms | $(grep -h 'device ingest: buffers' /tmp/err.log | sed 's/.*file read/file read/' | cut -c1-150) | $(grep -h 'main:' /tmp/err.log | sed 's/.*main: //' | cut -c1-60)" >> $OUT/e2e_runs.log
  sleep 2
done
cat $OUT/pytest_gpu.log $OUT/e2e_runs.log
rm -f /dev/shm/e2e.bam /dev/shm/e2e.tsv
