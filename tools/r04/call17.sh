#!/bin/bash
# the split extraction measured on one box: COVERM_EXT_PARTS=1 (a lane per segment) and the default (a lane per quarter), alternating
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call17; mkdir -p $OUT; rm -f $OUT/*
cd $R
timeout 240 python tools/make_bam.py /dev/shm/e2e.bam 200000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/e2e.bam -m mean trimmed_mean covered_fraction covered_bases variance length count reads_per_base anir rpkm tpm --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/e2e.tsv"
for rep in 1 2 3 4 5; do
 for parts in 1 4; do
  s=$(date +%s%N); COVERM_EXT_PARTS=$parts COVERM_CLI_TIMING=1 timeout 30 $CMD 2> /tmp/err.log; e=$(date +%s%N)
  echo "parts $parts run $rep: wall $(( (e - s) / 1000000 )) ms | $(grep -h 'device ingest: buffers' /tmp/err.log | sed 's/.*file read/file read/' | cut -c1-150) | $(md5sum /dev/shm/e2e.tsv | cut -c1-8)" >> $OUT/e2e_runs.log
  sleep 1
 done
done
( timeout 200 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pair_filter.py tests/test_genes.py -m gpu -x -q --timeout 120 2>&1 | tail -4 ) > $OUT/pytest.log 2>&1
cat $OUT/e2e_runs.log $OUT/pytest.log
rm -f /dev/shm/e2e.bam /dev/shm/e2e.tsv
