#!/bin/bash
# k_bam_extract with a lane per quarter of a segment's chain: ingest / pair / genes tests, kernel times, end to end
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call16; mkdir -p $OUT; rm -f $OUT/*
cd $R
( timeout 300 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pair_filter.py tests/test_genes.py -m gpu -x -q --timeout 120 2>&1 | tail -6 ) > $OUT/pytest.log 2>&1
timeout 100 tools/r03/wave_variants.sh r04_call16 20000000 "split4:X=0" > /dev/null 2>&1
timeout 240 python tools/make_bam.py /dev/shm/e2e.bam 200000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/e2e.bam -m mean trimmed_mean covered_fraction covered_bases variance length count reads_per_base anir rpkm tpm --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/e2e.tsv"
for rep in 1 2 3 4 5 6; do
  s=$(date +%s%N); COVERM_CLI_TIMING=1 timeout 30 $CMD 2> /tmp/err.log; e=$(date +%s%N)
  echo "level 1 run $rep: wall $(( (e - s) / 1000000 )) ms | $(grep -h 'device ingest: buffers' /tmp/err.log | sed 's/.*file read/file read/' | cut -c1-150) | $(grep -h 'main:' /tmp/err.log | sed 's/.*main: //' | cut -c1-60) | $(md5sum /dev/shm/e2e.tsv | cut -c1-8)" >> $OUT/e2e_runs.log
  sleep 2
done
cat $OUT/pytest.log $OUT/variants.log $OUT/e2e_runs.log
rm -f /dev/shm/e2e.bam /dev/shm/e2e.tsv
