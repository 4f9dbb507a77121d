#!/bin/bash
# where the host waits inside cov_ingest_feed at 100 M reads (0.9 s in every run of call 5), the sliding-window LZ kernel, hardware queues
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call6; mkdir -p $OUT; rm -f $OUT/*
cd $R
timeout 150 python tools/make_bam.py /dev/shm/e2e.bam 100000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/e2e.bam -m mean trimmed_mean covered_fraction variance count --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/e2e.tsv"
run() {  # name, env...
  name=$1; shift
  for rep in 1 2 3; do
    rm -f /dev/shm/e2e.tsv
    s=$(date +%s%N); env "$@" COVERM_CLI_TIMING=1 timeout 30 $CMD 2> /tmp/err.log; e=$(date +%s%N)
    echo "$name rep $rep: wall $(( (e - s) / 1000000 )) ms | $(md5sum /dev/shm/e2e.tsv | cut -c1-8)" >> $OUT/sweep.log
    grep -h "covermhip\] ingest\|device ingest: buffers\|main:" /tmp/err.log | sed 's/^/      /' >> $OUT/sweep.log
  done
}
run "default" X=0
run "lz slide" COVERM_LZ_SLIDE=1
run "queues 8" GPU_MAX_HW_QUEUES=8
run "queues 8, lz slide" GPU_MAX_HW_QUEUES=8 COVERM_LZ_SLIDE=1
( COVERM_LZ_SLIDE=1 timeout 200 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --timeout 60 -k "default" 2>&1 | tail -4 ) > $OUT/pytest_slide.log 2>&1
timeout 100 tools/r03/wave_variants.sh r04_call6 20000000 "v3:X=0 v3_slide:COVERM_LZ_SLIDE=1" > /dev/null 2>&1
cat $OUT/sweep.log $OUT/pytest_slide.log $OUT/variants.log
rm -f /dev/shm/e2e.bam /dev/shm/e2e.tsv
