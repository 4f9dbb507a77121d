#!/bin/bash
# short first windows: ingest tests (many window sizes), end to end at 200 M reads
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call10; mkdir -p $OUT; rm -f $OUT/*
cd $R
( timeout 250 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pair_filter.py -m gpu -x -q --timeout 90 2>&1 | tail -6 ) > $OUT/pytest.log 2>&1
timeout 240 python tools/make_bam.py /dev/shm/e2e.bam 200000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/e2e.bam -m mean trimmed_mean covered_fraction variance count --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/e2e.tsv"
run() {
  name=$1; shift
  for rep in 1 2 3 4; do
    rm -f /dev/shm/e2e.tsv
    s=$(date +%s%N); env "$@" COVERM_CLI_TIMING=1 timeout 30 $CMD 2> /tmp/err.log; e=$(date +%s%N)
    echo "$name rep $rep: wall $(( (e - s) / 1000000 )) ms | $(md5sum /dev/shm/e2e.tsv | cut -c1-8)" >> $OUT/sweep.log
    grep -h "covermhip\] ingest\|device ingest: buffers\|main:" /tmp/err.log | sed 's/^/      /' >> $OUT/sweep.log
  done
}
run "ramp" X=0
run "ramp, queues 8" GPU_MAX_HW_QUEUES=8
cat $OUT/pytest.log $OUT/sweep.log
rm -f /dev/shm/e2e.bam /dev/shm/e2e.tsv
