#!/bin/bash
# Kernel trace of one coverm-amd run (device ingest) over a synthetic BAM on tmpfs: tools/prof_ingest.sh <tag> <reads>
# -> gpurun_out/prof_<tag>/ (kernel_stats.csv + the launch timeline kernel_trace.csv, which is small for this command)
R=$GRAFT_REPO_ROOT
TAG=$1; READS=${2:-50000000}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python $R/tools/make_bam.py /dev/shm/prof_ingest.bam $READS 16 > $OUT/make.log 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/prof_ingest.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/prof_ingest.tsv"
COVERM_CLI_TIMING=1 $CMD 2> $OUT/plain_run.log
COVERM_NO_FAST_EXIT=1 COVERM_CLI_TIMING=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rm -f /dev/shm/prof_ingest.bam /dev/shm/prof_ingest.tsv
find $OUT -name "*kernel_trace.csv" -size +8M -delete
grep -h "ingest\|main:" $OUT/plain_run.log $OUT/trace.log | head
