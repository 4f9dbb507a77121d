#!/bin/bash
# round 6, call 32: is the link or the device the bound of the ingest now?  Kernel + memory-copy trace of coverm-amd over a 200 M-read BAM
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_call32; mkdir -p $OUT
python $R/tools/make_bam.py /dev/shm/t.bam 200000000 16 > $OUT/make.log 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/t.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/t.tsv"
COVERM_CLI_TIMING=1 $CMD 2> $OUT/plain1.log; sleep 3
COVERM_CLI_TIMING=1 $CMD 2> $OUT/plain2.log; sleep 3
COVERM_NO_FAST_EXIT=1 COVERM_CLI_TIMING=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rm -f /dev/shm/t.bam /dev/shm/t.tsv
grep -h "ingest\|main:" $OUT/plain2.log $OUT/trace.log | head -8
ls $OUT/trace/*/
