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
if [ -n "$PROF_PMC" ]; then
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_IFETCH"; do
  d=$OUT/pmc_$(echo $set | cut -d" " -f1)
  COVERM_NO_FAST_EXIT=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $d -- $CMD > $d.log 2>&1
done
find $OUT -name "*counter_collection.csv" -size +16M -exec sh -c 'head -4000 "$1" > "$1.head"; rm "$1"' _ {} \;
fi
rm -f /dev/shm/prof_ingest.bam /dev/shm/prof_ingest.tsv
find $OUT -name "*kernel_trace.csv" -size +8M -delete
grep -h "ingest\|main:" $OUT/plain_run.log $OUT/trace.log | head
