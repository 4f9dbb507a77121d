#!/bin/bash
# HISTORICAL (round 3): COVERM_INFLATE_BITS / _DIST_BITS / _SORT8 no longer exist (the losing variants were deleted in round 4); today
# every pass runs the default.  The round-5 equivalents are tools/r05/call*.sh (COVERM_INFLATE_SINK, COVERM_LZ_V).
# coverm-amd wall time over one synthetic BAM on tmpfs for several k_inflate table sizes: tools/ingest_variants.sh <reads> "<lb,db> ..." [reps]
R=$GRAFT_REPO_ROOT
READS=${1:-50000000}; VARS=${2:-"8,6 7,6 7,5 6,5"}; REPS=${3:-2}
python $R/tools/make_bam.py /dev/shm/variants.bam $READS 16
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/variants.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/variants.tsv"
for v in $VARS; do
  IFS=, read lb db s8 <<< "$v"; s8=${s8:-0}
  for r in $(seq $REPS); do
    sleep ${PAUSE:-0}
    s=$(date +%s.%N)
    COVERM_INFLATE_BITS=$lb COVERM_INFLATE_DIST_BITS=$db COVERM_INFLATE_SORT8=$s8 COVERM_CLI_TIMING=1 $CMD 2> /tmp/variants.err
    e=$(date +%s.%N)
    echo "lit $lb dist $db sort8 $s8: wall $(python -c "print(round($e - $s, 3))") s | $(grep -h 'windows of' /tmp/variants.err | sed 's/.*ingest: //') | $(grep -h 'device ingest: buffers' /tmp/variants.err | sed 's/.*device ingest: //')"
    md5sum /dev/shm/variants.tsv | cut -c1-12
  done
done
rm -f /dev/shm/variants.bam /dev/shm/variants.tsv
