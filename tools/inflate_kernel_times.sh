#!/bin/bash
# HISTORICAL (round 3): the lane-per-block inflate's table-size and sort variants this sweeps (COVERM_INFLATE_BITS / _DIST_BITS / _SORT8) were
# deleted in round 4 with the measurements that decided them (profiles/r03_*); today every pass runs the default.
# Per-variant kernel times of the device ingest (rocprofv3 --kernel-trace --stats of one coverm-amd run each) over one synthetic BAM:
# tools/inflate_kernel_times.sh <reads> "<lit,dist,sort8> ..."
R=$GRAFT_REPO_ROOT
READS=${1:-50000000}; VARS=${2:-"7,6,0 7,6,1 6,5,1 5,5,1"}
python $R/tools/make_bam.py /dev/shm/ikt.bam $READS 16
cd /tmp && export TMPDIR=/tmp
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/ikt.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/ikt.tsv"
for v in $VARS; do
  IFS=, read lb db s8 <<< "$v"
  rm -rf /tmp/ikt_prof
  COVERM_INFLATE_BITS=$lb COVERM_INFLATE_DIST_BITS=$db COVERM_INFLATE_SORT8=$s8 COVERM_NO_FAST_EXIT=1 COVERM_CLI_TIMING=1 timeout 300 \
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ikt_prof -- $CMD > /tmp/ikt.log 2>&1
  f=$(find /tmp/ikt_prof -name "*kernel_stats.csv" | head -1)
  echo "== lit $lb dist $db sort8 $s8 | $(grep -h 'windows of' /tmp/ikt.log | sed 's/.*ingest: //') | $(grep -h 'device ingest: buffers' /tmp/ikt.log | sed 's/.*inflate tail/tail/') | tsv $(md5sum /dev/shm/ikt.tsv | cut -c1-10)"
  python - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_inflate" in n or "k_lz_resolve" in n or "k_crc32" in n:
        print("   %-28s calls %s  avg %.2f ms  total %.1f ms" % (n.split("(")[0].replace("void ", "")[:28], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -f /dev/shm/ikt.bam /dev/shm/ikt.tsv
