#!/bin/bash
# kernel times of the device ingest per k_inflate variant: tools/r03/inflate_variants.sh <tag> <reads> "<version,lit,dist,sort8> ..."
R=$GRAFT_REPO_ROOT; TAG=$1; READS=${2:-50000000}; VARS=$3
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
[ -n "$PYTEST" ] && ( COVERM_INFLATE_V=2 timeout 600 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q 2>&1 | tail -3 ) > $OUT/pytest.log 2>&1
python tools/make_bam.py /dev/shm/ikt.bam $READS 16 > $OUT/make.log 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/ikt.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/ikt.tsv"
for v in $VARS; do
  IFS=, read ver lb db s8 <<< "$v"
  rm -rf /tmp/ikt_prof
  COVERM_INFLATE_V=$ver COVERM_INFLATE_BITS=$lb COVERM_INFLATE_DIST_BITS=$db COVERM_INFLATE_SORT8=$s8 COVERM_NO_FAST_EXIT=1 COVERM_CLI_TIMING=1 timeout 300 \
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ikt_prof -- $CMD > /tmp/ikt.log 2>&1
  f=$(find /tmp/ikt_prof -name "*kernel_stats.csv" | head -1)
  echo "== v$ver lit $lb dist $db sort8 $s8 | $(grep -h 'windows of' /tmp/ikt.log | sed 's/.*ingest: //') | $(grep -h 'device ingest: buffers' /tmp/ikt.log | sed 's/.*inflate tail/tail/') | tsv $(md5sum /dev/shm/ikt.tsv | cut -c1-10)" >> $OUT/variants.log
  python - "$f" >> $OUT/variants.log <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_inflate" in n or "k_lz_resolve" in n:
        print("   %-34s calls %s  avg %.2f ms  total %.1f ms" % (n.split("(")[0].replace("void ", "")[:34], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
done
cat $OUT/pytest.log 2>/dev/null; cat $OUT/variants.log
rm -f /dev/shm/ikt.bam /dev/shm/ikt.tsv
