#!/bin/bash
# kernel times of k_inflate_wave against k_inflate, with ablations: tools/r03/wave_variants.sh <tag> <reads> "<name>:<ENV=V,ENV=V> ..."
R=$GRAFT_REPO_ROOT; TAG=$1; READS=${2:-20000000}; VARS=$3
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 60 python tools/make_bam.py /dev/shm/ikt.bam $READS 16 > $OUT/make.log 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/ikt.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/ikt.tsv"
for v in $VARS; do
  name=${v%%:*}; envs=$(echo "${v#*:}" | tr ',' ' ')
  rm -rf /tmp/ikt_prof /dev/shm/ikt.tsv
  env $envs COVERM_NO_FAST_EXIT=1 COVERM_CLI_TIMING=1 timeout 25 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ikt_prof -- $CMD > /tmp/ikt.log 2>&1
  f=$(find /tmp/ikt_prof -name "*kernel_stats.csv" | head -1)
  echo "== $name [$envs] | $(grep -h 'windows of' /tmp/ikt.log | sed 's/.*ingest: //') | $(grep -h 'device ingest: buffers' /tmp/ikt.log | sed 's/.*inflate tail/tail/') | tsv $(md5sum /dev/shm/ikt.tsv 2>/dev/null | cut -c1-10)" >> $OUT/variants.log
  [ -n "$f" ] && python - "$f" >> $OUT/variants.log <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_inflate" in n or "k_lz_resolve" in n or "k_crc32" in n or "k_bam" in n:
        print("   %-34s calls %s  avg %.2f ms  min %.2f max %.2f  total %.1f ms" % (n.split("(")[0].replace("void ", "")[:34], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
done
cat $OUT/variants.log
rm -f /dev/shm/ikt.bam /dev/shm/ikt.tsv
