#!/bin/bash
# round 3, GPU call 3: k_inflate2 correctness (ingest tests) + kernel times per variant; exit time against footprint
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_call3; mkdir -p $OUT
cd $R
( time timeout 900 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q ) > $OUT/pytest_v2.log 2>&1
tail -15 $OUT/pytest_v2.log
python tools/make_bam.py /dev/shm/ikt.bam 50000000 16 > $OUT/make.log 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/ikt.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/ikt.tsv"
for v in "1,7,6" "2,7,5" "2,7,6" "2,6,5" "2,8,5"; do
  IFS=, read ver lb db <<< "$v"
  rm -rf /tmp/ikt_prof
  COVERM_INFLATE_V=$ver COVERM_INFLATE_BITS=$lb COVERM_INFLATE_DIST_BITS=$db COVERM_NO_FAST_EXIT=1 COVERM_CLI_TIMING=1 timeout 300 \
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ikt_prof -- $CMD > /tmp/ikt.log 2>&1
  f=$(find /tmp/ikt_prof -name "*kernel_stats.csv" | head -1)
  echo "== v$ver lit $lb dist $db | $(grep -h 'windows of' /tmp/ikt.log | sed 's/.*ingest: //') | $(grep -h 'device ingest: buffers' /tmp/ikt.log | sed 's/.*inflate tail/tail/') | tsv $(md5sum /dev/shm/ikt.tsv | cut -c1-10)" >> $OUT/variants.log
  python - "$f" >> $OUT/variants.log <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_inflate" in n or "k_lz_resolve" in n or "k_crc32" in n or "k_bam" in n:
        print("   %-28s calls %s  avg %.2f ms  total %.1f ms" % (n.split("(")[0].replace("void ", "")[:28], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
done
cat $OUT/variants.log
# exit time against the footprint: smaller rounds = smaller buffers
cd $R
for rb in 81920 40960 20480; do
  echo "== round blocks $rb" >> $OUT/exit.log
  for i in 1 2; do sleep 2; COVERM_INGEST_ROUND_BLOCKS=$rb python - >> $OUT/exit.log 2>&1 <<PY
import subprocess, time, os
cmd = "$CMD".split()
t = time.time(); r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1")); dt = time.time() - t
st = [float(l.split()[-1]) for l in r.stderr.splitlines() if "wall clock at" in l]
print("wall %.3f  spawn->main %.3f  main %.3f  exit->reaped %.3f" % (dt, st[0] - t, st[1] - st[0], t + dt - st[1]))
PY
  done
done
cat $OUT/exit.log
rm -f /dev/shm/ikt.bam /dev/shm/ikt.tsv
