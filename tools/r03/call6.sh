#!/bin/bash
# round 3, GPU call 6: reader chunk sweep at 200 M reads, NUMA placement, new tests
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_call6; mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_configs.py tests/test_cli_binary.py -m gpu -x -q 2>&1 | tail -4 ) > $OUT/pytest.log 2>&1
cat $OUT/pytest.log
for n in /sys/devices/system/node/node*; do echo "$n cpus $(cat $n/cpulist) memfree $(grep MemFree $n/meminfo | awk '{print $4}')"; done > $OUT/numa.txt
for c in /sys/class/drm/card*/device; do echo "$c numa_node $(cat $c/numa_node 2>/dev/null) $(cat $c/vendor 2>/dev/null)"; done >> $OUT/numa.txt
cat /proc/self/status | grep -i "cpus_allowed_list\|mems_allowed_list" >> $OUT/numa.txt
cat $OUT/numa.txt
python tools/make_bam.py /dev/shm/p.bam 200000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/p.bam -m mean trimmed_mean covered_fraction variance -o /dev/shm/p.tsv"
run() { # label, threads, prefix-cmd (quoted, may be empty), env...
  python - "$@" >> $OUT/sweep.log 2>&1 <<PY
import subprocess, time, os, sys
label, thr, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
env = dict(os.environ, COVERM_CLI_TIMING="1")
for kv in sys.argv[4:]:
    k, v = kv.split("="); env[k] = v
cmd = prefix.split() + "$CMD".split() + ["-t", thr]
rows = []
for rep in range(3):
    time.sleep(3)
    t = time.time(); r = subprocess.run(cmd, capture_output=True, text=True, env=env); dt = time.time() - t
    ing = [l for l in r.stderr.splitlines() if "device ingest: buffers" in l]
    al = [l for l in r.stderr.splitlines() if "device allocations" in l]
    rows.append((dt, (ing[0].split("device ingest: ")[1] if ing else r.stderr[-300:]) + " | " + (al[0].split("windows of ")[1] if al else "")))
rows.sort()
print("%-40s walls %s | median: %s" % (label, " ".join("%.3f" % x[0] for x in rows), rows[1][1]))
PY
}
run "default (chunk 512 KiB, piece 32 MiB)" 16 ""
run "chunk 256 KiB" 16 "" COVERM_INGEST_CHUNK_KB=256
run "chunk 1 MiB" 16 "" COVERM_INGEST_CHUNK_KB=1024
run "chunk 512 KiB, piece 64 MiB" 16 "" COVERM_INGEST_PIECE_KB=65536
run "chunk 1 MiB, piece 128 MiB" 16 "" COVERM_INGEST_CHUNK_KB=1024 COVERM_INGEST_PIECE_KB=131072
NODES=$(ls -d /sys/devices/system/node/node* | wc -l)
if [ "$NODES" -gt 1 ]; then
  for n in $(seq 0 $((NODES-1))); do
    run "taskset node $n" 16 "taskset -c $(cat /sys/devices/system/node/node$n/cpulist)"
  done
fi
run "threads 10" 10 ""
run "threads 24" 24 ""
cat $OUT/sweep.log
rm -f /dev/shm/p.bam /dev/shm/p.tsv
