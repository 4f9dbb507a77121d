#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_call7; mkdir -p $OUT
cd $R
python tools/make_bam.py /dev/shm/p.bam 200000000 16 > $OUT/make.log 2>&1
timeout 120 tools/ubench/devwrite_probe /dev/shm/p.bam 14 > $OUT/devwrite.log 2>&1; cat $OUT/devwrite.log
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/p.bam -m mean trimmed_mean covered_fraction covered_bases variance length count reads_per_base rpkm tpm anir --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/p.tsv"
python - > $OUT/e2e.log 2>&1 <<PY
import subprocess, time, os
for label, env in (("default", {}), ("no numa bind", {"COVERM_NUMA_BIND": "0"}), ("default again", {})):
    rows = []
    for rep in range(5):
        time.sleep(4)
        t = time.time(); r = subprocess.run("$CMD".split(), capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1", **env)); dt = time.time() - t
        st = [float(l.split()[-1]) for l in r.stderr.splitlines() if "wall clock at" in l]
        ing = [l.split("device ingest: ")[1] for l in r.stderr.splitlines() if "device ingest: buffers" in l]
        mn = [l.split("main: ")[1] for l in r.stderr.splitlines() if "main:" in l]
        rows.append((dt, "spawn %.3f main %.3f exit %.3f | %s | %s" % (st[0] - t, st[1] - st[0], t + dt - st[1], ing[0] if ing else "?", mn[0][:70] if mn else "")))
    rows.sort()
    print("%-14s walls %s" % (label, " ".join("%.3f" % x[0] for x in rows)))
    for x in rows: print("      %.3f  %s" % x)
PY
cat $OUT/e2e.log
rm -f /dev/shm/p.bam /dev/shm/p.tsv
timeout 300 python tools/host_overhead.py 50000000 > $OUT/host_overhead.log 2>&1; tail -7 $OUT/host_overhead.log
