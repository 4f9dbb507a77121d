#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_call14; mkdir -p $OUT
cd $R
( time timeout 170 python -m pytest tests -m gpu -x -q --timeout 60 2>&1 | grep -E "passed|failed|rror" | tail -5 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
timeout 60 python tools/make_bam.py /dev/shm/p.bam 50000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/p.bam -m mean trimmed_mean covered_fraction covered_bases variance length count reads_per_base rpkm tpm anir --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/p.tsv"
timeout 40 python - > $OUT/e2e.log 2>&1 <<PY
import subprocess, time, os
rows = []
for rep in range(5):
    time.sleep(1.5)
    t = time.time(); r = subprocess.run("$CMD".split(), capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1"), timeout=15); dt = time.time() - t
    st = [float(l.split()[-1]) for l in r.stderr.splitlines() if "wall clock at" in l]
    ing = [l.split("device ingest: ")[1] for l in r.stderr.splitlines() if "device ingest: buffers" in l]
    rows.append((dt, "main %.3f exit %.3f | %s" % (st[1] - st[0], t + dt - st[1], ing[0] if ing else r.stderr[-200:])))
rows.sort()
print("walls", " ".join("%.3f" % x[0] for x in rows))
for x in rows: print("   %.3f  %s" % x)
PY
cat $OUT/e2e.log
rm -f /dev/shm/p.bam /dev/shm/p.tsv
