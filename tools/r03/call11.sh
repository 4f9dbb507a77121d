#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_call11; mkdir -p $OUT
cd $R
timeout 100 python tools/make_bam.py /dev/shm/p.bam 50000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/p.bam -m mean trimmed_mean covered_fraction covered_bases variance length count reads_per_base rpkm tpm anir --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/p.tsv"
timeout 150 python - > $OUT/tail_variants.log 2>&1 <<PY
import subprocess, time, os
for label, env in (("shared ext stream, slots kept", {}), ("own ext stream, slots kept", {"COVERM_INGEST_EXT_STREAM": "1"}),
                   ("shared ext stream, slots released early", {"COVERM_RELEASE_STAGING": "1"}), ("own ext stream, slots released early", {"COVERM_INGEST_EXT_STREAM": "1", "COVERM_RELEASE_STAGING": "1"}),
                   ("shared ext stream, slots kept (again)", {}), ("own ext stream, slots kept (again)", {"COVERM_INGEST_EXT_STREAM": "1"})):
    rows = []
    for rep in range(4):
        time.sleep(1.5)
        t = time.time(); r = subprocess.run("$CMD".split(), capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1", **env), timeout=20); dt = time.time() - t
        st = [float(l.split()[-1]) for l in r.stderr.splitlines() if "wall clock at" in l]
        ing = [l.split("device ingest: ")[1] for l in r.stderr.splitlines() if "device ingest: buffers" in l]
        rows.append((dt, st[1] - st[0], t + dt - st[1], ing[0] if ing else "?"))
    rows.sort()
    print("%-42s walls %s | median: main %.3f exit %.3f | %s" % (label, " ".join("%.3f" % x[0] for x in rows), rows[1][1], rows[1][2], rows[1][3]))
PY
cat $OUT/tail_variants.log
rm -f /dev/shm/p.bam /dev/shm/p.tsv
