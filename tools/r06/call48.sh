#!/bin/bash
# round 6, call 48: twenty runs of a 2 M-read file back to back, no pause — the launcher / child split against one process: total wall
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call48
python $R/tools/make_bam.py /dev/shm/s.bam 2000000 8 > /dev/null 2>&1
python - <<PY | tee $R/gpurun_out/r06_call48/back_to_back.log
import os, subprocess, time
cmd = ["$R/coverm_amd/coverm-amd", "contig", "-b", "/dev/shm/s.bam", "-m", "mean", "-t", "16", "-o", "/dev/shm/s.tsv"]
subprocess.run(cmd, capture_output=True); time.sleep(3)
for mode in ("default", "one process", "default", "one process"):
    env = dict(os.environ)
    if mode == "one process":
        env["COVERM_NO_FAST_EXIT"] = "1"
    t0 = time.time(); walls = []
    for i in range(20):
        t = time.time(); subprocess.run(cmd, capture_output=True, env=env); walls.append(round(time.time() - t, 3))
    print("%s: 20 runs back to back in %.2f s (each: min %.3f median %.3f max %.3f)" % (mode, time.time() - t0, min(walls), sorted(walls)[10], max(walls)), flush=True)
    time.sleep(3)
PY
rm -f /dev/shm/s.bam /dev/shm/s.tsv
