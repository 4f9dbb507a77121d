"""coverm-amd over one BAM with GPU_MAX_HW_QUEUES = 1 .. 8: every hardware queue of the runtime comes with a 173 MB context-save area in
host memory (tools/r06/call26.sh: seven of them are the process's largest mappings), created with the queue and taken apart when the
process ends.  Wall time, time in main(), exit -> reaped, session creation, ingest.   python tools/r06/queues_probe.py [reads] [out.json]"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coverm_amd import bam as cbam, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
out = sys.argv[2] if len(sys.argv) > 2 else None
p = "/dev/shm/queues_probe.bam"
ref = synth.make_reference(5000, 1_000_000_000, seed=1)
b = synth.make_reads(ref, reads, seed=3)
cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=16)
del b
BIN = os.path.join(ROOT, "coverm_amd", "coverm-amd")
cmd = [BIN, "contig", "-b", p, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "-t", "16", "-o", "/dev/shm/queues_probe.tsv"]
rows = []
tables = set()
for rep in range(3):
    for q in ("default", "1", "2", "3", "4", "8"):
        env = dict(os.environ, COVERM_CLI_TIMING="1")
        if q != "default":
            env["GPU_MAX_HW_QUEUES"] = q
        time.sleep(1.5)
        t = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        dt = time.time() - t
        tables.add(open("/dev/shm/queues_probe.tsv").read())
        st = [float(l.split()[-1]) for l in r.stderr.splitlines() if "wall clock at" in l]
        big = len(re.findall(r"mapping \d+: 17\d MB resident", r.stderr))
        m = re.search(r"inflate tail \+ parse ([0-9.]+)s, total ([0-9.]+)s", r.stderr)
        mm = re.search(r"device sessions ([0-9.]+)s", r.stderr)
        row = {"queues": q, "rep": rep, "wall_s": round(dt, 3), "rc": r.returncode, "main_s": round(st[1] - st[0], 3) if len(st) == 2 else None,
               "exit_to_reaped_s": round(t + dt - st[1], 3) if len(st) == 2 else None, "sessions_s": float(mm.group(1)) if mm else None,
               "ingest_s": float(m.group(2)) if m else None, "tail_s": float(m.group(1)) if m else None, "save_areas_among_top8": big}
        print(row, flush=True)
        rows.append(row)
res = {"reads": reads, "tables_identical": len(tables) == 1, "runs": rows}
for q in ("default", "1", "2", "3", "4", "8"):
    w = sorted(x["wall_s"] for x in rows if x["queues"] == q and x["rep"] > 0)
    g = sorted(x["ingest_s"] for x in rows if x["queues"] == q and x["rep"] > 0 and x["ingest_s"])
    res["GPU_MAX_HW_QUEUES=" + q] = {"wall_s": w, "ingest_s": g}
print(json.dumps({k: v for k, v in res.items() if k != "runs"}, indent=1))
if out:
    json.dump(res, open(out, "w"), indent=1)
os.remove(p)
