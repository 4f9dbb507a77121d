"""`coverm-amd contig` over several BAM files on ONE GPU: one session (--devices 0) against two sessions on the device (--devices 0,0: two
samples in flight — one's start-up, tail and finish beside the other's stream).   python tools/r06/two_sessions_per_device.py [files] [reads] [out.json]"""
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coverm_amd import bam as cbam, synth  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
out = sys.argv[3] if len(sys.argv) > 3 else None
ref = synth.make_reference(5000, 1_000_000_000, seed=1)
b = synth.make_reads(ref, reads, seed=3)
paths = ["/dev/shm/two_sess_%d.bam" % i for i in range(n_files)]
cbam.write_bam(paths[0], ref.names, ref.lengths, b, with_seq=2, threads=16)
for p in paths[1:]:
    shutil.copyfile(paths[0], p)
del b
BIN = os.path.join(ROOT, "coverm_amd", "coverm-amd")
base = [BIN, "contig", "-b"] + paths + ["-m", "mean", "trimmed_mean", "covered_fraction", "variance", "-t", "16", "-o", "/dev/shm/two_sess.tsv"]
res = {"files": n_files, "reads_per_file": reads, "bam_bytes_each": os.path.getsize(paths[0]), "runs": []}
tables = set()
for k in range(7):
    mode = "warm-up" if k == 0 else ("--devices 0" if k % 2 else "--devices 0,0")
    devs = "0,0" if mode == "--devices 0,0" else "0"
    time.sleep(2)
    t = time.time()
    r = subprocess.run(base + ["--devices", devs], capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1"))
    dt = time.time() - t
    tables.add(open("/dev/shm/two_sess.tsv").read())
    hw = [l for l in r.stderr.splitlines() if "VmHWM" in l]
    row = {"mode": mode, "wall_s": round(dt, 3), "rc": r.returncode, "vm_hwm_mb": int(hw[0].split()[-2]) // 1024 if hw else None}
    print(row, flush=True)
    res["runs"].append(row)
res["tables_identical"] = len(tables) == 1
for mode in ("--devices 0", "--devices 0,0"):
    w = sorted(x["wall_s"] for x in res["runs"] if x["mode"] == mode)
    res[mode] = {"wall_s": w, "median_s": w[len(w) // 2]}
res["one_over_two"] = round(res["--devices 0"]["median_s"] / res["--devices 0,0"]["median_s"], 3)
print(json.dumps({k: v for k, v in res.items() if k != "runs"}, indent=1))
if out:
    json.dump(res, open(out, "w"), indent=1)
for p in paths:
    os.remove(p)
