"""Round 6: eight feeders on the one-GPU lease box at config-5 size — where should their bytes come from?
`coverm-amd contig --devices 0,0,0,0,0,0,0,0` over ONE 200 M-read BAM cut into eight tid spans, with the mapped file registered up front (the
default for more than two feeders until this measurement) and with page-locked staging slots filled by threaded preads (COVERM_INGEST_IO=pread),
three runs each, alternating; one device sits behind all eight feeders, so this measures the HOST side (registration, reader threads under
the box's CPU quota), not eight GPUs.

    python tools/r06/eight_feeders_io.py [--reads N] > profiles/r06_eight_feeders_io.json
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coverm_amd import bam as cbam  # noqa: E402
from coverm_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200_000_000)
    ap.add_argument("--tmp", default="/dev/shm")
    a = ap.parse_args()
    ref = synth.make_reference(5000, 1_000_000_000, seed=1)
    t0 = time.time()
    batch = synth.make_reads(ref, a.reads, seed=3)
    path = os.path.join(a.tmp, "r06_feeders.bam")
    cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=2, level=1, threads=min(32, os.cpu_count() or 8))
    res = {"reads": int(batch.n_records), "bam_bytes": os.path.getsize(path), "generated_and_written_s": round(time.time() - t0, 1), "runs": []}
    del batch
    exe = os.path.join(ROOT, "coverm_amd", "coverm-amd")
    base = [exe, "contig", "-b", path, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "-t", "16", "-o", os.path.join(a.tmp, "r06_feeders.tsv")]
    tables = {}
    for rep in range(3):
        for name, devs, env in (("one device", "0", {}), ("eight feeders, mapped file registered up front", "0,0,0,0,0,0,0,0", {"COVERM_INGEST_IO": "mmap-upfront"}),
                                ("eight feeders, staging slots (pread)", "0,0,0,0,0,0,0,0", {"COVERM_INGEST_IO": "pread"}), ("eight feeders, default", "0,0,0,0,0,0,0,0", {})):
            time.sleep(4.0)
            e = dict(os.environ, COVERM_CLI_TIMING="1"); e.pop("COVERM_INGEST_IO", None); e.update(env)
            t1 = time.perf_counter()
            p = subprocess.run(base + ["--devices", devs], env=e, capture_output=True, text=True)
            wall = time.perf_counter() - t1
            txt = open(os.path.join(a.tmp, "r06_feeders.tsv")).read() if p.returncode == 0 else ""
            tables.setdefault("first", txt)
            spans = re.findall(r"span (\d+)/\d+: device ingest: first upload after ([0-9.]+)s, file read ([0-9.]+)s, staging waits ([0-9.]+)s, header walk [0-9.]+s, feed calls ([0-9.]+)s, inflate tail \+ parse ([0-9.]+)s, total ([0-9.]+)s, \d+ records, bytes from ([a-z ()]+)", p.stderr)
            res["runs"].append({"mode": name, "rep": rep, "wall_s": round(wall, 3), "rc": p.returncode, "table_equals_first_run": txt == tables["first"],
                                "spans": [{"span": int(s[0]), "buffers_s": float(s[1]), "file_read_s": float(s[2]), "staging_waits_s": float(s[3]), "feed_calls_s": float(s[4]),
                                           "tail_s": float(s[5]), "total_s": float(s[6]), "bytes_from": s[7]} for s in spans]})
            if p.returncode != 0:
                res["runs"][-1]["stderr_tail"] = p.stderr[-600:]
    os.remove(path)
    by = {}
    for r in res["runs"]:
        by.setdefault(r["mode"], []).append(r["wall_s"])
    res["median_wall_s"] = {k: sorted(v)[len(v) // 2] for k, v in by.items()}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
