"""End to end over one BAM on tmpfs, alternating the two ways of filling the staging slots (default: copied from a mapping with non-temporal
stores; COVERM_INGEST_IO=pread): wall times and the driver's timing lines.   python tools/r06/feed_ab.py [reads] [pairs] [out.json]"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coverm_amd import bam as cbam, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
out = sys.argv[3] if len(sys.argv) > 3 else None
d = "/dev/shm"
p = os.path.join(d, "feed_ab.bam")
ref = synth.make_reference(5000, 1_000_000_000, seed=1)
b = synth.make_reads(ref, reads, seed=3)
t = time.time()
cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=16)
size = os.path.getsize(p)
print("write %.1fs %.2f GB" % (time.time() - t, size / 1e9), flush=True)
del b
BIN = os.path.join(ROOT, "coverm_amd", "coverm-amd")
cmd = [BIN, "contig", "-b", p, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "-t", "16", "-o", os.path.join(d, "feed_ab.tsv")]
res = {"reads": reads, "bam_bytes": size, "runs": []}
tables = set()
MODES = [("mapping + non-temporal copy", {}), ("pread", {"COVERM_INGEST_IO": "pread"})]
if os.environ.get("FEED_AB_PREPARE"):      # the ingest's streams and events created from cov_create on (COV_WANT_INGEST) against inside cov_ingest_begin
    MODES = [("ingest prepared from cov_create on (default)", {}), ("ingest prepared in cov_ingest_begin", {"COVERM_KNOBS": "ingest_prepare=0"})]
elif os.environ.get("FEED_AB_ROUNDS"):      # blocks per inflate round = per window (COVERM_KNOBS ingest_round_blocks; default 81920)
    MODES = [("rounds of %d blocks" % n, {"COVERM_KNOBS": "ingest_round_blocks=%d" % n}) for n in (81920, 61440, 40960, 122880)]
elif os.environ.get("FEED_AB_COPY_STREAMS"):      # (needs the build of tools/r06/call33.sh: the knob left the library with the measurement)      # one upload stream against two that take the pieces in turn (COVERM_KNOBS ingest_copy_streams)
    MODES = [("one upload stream", {"COVERM_KNOBS": "ingest_copy_streams=1"}), ("two upload streams", {"COVERM_KNOBS": "ingest_copy_streams=2"})]
elif os.environ.get("FEED_AB_PIECES"):      # bytes per staging piece = per H2D copy (COVERM_KNOBS ingest_piece_kb; default 32 MiB)
    MODES = [("pieces of %d MiB" % mb, {"COVERM_KNOBS": "ingest_piece_kb=%d" % (mb << 10)}) for mb in (32, 64, 128, 16)]
elif os.environ.get("FEED_AB_EXIT"):      # the command with its teardown in a detached child (default) against one process (COVERM_NO_FAST_EXIT)
    MODES = [("teardown in the background (default)", {}), ("one process", {"COVERM_NO_FAST_EXIT": "1"})]
elif os.environ.get("FEED_AB_ZAPS"):      # how the mapping's pages leave the page table again (COVERM_KNOBS ingest_zap)
    MODES = [("mapping, zap the piece in front (default)", {}), ("mapping, zap per chunk", {"COVERM_KNOBS": "ingest_zap=1"}),
             ("mapping, zap at the end", {"COVERM_KNOBS": "ingest_zap=0"}), ("pread", {"COVERM_INGEST_IO": "pread"})]
for k in range(len(MODES) * pairs + 1):
    mode, extra = ("warm-up", {}) if k == 0 else MODES[(k - 1) % len(MODES)]
    env = dict(os.environ, COVERM_CLI_TIMING="1", **extra)
    time.sleep(2)
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    dt = time.time() - t
    tables.add(open(os.path.join(d, "feed_ab.tsv")).read())
    m = re.search(r"file read ([0-9.]+)s, staging waits ([0-9.]+)s.*inflate tail \+ parse ([0-9.]+)s, total ([0-9.]+)s.*bytes from (.*)", r.stderr)
    row = {"mode": mode, "wall_s": round(dt, 3), "rc": r.returncode}
    hw = re.search(r"VmHWM:\s+(\d+) kB", r.stderr)
    if hw:
        row["vm_hwm_mb"] = int(hw.group(1)) // 1024
    stamps = [float(l.split()[-1]) for l in r.stderr.splitlines() if "wall clock at" in l]
    if len(stamps) == 2:
        row.update(spawn_to_main_s=round(stamps[0] - t, 3), main_s=round(stamps[1] - stamps[0], 3), exit_to_reaped_s=round(t + dt - stamps[1], 3))
    mm = re.search(r"main: arguments \+ device sessions ([0-9.]+)s, samples ([0-9.]+)s", r.stderr)
    if mm:
        row.update(sessions_s=float(mm.group(1)), samples_s=float(mm.group(2)))
    hm = re.search(r"host time in drain ([0-9.]+)s \(waiting for a verification ([0-9.]+)s\), in launches ([0-9.]+)s \(drains inside included\), in upload calls ([0-9.]+)s", r.stderr)
    if hm:
        row.update(drain_s=float(hm.group(1)), launches_s=float(hm.group(3)), upload_calls_s=float(hm.group(4)))
    um = re.search(r"first upload after ([0-9.]+)s", r.stderr)
    if um:
        row["first_upload_after_s"] = float(um.group(1))
    fm = re.search(r"feed calls ([0-9.]+)s", r.stderr)
    if fm:
        row["feed_calls_s"] = float(fm.group(1))
    if m:
        row.update(file_read_s=float(m.group(1)), staging_waits_s=float(m.group(2)), tail_s=float(m.group(3)), ingest_s=float(m.group(4)), bytes_from=m.group(5))
    print(row, flush=True)
    res["runs"].append(row)
res["tables_identical"] = len(tables) == 1
for mode, _ in MODES:
    w = sorted(x["wall_s"] for x in res["runs"] if x["mode"] == mode)
    g = sorted(x.get("ingest_s", 0) for x in res["runs"] if x["mode"] == mode)
    res[mode] = {"wall_median_s": w[len(w) // 2], "wall_min_s": w[0], "ingest_median_s": g[len(g) // 2], "feed_GBps_median": round(size / g[len(g) // 2] / 1e9, 1)}
print(json.dumps({k: v for k, v in res.items() if k != "runs"}, indent=1))
if out:
    json.dump(res, open(out, "w"), indent=1)
os.remove(p)
