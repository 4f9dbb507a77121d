"""Pair-mode end to end: paired realistic-entropy BAM on tmpfs -> coverm-amd with a pair filter, device ingest + device pair filter
against (a) the same file with a single-read filter (within 1.3x) and (b) the whole-file host path
(COVERM_PAIR_ON_HOST=1), whose table must be identical."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coverm_amd import bam as cbam, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
p = os.path.join(d, "pair_probe.bam")
ref = synth.make_reference(5000, 1_000_000_000, seed=1)
b = synth.make_reads(ref, reads, seed=3)
t = time.time()
cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=3, threads=threads)
print("write %.1fs %.2f GB" % (time.time() - t, os.path.getsize(p) / 1e9), flush=True)
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "coverm_amd", "coverm-amd")
base = [BIN, "contig", "-b", p, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "count", "-t", str(threads)]
runs = [("single-read filter, device ingest", ["--min-read-percent-identity", "95", "--proper-pairs-only"], {}, "single.tsv"),
        ("pair filter, device ingest + device join", ["--min-read-percent-identity-pair", "95", "--min-read-aligned-length-pair", "200", "--proper-pairs-only"], {}, "pair_dev.tsv"),
        ("pair filter, device ingest + device join", ["--min-read-percent-identity-pair", "95", "--min-read-aligned-length-pair", "200", "--proper-pairs-only"], {}, "pair_dev.tsv"),
        ("pair filter, whole file on the host", ["--min-read-percent-identity-pair", "95", "--min-read-aligned-length-pair", "200", "--proper-pairs-only"], {"COVERM_PAIR_ON_HOST": "1"}, "pair_host.tsv")]
for name, flags, env, out in runs:
    time.sleep(2)
    t = time.time()
    r = subprocess.run(base + flags + ["-o", os.path.join(d, out)], capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1", **env))
    dt = time.time() - t
    print("%s: wall %.3fs = %.1f M records/s (rc %d)" % (name, dt, reads / dt / 1e6, r.returncode), flush=True)
    for l in r.stderr.splitlines():
        if "pair filter" in l or "VmHWM" in l or "main:" in l or "reads mapped" in l or "ERROR" in l or "device ingest:" in l:
            print("    " + l)
same = open(os.path.join(d, "pair_dev.tsv")).read() == open(os.path.join(d, "pair_host.tsv")).read()
print("device and host pair-mode tables identical:", same)
for f in ("single.tsv", "pair_dev.tsv", "pair_host.tsv"):
    os.remove(os.path.join(d, f))
os.remove(p)
sys.exit(0 if same else 3)
