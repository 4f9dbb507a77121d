"""k_inflate (lane per block) against k_inflate_wave (wave per block, COVERM_INFLATE_V=3) through the binary: same file, tables must be identical."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coverm_amd import bam as cbam, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
threads = 16
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
p = os.path.join(d, "wave_probe.bam")
t = time.time()
ref = synth.make_reference(2000, 400_000_000, seed=1)
b = synth.make_reads(ref, reads, seed=3)
cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=threads)
print("synth + write %.1fs %.2f GB" % (time.time() - t, os.path.getsize(p) / 1e9), flush=True)
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "coverm_amd", "coverm-amd")
base = [BIN, "contig", "-b", p, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "count", "-t", str(threads)]
outs = {}
for name, env in (("v1", {}), ("v3", {"COVERM_INFLATE_V": "3"}), ("v1", {}), ("v3", {"COVERM_INFLATE_V": "3"})):
    t = time.time()
    o = os.path.join(d, name + ".tsv")
    try:
        r = subprocess.run(base + ["-o", o], capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1", **env), timeout=20)
    except subprocess.TimeoutExpired:
        print(name, "TIMEOUT", flush=True)
        continue
    dt = time.time() - t
    print("%s: wall %.3fs (rc %d)" % (name, dt, r.returncode), flush=True)
    for l in r.stderr.splitlines():
        if "device ingest" in l or "ERROR" in l or "fallback" in l.lower() or "[covermhip]" in l:
            print("    " + l)
    if r.returncode == 0:
        outs[name] = open(o).read()
print("tables identical:", outs.get("v1") is not None and outs.get("v1") == outs.get("v3"), flush=True)
os.remove(p)
