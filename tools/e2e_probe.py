"""End-to-end probe: realistic-entropy BAM on tmpfs -> coverm-amd (device ingest | CPU stream) wall times with the CLI's own timing lines."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coverm_amd import bam as cbam, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
p = os.path.join(d, "e2e_probe.bam")
ref = synth.make_reference(5000, 1_000_000_000, seed=1)
b = synth.make_reads(ref, reads, seed=3)
t = time.time()
cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=threads)
print("write %.1fs %.2f GB" % (time.time() - t, os.path.getsize(p) / 1e9), flush=True)
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "coverm_amd", "coverm-amd")
cmd = [BIN, "contig", "-b", p, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "-t", str(threads), "-o", os.path.join(d, "out.tsv")]
runs = [("device ingest", {})] * int(os.environ.get("PROBE_REPS", "2"))
if not os.environ.get("PROBE_NO_CPU"):
    runs.append(("cpu stream", {"COVERM_NO_GPU_INGEST": "1"}))
for name, env in runs:
    time.sleep(float(os.environ.get("PROBE_SLEEP", "0")))
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1", **env))
    dt = time.time() - t
    print("%s: wall %.3fs = %.1f M reads/s (rc %d)" % (name, dt, reads / dt / 1e6, r.returncode), flush=True)
    stamps = [float(l.split()[-1]) for l in r.stderr.splitlines() if "wall clock at" in l]
    if len(stamps) == 2:
        print("    spawn -> main() %.3fs, main() %.3fs, exit -> reaped %.3fs" % (stamps[0] - t, stamps[1] - stamps[0], t + dt - stamps[1]))
    for l in r.stderr.splitlines():
        if "ingest" in l or "VmHWM" in l or "stream read" in l or "main:" in l:
            print("    " + l)
os.remove(p)
