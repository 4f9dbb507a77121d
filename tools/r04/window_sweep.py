"""End to end at config-5 size with the wave-per-block inflate and smaller windows: with k_inflate a round costs a block's serial latency
whatever its size, so a window was one round of resident lanes (81 920 blocks, 1.8 GB of compressed bytes before the first kernel can start,
a 60 ms tail behind the last byte); k_inflate_wave's rounds cost what they hold.   python tools/r04/window_sweep.py [reads] [threads]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coverm_amd import bam as cbam, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
p = os.path.join(d, "window_sweep.bam")
ref = synth.make_reference(5000, 1_000_000_000, seed=1)
b = synth.make_reads(ref, reads, seed=3)
t = time.time()
cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=threads)
print("write %.1fs %.2f GB" % (time.time() - t, os.path.getsize(p) / 1e9), flush=True)
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "coverm_amd", "coverm-amd")
cmd = [BIN, "contig", "-b", p, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "count", "--min-read-percent-identity", "95", "--min-read-aligned-length", "50",
       "--proper-pairs-only", "-t", str(threads)]
configs = [("k_inflate (lane per block), 81920-block windows", {"COVERM_INFLATE_V": "1"})]
for rb in (81920, 40960, 20480, 10240, 5120):
    configs.append(("k_inflate_wave, %d-block windows" % rb, {"COVERM_INGEST_ROUND_BLOCKS": str(rb)}))
if os.environ.get("SWEEP_ONLY_LDS"):
    configs = [("k_inflate_wave, 81920-block windows", {}), ("k_inflate_lds, 81920-block windows", {"COVERM_INFLATE_V": "4"}),
               ("k_inflate_lds, 40960-block windows", {"COVERM_INFLATE_V": "4", "COVERM_INGEST_ROUND_BLOCKS": "40960"}),
               ("k_inflate_lds, 20480-block windows", {"COVERM_INFLATE_V": "4", "COVERM_INGEST_ROUND_BLOCKS": "20480"})]
tables = {}
for name, env in configs:
    walls = []
    for rep in range(3):
        time.sleep(1.0)
        out = os.path.join(d, "sweep.tsv")
        t = time.time()
        try:
            r = subprocess.run(cmd + ["-o", out], capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1", **env), timeout=60)
        except subprocess.TimeoutExpired:
            print("%s: TIMEOUT" % name, flush=True)
            break
        walls.append(time.time() - t)
        if rep == 0:
            tables[name] = open(out).read() if r.returncode == 0 else None
            for l in r.stderr.splitlines():
                if "device ingest" in l or "windows of" in l or "VmHWM" in l or "fallback" in l.lower() or "[cli]" in l or "exit" in l:
                    print("    " + l.strip())
    if walls:
        print("%s: wall %s s, median %.3f s (rc %d)" % (name, " ".join("%.3f" % w for w in walls), sorted(walls)[len(walls) // 2], r.returncode), flush=True)
first = tables.get(configs[0][0])
print("all tables identical:", first is not None and all(v == first for v in tables.values()), flush=True)
os.remove(p)
