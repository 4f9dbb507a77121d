"""Every timing line of `coverm-amd contig` over a 2 000 000-contig sample (50 M reads): where the run's 0.8 s go.   python tools/r06/two_million_timing.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coverm_amd import bam as cbam, synth  # noqa: E402

ref = synth.make_reference(2_000_000, 2_000_000_000, seed=1, min_len=1000)
b = synth.make_reads(ref, 50_000_000, seed=2)
p = "/dev/shm/two_million.bam"
cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=0, threads=16)
del b
cmd = [os.path.join(ROOT, "coverm_amd", "coverm-amd"), "contig", "-b", p, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "-t", "16", "-o", "/dev/shm/two_million.tsv"]
for k in range(3):
    time.sleep(2)
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1"))
    print("== run %d: wall %.3f s" % (k, time.time() - t))
    if k == 2:
        print("\n".join(l for l in r.stderr.splitlines() if "mapping " not in l and "Rss" not in l))
os.remove(p)
