"""Writes the realistic-entropy synthetic BAM used by the end-to-end probes: tools/make_bam.py <path> <reads> [threads]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coverm_amd import bam as cbam, synth  # noqa: E402

path, reads = sys.argv[1], int(sys.argv[2])
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ref = synth.make_reference(5000, 1_000_000_000, seed=1)
b = synth.make_reads(ref, reads, seed=3)
t = time.time()
cbam.write_bam(path, ref.names, ref.lengths, b, with_seq=2, threads=threads)
print("write %.1fs %.2f GB" % (time.time() - t, os.path.getsize(path) / 1e9), flush=True)
