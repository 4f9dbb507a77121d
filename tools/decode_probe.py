"""Does an initialised HIP runtime slow the host decoder down?  (investigation aid, not part of the product)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coverm_amd import bam as cbam, synth  # noqa: E402
from coverm_amd.engine import FilterConfig, Session  # noqa: E402

ref = synth.make_reference(1000, 200_000_000, seed=1)
batch = synth.make_reads(ref, 10_000_000, seed=2)
p = "/tmp/probe.bam"
cbam.write_bam(p, ref.names, ref.lengths, batch, with_seq=True, threads=64)
L = cbam._lib()
L.covh_bam_set_buffer_cache.argtypes = [C.c_int]
L.covh_bam_set_pinned.argtypes = [C.c_int]
err = C.create_string_buffer(512)
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def rounds(tag, n=3):
    for _ in range(n):
        t = time.time()
        h = L.covh_bam_open(p.encode(), thr, 0, err, 512)
        t1 = time.time()
        L.covh_bam_close(h)
        print("%s: open %.3f close %.3f" % (tag, t1 - t, time.time() - t1), flush=True)


rounds("before HIP init")
s = Session(0, FilterConfig(), 0, want_hist=False)
s.set_targets(ref.lengths)
rounds("after HIP init")
L.covh_bam_set_buffer_cache(1)
rounds("after HIP init, buffer cache")
L.covh_bam_set_pinned(1)
rounds("after HIP init, buffer cache + pinned records", 4)
