"""Host-side decode probe: streamed reader vs whole-file reader at several thread counts / window sizes (no GPU work)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coverm_amd import bam as cbam, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
ref = synth.make_reference(5000, 1_000_000_000, seed=1)
b = synth.make_reads(ref, reads, seed=3)
p = "/tmp/probe.bam"
t = time.time()
cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=os.cpu_count())
print("write %.2fs, %.2f GB" % (time.time() - t, os.path.getsize(p) / 1e9), flush=True)
L = cbam._lib()
import ctypes as C
from coverm_amd.native import CovBatch
os.environ["COVERM_BAM_TIMING"] = "1"
for thr in (16, 32, 64, 128, 256):
    err = C.create_string_buffer(512)
    t = time.time()
    h = L.covh_bam_open(p.encode(), thr, 0, err, 512)
    t1 = time.time() - t
    L.covh_bam_close(h)
    print("whole-file threads %d: %.3fs = %.1f M rec/s" % (thr, t1, reads / t1 / 1e6), flush=True)
for win_kb in (32768, 262144, 1048576):
    os.environ["COVERM_KNOBS"] = "stream_window_kb=%d" % win_kb
    for thr in (16, 64, 128, 256):
        st = {}
        t = time.time()
        n = 0
        it = cbam.stream_batches(p, thr, stats=st)
        next(it)
        # count only: do not copy the batches out (the generator copies; measure via the C API directly instead)
        it.close()
        err = C.create_string_buffer(512)
        t = time.time()
        h = L.covh_bam_stream_open(p.encode(), thr, 0, 1, err, 512)
        cb = CovBatch()
        while L.covh_bam_stream_next(h, C.byref(cb)) == 1:
            n += cb.n_records
        t1 = time.time() - t
        tm = (C.c_double * 5)()
        L.covh_bam_stream_timing(h, tm)
        L.covh_bam_stream_close(h)
        print("stream window %d MB threads %d: %.3fs = %.1f M rec/s (%d records) read %.2f inflate %.2f parse %.2f waitI %.2f waitP %.2f" % (
            win_kb >> 10, thr, t1, reads / t1 / 1e6, n, tm[0], tm[1], tm[2], tm[3], tm[4]), flush=True)
