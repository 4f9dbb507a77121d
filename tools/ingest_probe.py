"""Device-ingest probe: realistic-entropy synthetic BAM -> cov_ingest (GPU inflate + parse) vs the streamed CPU reader."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coverm_amd import bam as cbam, synth  # noqa: E402
from coverm_amd.engine import FilterConfig, Session  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
p = "/tmp/probe.bam"
if not os.path.exists(p) or os.environ.get("REGEN"):
    ref = synth.make_reference(5000, 1_000_000_000, seed=1)
    b = synth.make_reads(ref, reads, seed=3)
    t = time.time()
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=threads)
    print("write %.2fs, %.2f GB" % (time.time() - t, os.path.getsize(p) / 1e9), flush=True)
for it in range(3):
    with Session(0, FilterConfig(), 75, want_hist=True) as s:
        t = time.time()
        names, lens, n, tm = cbam.gpu_ingest(s, p, threads=threads, check_crc=(it != 2))
        t1 = time.time() - t
        t = time.time()
        st, su = s.finish()
        t2 = time.time() - t
        print("gpu ingest it %d (crc %s): %d records in %.3fs = %.1f M rec/s | read %.3f slot_wait %.3f end %.3f | finish %.4fs" % (
            it, it != 2, n, t1, n / t1 / 1e6, tm["read"], tm["slot_wait"], tm["end"], t2), flush=True)
        chk = (int(st["n_pass"].sum()), int(st["win_sum_d"].sum()))
with Session(0, FilterConfig(), 75, want_hist=True) as s:
    t = time.time()
    it = cbam.stream_batches(p, threads)
    names, lens = next(it)
    s.set_targets(lens)
    n = 0
    for batch in it:
        s.push(batch)
        n += batch.n_records
    t1 = time.time() - t
    st, su = s.finish()
    print("cpu stream + push: %d records in %.3fs = %.1f M rec/s" % (n, t1, n / t1 / 1e6))
    assert chk == (int(st["n_pass"].sum()), int(st["win_sum_d"].sum())), "device ingest and CPU reader disagree"
    print("results agree")
