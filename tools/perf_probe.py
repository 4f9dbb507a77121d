"""Ad-hoc performance probe: synthetic workload -> repeated cov_finish, per-kernel HIP-event timings."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from coverm_amd import synth  # noqa: E402
from coverm_amd.engine import FilterConfig, Session  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--contigs", type=int, default=1000)
ap.add_argument("--bp", type=int, default=200_000_000)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--hist", type=int, default=1)
ap.add_argument("--identity", type=int, default=0, help="1 both identity sums, 2 primary-read sum only")
ap.add_argument("--filter", type=int, default=0)
ap.add_argument("--long", type=int, default=0, help="mean read length of a long-read profile (0 = short reads)")
ap.add_argument("--excl", type=int, default=75)
a = ap.parse_args()

t = time.time()
ref = synth.make_reference(a.contigs, a.bp, seed=1, **(dict(min_len=20 * a.long, max_len=2000 * a.long) if a.long else {}))
batch = synth.make_long_reads(ref, a.reads, seed=2, mean_len=a.long) if a.long else synth.make_reads(ref, a.reads, seed=2)
print("generated %d reads over %d contigs (%.0f Mbp) in %.1fs" % (a.reads, a.contigs, ref.lengths.sum() / 1e6,
                                                                   time.time() - t), flush=True)
filt = FilterConfig(include_improper_pairs=not a.filter, filter_single=bool(a.filter), min_aligned_length=50 if a.filter else 0,
                    min_percent_identity=0.95 if a.filter else 0.0)
with Session(0, filt, a.excl, want_hist=bool(a.hist), want_identity=("primary" if a.identity == 2 else bool(a.identity))) as s:
    s.set_targets(ref.lengths)
    t = time.time()
    s.push(batch)
    st, summ = s.finish()
    print("push+first finish %.3fs, considered %d" % (time.time() - t, summ.n_considered))
    for it in range(a.iters):
        t = time.time()
        st, summ = s.finish()
        t1 = time.time()
        h = s.hist() if a.hist else None
        t2 = time.time()
        km = s.kernel_ms()
        print("iter %d: finish %.3f ms, hist fetch %.3f ms (%d bins) | " % (it, (t1 - t) * 1e3, (t2 - t1) * 1e3,
              0 if h is None else len(h)) + " ".join("%s=%.3f" % (k, v[0]) for k, v in km.items()), flush=True)
    ab = s.algorithmic_bytes()
    kp = s.kernel_ms()
    tot = sum(v[0] for v in kp.values())
    print("algorithmic bytes %.3f GB; sum kernels %.3f ms -> %.1f GB/s; reads/s (finish wall) %.3e" % (
        ab / 1e9, tot, ab / 1e6 / tot, summ.n_considered / (t1 - t)))
    if a.excl == 0:   # conservation law: summed depth == aligned M/=/X bases of the considered records
        op = batch.cigar & 15
        ln = (batch.cigar >> 4).astype(np.int64)
        m = np.where((op == 0) | (op == 7) | (op == 8), ln, 0)
        per = np.add.reduceat(np.concatenate([m, [0]]), batch.cigar_off[:-1].astype(np.int64))
        per[batch.cigar_off[1:] == batch.cigar_off[:-1]] = 0
        cons = ((batch.flag & 0x4) == 0) & ((batch.flag & 0x100) == 0) & ((batch.flag & 0x2) != 0 if a.filter else True)
        if not a.filter:
            want = int(per[cons].sum())
            print("conservation: sum depth %d vs aligned bases %d -> %s" % (int(st["win_sum_d"].sum()), want,
                  "OK" if int(st["win_sum_d"].sum()) == want else "MISMATCH"))
