"""Round 6: two samples in flight on ONE GPU.  Two sessions (each has its own stream), each holding a BASELINE config 2
sample (different seeds) in HBM; cov_finish + the floats' fetch run (a) one session at a time, (b) both at once from two host threads.
Aggregate aligned reads/s of both modes, the kernel groups' times as each session's own events saw them, outputs compared with the
sequential run's.

    python tools/r06/two_samples.py [--reads N] [--steps K]
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from coverm_amd import host, synth  # noqa: E402
from coverm_amd.engine import FilterConfig, Session  # noqa: E402
from coverm_amd.host import CoverageEstimator as E  # noqa: E402

FIELDS = ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ref = synth.make_reference(5000, 1_000_000_000, seed=1)
    est = [E.new_estimator_mean(0.0, 75, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, 75),
           E.new_estimator_covered_fraction(0.0), E.new_estimator_variance(0.0, 75)]
    want_hist, _ = host.wants(est)
    sess, keep, considered = [], [], []
    for seed in (2, 12):
        batch = synth.make_reads(ref, a.reads, seed=seed)
        dt = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in FIELDS}
        keep.append(dt)
        s = Session(0, FilterConfig(), 75, want_hist, False)
        s.set_targets(ref.lengths)
        s.set_estimators(est)
        s.push_device(dt, batch.n_records)
        sess.append(s)
    torch.cuda.synchronize()

    def run(s, n, out):
        acc = {}
        for _ in range(n):
            st, summ = s.finish()
            f = s.estimates()
            for k, v in s.kernel_ms().items():
                acc[k] = acc.get(k, 0.0) + v[0]
        out.append((st.tobytes(), f.tobytes(), int(summ.n_considered), {k: v / n for k, v in acc.items() if v > 0}))

    base = []
    for s in sess:
        run(s, 3, [])
    t0 = time.perf_counter()
    for s in sess:
        run(s, a.steps, base)
    t_seq = time.perf_counter() - t0
    reads = sum(b[2] for b in base)
    both = [[], []]
    th = [threading.Thread(target=run, args=(sess[k], 3, [])) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(sess[k], a.steps, both[k])) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    t_con = time.perf_counter() - t0
    same = all(both[k][0][:3] == base[k][:3] for k in range(2))
    res = {"reads_per_sample": a.reads, "steps": a.steps,
           "sequential": {"wall_s": t_seq, "aligned_reads_per_s": reads * a.steps / t_seq, "kernel_ms": [b[3] for b in base]},
           "concurrent": {"wall_s": t_con, "aligned_reads_per_s": reads * a.steps / t_con, "kernel_ms_as_each_sessions_events_saw_them": [both[k][0][3] for k in range(2)]},
           "concurrent_over_sequential": t_seq / t_con, "outputs_equal": same}
    print(json.dumps(res, indent=1))
    for s in sess:
        s.close()
    sys.exit(0 if same else 4)


if __name__ == "__main__":
    main()
