"""Round 5, last A/B: k_prep with tid / cigar_off loaded ONE PASS AHEAD (in registers: k_prep5p; in its first run also through LDS by
global_load_lds), with one record per thread and pass at seven and eight waves per SIMD, against k_prep6, in ONE process on ONE sample (BASELINE config 2: 50 M reads, 5 000 contigs), variants alternating.

Every variant's integer statistics, histogram and estimator floats are compared byte for byte with k_prep6's (which bench.py
checks against the oracle at this size), and the same comparison runs once per variant with the reader-stage filter on, with a target
mask, and with the identity streams — the other compile-time shapes of the kernel.

    python tools/r05/prep_prefetch_ab.py [--reads N] [--steps K] > gpurun_out/.../ab.log
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from coverm_amd import host, synth  # noqa: E402
from coverm_amd.engine import FilterConfig, RecordBatch, Session  # noqa: E402
from coverm_amd.host import CoverageEstimator as E  # noqa: E402

FIELDS = ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")
VARIANTS = [("k_prep6: 2 records/pass, no prefetch, 6 w", {"COVERM_PREP_KERNEL": "6"}),      # the default until these measurements; the byte comparison's base
            ("k_prep5p: 2 records/pass, prefetch, 5 w", {"COVERM_PREP_KERNEL": "5"}),
            ("k_prep7s: 1 record/pass, prefetch, 7 w", {"COVERM_PREP_KERNEL": "7"}),
            ("k_prep8s: 1 record/pass, prefetch, 8 w", {"COVERM_PREP_KERNEL": "8"}),
            ("default (by shape)", {})]
KEYS = ("COVERM_PREP_KERNEL",)
# (profiles/r05_prep_prefetch_ab.log holds the earlier runs, whose builds had more compilations — the prefetch through LDS, every independent
# field ahead, one record per pass at six waves and without the prefetch, two records per pass with it at six waves — under other values of the
# environment variables; all slower, removed)


def session(env, ref, dt, n, est, filt=None, mask=None, want_id=False):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    want_hist, _ = host.wants(est)
    s = Session(0, filt or FilterConfig(), 75, want_hist, want_id)      # cov_create reads the environment
    s.set_targets(ref.lengths, mask)
    if mask is None:
        s.set_estimators(est)      # (with a target mask the entries are genomes: no device estimators)
    s.push_device(dt, n)
    return s


def outputs(s):
    stats, summ = s.finish()
    return stats.tobytes(), s.hist().tobytes(), (s.estimates().tobytes() if getattr(s, "_n_est", 0) else b""), int(summ.n_considered)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--contigs", type=int, default=5000)
    ap.add_argument("--bp", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    t0 = time.time()
    ref = synth.make_reference(a.contigs, a.bp, seed=1)
    batch = synth.make_reads(ref, a.reads, seed=2)
    dt = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in FIELDS}
    torch.cuda.synchronize()
    print("sample: %d reads over %d contigs, generated and uploaded in %.1f s" % (batch.n_records, a.contigs, time.time() - t0), flush=True)
    est = [E.new_estimator_mean(0.0, 75, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, 75),
           E.new_estimator_covered_fraction(0.0), E.new_estimator_variance(0.0, 75)]

    # ---- the other compile-time shapes, once per variant, against the shipped kernel
    rng = np.random.default_rng(7)
    mask = (rng.random(a.contigs) < 0.7).astype(np.uint8)
    shapes = [("plain", dict()),
              ("reader-stage filter", dict(filt=FilterConfig(filter_single=True, min_mapq=10, min_aligned_length=60, min_percent_identity=0.95, min_aligned_percent=0.8))),
              ("target mask", dict(mask=mask)),
              ("identity streams + filter + mask", dict(want_id=True, mask=mask, filt=FilterConfig(filter_single=True, min_aligned_length=50)))]
    ok = True
    for sname, kw in shapes:
        base = None
        for vname, env in VARIANTS:
            s = session(env, ref, dt, batch.n_records, est, **kw)
            o = outputs(s)
            s.close()
            if base is None:
                base = o
                print("shape %-34s %d reads considered by k_prep6" % (sname + ":", o[3]), flush=True)
            else:
                same = o == base
                ok &= same
                print("   %-40s %s" % (vname, "same bytes (statistics, histogram, floats)" if same else "DIFFERENT"), flush=True)

    # ---- timing, variants alternating
    res = {v[0]: [] for v in VARIANTS}
    for rd in range(a.rounds):
        for vname, env in VARIANTS:
            s = session(env, ref, dt, batch.n_records, est)
            for _ in range(3):
                s.finish(); s.estimates()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kp = 0.0
            allk = 0.0
            for _ in range(a.steps):
                s.finish(); s.estimates()
                km = s.kernel_ms()
                kp += km["k_prep"][0]
                allk += sum(v[0] for v in km.values())
            step = (time.perf_counter() - t0) / a.steps * 1e3
            s.close()
            res[vname].append((kp / a.steps, allk / a.steps, step))
            print("round %d  %-40s k_prep %.4f ms   kernels %.4f ms   step %.4f ms" % (rd, vname, kp / a.steps, allk / a.steps, step), flush=True)
    print()
    for vname, _ in VARIANTS:
        r = np.array(res[vname])
        print("%-40s k_prep %.4f ms (%s)   step %.4f ms" % (vname, r[:, 0].mean(), " ".join("%.4f" % x for x in r[:, 0]), r[:, 2].mean()))
    print("all outputs equal to k_prep6's: %s" % ok)
    sys.exit(0 if ok else 4)


if __name__ == "__main__":
    main()
