"""Round 6 A/B harness: kernel variants (selected by environment variables that cov_create reads) alternating in ONE process on ONE sample.

Every variant's integer statistics, histogram and estimator floats are compared byte for byte with the FIRST variant's — over the
compile-time shapes of k_prep (plain, reader-stage filter, target mask, identity streams + filter + mask) — then the variants are timed
alternating: per kernel group (cov_kernel_ms) and the host's step.

    python tools/r06/kernel_ab.py --variant base=COVERM_PREP_KERNEL:7 --variant lean= [--reads N --contigs C --bp B --steps K --rounds R] [--shapes 0|1]
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
from coverm_amd.engine import FilterConfig, Session  # noqa: E402
from coverm_amd.host import CoverageEstimator as E  # noqa: E402

FIELDS = ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")


def parse_variant(v):
    name, _, envs = v.partition("=")
    env = {}
    for kv in filter(None, envs.split(",")):
        k, _, val = kv.partition(":")
        env[k] = val
    return name, env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", action="append", required=True)
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--contigs", type=int, default=5000)
    ap.add_argument("--bp", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--shapes", type=int, default=1)
    ap.add_argument("--min-len", type=int, default=0)
    a = ap.parse_args()
    variants = [parse_variant(v) for v in a.variant]
    keys = sorted({k for _, e in variants for k in e})
    dev = torch.device("cuda", 0)
    t0 = time.time()
    kw = dict(min_len=a.min_len) if a.min_len else {}
    ref = synth.make_reference(a.contigs, a.bp, seed=1, **kw)
    batch = synth.make_reads(ref, a.reads, seed=2)
    dt = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in FIELDS}
    torch.cuda.synchronize()
    print("sample: %d reads over %d contigs (%d bp), generated and uploaded in %.1f s" % (batch.n_records, a.contigs, int(ref.lengths.sum()), time.time() - t0), flush=True)
    est = [E.new_estimator_mean(0.0, 75, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, 75),
           E.new_estimator_covered_fraction(0.0), E.new_estimator_variance(0.0, 75)]

    def session(env, filt=None, mask=None, want_id=False):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        want_hist, _ = host.wants(est)
        s = Session(0, filt or FilterConfig(), 75, want_hist, want_id)      # cov_create reads the environment
        s.set_targets(ref.lengths, mask)
        if mask is None:
            s.set_estimators(est)      # (with a target mask the entries are genomes: no device estimators)
        s.push_device(dt, batch.n_records)
        return s

    def outputs(s):
        stats, summ = s.finish()
        return stats.tobytes(), s.hist().tobytes(), (s.estimates().tobytes() if getattr(s, "_n_est", 0) else b""), int(summ.n_considered), int(summ.num_detected_primary_alignments)

    rng = np.random.default_rng(7)
    mask = (rng.random(a.contigs) < 0.7).astype(np.uint8)
    shapes = [("plain", dict())]
    if a.shapes:
        shapes += [("reader-stage filter", dict(filt=FilterConfig(filter_single=True, min_mapq=10, min_aligned_length=60, min_percent_identity=0.95, min_aligned_percent=0.8))),
                   ("target mask", dict(mask=mask)),
                   ("identity streams + filter + mask", dict(want_id=True, mask=mask, filt=FilterConfig(filter_single=True, min_aligned_length=50)))]
    ok = True
    for sname, kw2 in shapes:
        base = None
        for vname, env in variants:
            s = session(env, **kw2)
            o = outputs(s)
            s.close()
            if base is None:
                base = o
                print("shape %-34s %d reads considered, %d primaries (%s)" % (sname + ":", o[3], o[4], vname), flush=True)
            else:
                same = o == base
                ok &= same
                what = [n for n, x, y in zip(("statistics", "histogram", "floats", "considered", "primaries"), o, base) if x != y]
                print("   %-40s %s" % (vname, "same bytes (statistics, histogram, floats)" if same else "DIFFERENT: " + ", ".join(what)), flush=True)

    res = {v[0]: [] for v in variants}
    names = None
    for rd in range(a.rounds):
        for vname, env in variants:
            s = session(env)
            for _ in range(3):
                s.finish(); s.estimates()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            acc = {}
            for _ in range(a.steps):
                s.finish(); s.estimates()
                for k, v in s.kernel_ms().items():
                    acc[k] = acc.get(k, 0.0) + v[0]
            step = (time.perf_counter() - t0) / a.steps * 1e3
            s.close()
            names = names or [k for k in acc if acc[k] > 0]
            row = [acc.get(k, 0.0) / a.steps for k in names] + [step]
            res[vname].append(row)
            print("round %d  %-28s %s   step %.4f ms" % (rd, vname, "  ".join("%s %.4f" % (k, x) for k, x in zip(names, row)), step), flush=True)
    print()
    for vname, _ in variants:
        r = np.array(res[vname])
        print("%-28s %s   step %.4f ms (min %.4f)" % (vname, "  ".join("%s %.4f" % (k, x) for k, x in zip(names, r.mean(0))), r[:, -1].mean(), r[:, -1].min()))
    print("all outputs equal to the first variant's: %s" % ok)
    sys.exit(0 if ok else 4)


if __name__ == "__main__":
    main()
