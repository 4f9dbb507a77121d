#!/usr/bin/env python3
"""Headline benchmark: aligned reads/s through `coverm contig` (BAM -> pileup -> per-contig) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1] at N = 1; configs[3] shape — one BAM per GPU — at N > 1):
50 M synthetic 150 bp reads over 5 000 contigs (1.0 Gbp), coordinate sorted, methods
`mean trimmed_mean covered_fraction variance` (SURVEY.md §8d profile, coverm_amd/synth.py).
A step = one full pass of the hot path over one sample whose record batch is already resident in
HBM: filter + CIGAR expansion + LDS pileup + statistics on the GPU (cov_finish), histogram fetch,
C++ finalisation of every estimator for every contig (coverm_amd.host.contig_coverage) and, for
N > 1, one RCCL gather of the per-contig coverages to rank 0.  Nothing is cached between steps.

Prints ONE JSON line on rank 0 (see the task contract); `roofline` describes the dominant kernel with
live HIP-event timings taken on the session's own stream; `cpu_baseline` times the CPU oracle
(a literal port of the reference's scan + estimators) on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from coverm_amd import host, native, synth  # noqa: E402
from coverm_amd.engine import FilterConfig, Session  # noqa: E402
from coverm_amd.host import CoverageEstimator as E  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
METHODS = ["mean", "trimmed_mean", "covered_fraction", "variance"]


def estimators(excl=75):
    return [E.new_estimator_mean(0.0, excl, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, excl),
            E.new_estimator_covered_fraction(0.0), E.new_estimator_variance(0.0, excl)]


def cpu_baseline(ref, batch, max_seconds=30.0):
    """Times the CPU oracle (oracle/coverm_oracle.c: contig.rs scan loop + estimators, one thread like the
    reference's scan) on a prefix of the same workload sized to finish in roughly max_seconds."""
    import ctypes as C

    from oracle import oracle as O
    from oracle.bamio import BamData
    n = batch.n_records
    # calibrate on 1/50 of the records (whole contigs), then scale the sample
    def run(n_rec):
        n_rec = min(n_rec, n)
        last_tid = int(batch.tid[n_rec - 1])
        n_rec = int(np.searchsorted(batch.tid, last_tid, side="left")) if n_rec < n else n
        if n_rec == 0:
            n_rec = int(np.searchsorted(batch.tid, last_tid, side="right"))
        z = np.zeros(n_rec, np.int32)
        b = BamData(ref.names, ref.lengths, batch.tid[:n_rec], batch.pos[:n_rec], batch.flag[:n_rec],
                    batch.mapq[:n_rec], batch.l_seq[:n_rec].astype(np.int32), batch.nm[:n_rec], batch.nm_kind[:n_rec],
                    batch.cigar_off[:n_rec + 1], batch.cigar, z, z, z, [], "")
        est = [O.est_mean(0.0, 75, False), O.est_trimmed_mean(0.05, 0.95, 0.0, 75), O.est_covered_fraction(0.0),
               O.est_variance(0.0, 75)]

        class Null:
            def start_stoit(self, n): pass
        r, keep = O._records(b, None)
        tl = np.ascontiguousarray(b.ref_lens, np.int64)
        out = O._Out()
        rm = O._ReadsMapped()
        ff = O.FlagFilter(True, True, False).c()
        prim = int(((b.flag & 0x900) == 0).sum())
        t0 = time.perf_counter()
        rc = O.lib().orc_contig_coverage(C.byref(r), tl.ctypes.data_as(C.c_void_p), C.c_int32(len(tl)), O._params(est),
                                         C.c_int32(len(est)), C.c_int32(0), C.byref(ff), C.c_uint64(prim),
                                         C.byref(out), C.byref(rm))
        dt = time.perf_counter() - t0
        O.lib().orc_out_free(C.byref(out))
        assert rc == 0
        return int(rm.num_mapped_reads), dt, n_rec
    reads, dt, n_rec = run(max(1000, n // 50))
    if n_rec < n:
        scale = min(n / n_rec, max(1.0, 0.6 * max_seconds / max(dt, 1e-3)))
        reads, dt, n_rec = run(int(n_rec * scale))
    return dict(value=reads / dt, unit="aligned reads/s", cores=1, kind="port",
                sample="first %d of %d records (%.0f%% of the workload, whole contigs), %.1f s; oracle/coverm_oracle.c "
                       "= literal C port of CoverM 0.8.0's scan loop + estimators, not the coverm binary"
                       % (n_rec, n, 100.0 * n_rec / n, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("COVERM_BENCH_READS", 50_000_000)))
    ap.add_argument("--contigs", type=int, default=5000)
    ap.add_argument("--bp", type=int, default=1_000_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the coverage engine has no CPU fallback")
    # Functional check of the N > 1 path on a single-GPU box: COVERM_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # exchanges over gloo (RCCL refuses two ranks on one device).  Never set by the driver; such a line is not a measurement.
    share = os.environ.get("COVERM_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = torch.device("cpu") if share else dev          # where the exchanged tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- synthetic sample of this rank (one BAM per GPU; seed 2 at N=1, 10+rank otherwise)
    t0 = time.time()
    ref = synth.make_reference(a.contigs, a.bp, seed=1)
    batch = synth.make_reads(ref, a.reads, seed=2 if world == 1 else 10 + rank)
    gen_s = time.time() - t0
    dt = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in
          ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")}
    torch.cuda.synchronize()

    est = estimators()
    want_hist, want_id = host.wants(est)
    sess = Session(local_rank, FilterConfig(), 75, want_hist, want_id)
    sess.set_targets(ref.lengths)
    sess.push_device(dt, batch.n_records)
    n_cov = len(ref.lengths) * len(est)
    gather_buf = [torch.empty(n_cov, dtype=torch.float32, device=xdev) for _ in range(world)] if (dist and rank == 0) else None

    def step():
        stats, summ = sess.finish()
        hist = sess.hist()
        taker = host.CoverageTaker.new_cached_single_float_coverage_taker(len(est))
        sample = host.SampleResult("sample%d" % rank, stats, hist, int(summ.num_detected_primary_alignments))
        rm = host.contig_coverage(ref.names, ref.lengths, [sample], taker, est, True)
        if dist:
            # per-contig coverages of this sample -> rank 0 (one gather over RCCL/xGMI)
            cov = taker.cached_coverages(0)   # n_contigs x n_estimators f32, contig order (zeros printed)
            t = torch.from_numpy(cov).to(xdev)
            dist.gather(t, gather_buf, dst=0)
        return summ, rm, taker

    for _ in range(a.warmup):
        summ, rm, taker = step()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = {k: 0.0 for k in native.KERNEL_NAMES.values()}
    for _ in range(a.steps):
        summ, rm, taker = step()
        for k, v in sess.kernel_ms().items():
            kms[k] += v[0]
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    considered = int(summ.n_considered)
    if dist:
        tt = torch.tensor([elapsed, float(considered)], dtype=torch.float64, device=xdev)
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        total_reads = int(tt[1].item())
    else:
        total_reads = considered

    if rank == 0:
        for k in kms:
            kms[k] /= a.steps
        dom = max(kms, key=kms.get)
        kid = {v: k for k, v in native.KERNEL_NAMES.items()}[dom]
        R = batch.n_records
        ncig = int(batch.cigar_off[-1]) - int(batch.cigar_off[0])
        # algorithmic HBM bytes per launch (DESIGN.md "Algorithmic bytes")
        kbytes = {"k_prep": R * 24 + ncig * 4 + R * 8,                    # SoA + CIGAR read once, run words written
                  "k_pileup": R * 8 + sess_tiles(sess, ref) * 32 + len(ref.lengths) * 160,  # run words + tile descriptors + results
                  "k_ranges": sess_tiles(sess, ref) * 40, "k_identity": R * 32, "k_hist": 0}
        achieved = kbytes[dom] / (kms[dom] * 1e-3) / 1e9 if kms[dom] > 0 else 0.0
        pipe_bytes = sess.algorithmic_bytes()
        pipe_ms = sum(kms.values())
        # what an arena-in-HBM design would have to move for the same job (SURVEY.md §8d formula)
        A = int((ref.lengths + 1).sum())
        E_runs = int((((batch.cigar & 15) == 0) | ((batch.cigar & 15) == 7) | ((batch.cigar & 15) == 8)).sum())
        arena_bytes = R * 24 + ncig * 4 + E_runs * 16 + A * 4 * 3
        aligned_bp = synth.aligned_bases(batch) * (considered / max(1, R))
        out = {
            "metric": "aligned reads/s through coverm contig (mean trimmed_mean covered_fraction variance)",
            "value": total_reads * a.steps / elapsed, "unit": "aligned reads/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32 depth / u64 sums / f32 estimators", "data": "synthetic",
            "config": {"workload": "coverm contig, %d-read synthetic sorted BAM over %d contigs (%.2f Gbp) per GPU, "
                                   "--methods %s, records resident in HBM" % (a.reads, a.contigs,
                                                                             ref.lengths.sum() / 1e9, " ".join(METHODS)),
                       "reads_per_gpu": a.reads, "contigs": a.contigs, "reference_bp": int(ref.lengths.sum()),
                       "samples": world, "sharding": ("FUNCTIONAL CHECK ONLY: ranks share one GPU, gloo exchange" if share else
                                                       "one sample per GPU, RCCL gather of per-contig coverages") if world > 1 else "single GPU"},
            "gbp_per_s": aligned_bp * world * a.steps / elapsed / 1e9,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": profile_traffic(dom),
                         "kernel_ms": kms[dom], "algorithmic_bytes": kbytes[dom],
                         "all_kernels_ms": kms,
                         "note": "per-base depth lives only in LDS, so the dominant kernel's compulsory HBM bytes are ~0.4 GB "
                                 "per launch and its HBM fraction is small by construction; it is bound by VALU issue (~55-60 % busy, "
                                 "4 waves/SIMD) and dependent LDS round trips (DESIGN.md sections 4-5, profiles/r01g_*); "
                                 "SURVEY 8d's arena-in-HBM byte count for the same job is reported as arena_design_equivalent",
                         "pipeline": {"algorithmic_bytes": pipe_bytes, "kernels_ms": pipe_ms,
                                      "achieved_GBps": pipe_bytes / (pipe_ms * 1e-3) / 1e9 if pipe_ms else 0.0},
                         "arena_design_equivalent": {"bytes": arena_bytes,
                                                     "GBps_if_moved_in_same_time": arena_bytes / (pipe_ms * 1e-3) / 1e9 if pipe_ms else 0.0}},
            "host": {"generation_s": gen_s, "nproc": os.cpu_count()},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ref, batch)
        print(json.dumps(out), flush=True)
    sess.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def sess_tiles(sess, ref):
    """Tiles of the configured pileup kernel: 1024 bases per wave (streaming kernel) unless COVERM_PILEUP=tile."""
    tile = int(os.environ.get("COVERM_TILE", 4096)) if os.environ.get("COVERM_PILEUP") == "tile" else 1024
    return int(((ref.lengths + tile - 1) // tile).sum())


def profile_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of this workload, if present."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as fh:
            return json.load(fh).get(kernel)
    except Exception:
        return None


if __name__ == "__main__":
    main()
