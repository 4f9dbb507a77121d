"""Worker for tests/test_distributed_gloo.py: run with torch.distributed.run, backend gloo (CPU) or nccl (GPU).

Checks both sharding modes of coverm_amd.distributed end to end: every rank produces per-contig statistics for its
shard (from the CPU oracle with gloo, from the HIP engine with nccl), rank 0 gathers, runs the C++ scan drivers and
compares the text with the unsharded single-process result.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from coverm_amd import distributed, host, synth  # noqa: E402
from tests import harness_cli as cli  # noqa: E402
from tests.harness_cli import AlignmentFile  # noqa: E402
from coverm_amd.host import CoverageEstimator as E  # noqa: E402


def provider(backend):
    if backend == "nccl":
        return cli.device_sample
    from oracle import oracle as O
    from oracle.bamio import BamData
    from coverm_amd import native

    def oracle_sample(af, fp, excl, want_hist, want_identity, mask=None, device=0):
        r = af.records
        z = np.zeros(r.n_records, np.int32)
        b = BamData(af.ref_names, af.ref_lens, r.tid, r.pos, r.flag, r.mapq, r.l_seq.astype(np.int32), r.nm, r.nm_kind,
                    r.cigar_off, r.cigar, z, z, z, [], "")
        off = O.FlagFilter(fp.flag_filters.include_improper_pairs, fp.flag_filters.include_supplementary,
                           fp.flag_filters.include_secondary)
        st, hist, prim = O.integer_stats(b, off, None, excl, mask)
        out = np.zeros(len(st), dtype=native.CONTIG_STATS_DTYPE)
        for f in ("n_primary", "n_pass", "n_nonsupp", "sum_nm", "sum_indel", "win_sum_d", "win_sum_d2", "win_covered",
                  "full_covered", "first_record", "last_record", "win_min_d", "win_max_d", "hist_len", "hist_off"):
            out[f] = st[f]
        out["sum_identity_primary"] = st["id_primary"]
        out["sum_identity_nonsupp"] = st["id_nonsupp"]
        return host.SampleResult(af.stoit_name, out, hist if want_hist else None, prim)
    return oracle_sample


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl", device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group("gloo")
    prov = provider(backend)
    est = [E.new_estimator_mean(0.0, 75, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, 75),
           E.new_estimator_covered_fraction(0.0), E.new_estimator_variance(0.0, 75), E.new_estimator_read_count()]
    want_hist, want_id = host.wants(est)
    fp = cli.FilterParameters()
    ref = synth.make_reference(37, 1_500_000, seed=21, min_len=1200, max_len=200_000)

    # ---- mode 1: one sample per rank
    mine = synth.make_reads(ref, 20_000, seed=100 + rank)
    af = AlignmentFile("s%d.bam" % rank, ref.names, ref.lengths, mine)
    local = prov(af, fp, 75, want_hist, want_id, device=int(os.environ.get("LOCAL_RANK", 0)))
    gathered = distributed.gather_samples(local, dist, device)
    if rank == 0:
        taker = host.CoverageTaker.new_cached_single_float_coverage_taker(len(est))
        rm = host.contig_coverage(ref.names, ref.lengths, gathered, taker, est, True)
        headers = [h for e in est for h in e.column_headers()]
        host.finalise_printing(taker, host.PRINTER_DENSE, "Contig", headers, rm, [], None, None)
        # single-process expectation
        exp_samples = [prov(AlignmentFile("s%d.bam" % r, ref.names, ref.lengths, synth.make_reads(ref, 20_000, seed=100 + r)),
                            fp, 75, want_hist, want_id) for r in range(world)]
        t2 = host.CoverageTaker.new_cached_single_float_coverage_taker(len(est))
        rm2 = host.contig_coverage(ref.names, ref.lengths, exp_samples, t2, est, True)
        host.finalise_printing(t2, host.PRINTER_DENSE, "Contig", headers, rm2, [], None, None)
        assert taker.text() == t2.text(), "by-sample sharding changed the output"
        assert [(r.num_mapped_reads, r.num_reads) for r in rm] == [(r.num_mapped_reads, r.num_reads) for r in rm2]

    # ---- mode 1b: samples with DIFFERENT reference sets (each BAM brings its own header, contig.rs:29-32) and a stoit name that
    # is longer than the header field and would be cut inside a UTF-8 sequence
    def odd_sample(r):
        ref_r = synth.make_reference(11 + 5 * r, 400_000, seed=40 + r, min_len=1200, max_len=90_000)
        name = "s%d_" % r + "\u00e9" * 140          # 2 bytes per character: the 256-byte cut falls inside one
        return prov(AlignmentFile(name + ".bam", ref_r.names, ref_r.lengths, synth.make_reads(ref_r, 5_000, seed=60 + r)), fp, 75,
                    want_hist if r % 2 == 0 else False, want_id, device=int(os.environ.get("LOCAL_RANK", 0)))
    local = odd_sample(rank)
    gathered = distributed.gather_samples(local, dist, device)
    if rank == 0:
        for r, g in enumerate(gathered):
            e = odd_sample(r)
            assert len(g.stats) == len(e.stats) == 11 + 5 * r and g.stats.tobytes() == e.stats.tobytes(), "unequal reference sets: rows changed"
            assert (g.hist is None) == (e.hist is None) and (g.hist is None or (g.hist == e.hist).all())
            assert g.num_detected_primary_alignments == e.num_detected_primary_alignments
            assert e.stoit_name.startswith(g.stoit_name) and len(g.stoit_name.encode()) in (255, 256), g.stoit_name

    # ---- mode 2: one sample split by tid range
    whole = synth.make_reads(ref, 50_000, seed=7)
    shards = distributed.tid_range_shards(ref.lengths, world)
    lo, hi = shards[rank]
    part = distributed.shard_records(whole, lo, hi, include_unplaced=(rank == world - 1))
    af = AlignmentFile("whole.bam", ref.names, ref.lengths, part)
    local = prov(af, fp, 75, want_hist, want_id, device=int(os.environ.get("LOCAL_RANK", 0)))
    merged = distributed.gather_tid_shards(local, (lo, hi), dist, device)
    if rank == 0:
        taker = host.CoverageTaker.new_single_float_coverage_streaming_coverage_printer()
        rm = host.contig_coverage(ref.names, ref.lengths, [merged], taker, est, True)
        exp = prov(AlignmentFile("whole.bam", ref.names, ref.lengths, whole), fp, 75, want_hist, want_id)
        t2 = host.CoverageTaker.new_single_float_coverage_streaming_coverage_printer()
        rm2 = host.contig_coverage(ref.names, ref.lengths, [exp], t2, est, True)
        assert taker.text() == t2.text(), "tid-range sharding changed the output"
        assert (rm[0].num_mapped_reads, rm[0].num_reads) == (rm2[0].num_mapped_reads, rm2[0].num_reads)
        assert sum(h - l for l, h in shards) == len(ref.lengths)
        print("DIST_OK world=%d backend=%s" % (world, backend), flush=True)
    # ---- the host-side wait bench.py ends with (ranks > 0 must not return before rank 0 arrives, and no collective is involved)
    import time
    t0 = time.time()
    if rank == 0:
        time.sleep(1.5)
    used_store = distributed.wait_for_root(dist, rank, "dist_worker_done", minutes=2)
    waited = time.time() - t0
    assert used_store, "the process group's store was not reachable"
    assert rank == 0 or waited >= 1.0, "rank %d returned after %.2f s, before rank 0 arrived" % (rank, waited)
    if rank == world - 1:
        print("WAIT_OK rank=%d waited=%.2f" % (rank, waited), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
