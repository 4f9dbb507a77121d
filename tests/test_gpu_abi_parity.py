"""GPU parity tests through the C ABI (libcovermhip.so) against the CPU oracle.

Bar: every integer (depth arrays, window/full statistics, histograms, read counters) bit-exact;
identity sums bit-exact as well (the device keeps the reference's file-order f64 summation).
"""
import numpy as np
import pytest

from coverm_amd import synth
from coverm_amd.engine import FilterConfig, RecordBatch, Session
from coverm_amd.native import CovError, ERR_NM_MISSING, ERR_POS_OOB, ERR_UNSORTED
from oracle import oracle as O
from oracle.bamio import BamData
from tests.fixtures import load_fixture
from tests.golden import cases

pytestmark = pytest.mark.gpu


def to_batch(b: BamData) -> RecordBatch:
    return RecordBatch.from_arrays(b.tid, b.pos, b.flag, b.mapq, b.nm, b.nm_kind, b.l_seq, b.cigar_off, b.cigar)


def to_bamdata(batch: RecordBatch, ref_lens, names=None) -> BamData:
    n = batch.n_records
    names = names or ["c%d" % i for i in range(len(ref_lens))]
    z = np.zeros(n, np.int32)
    return BamData(names, np.asarray(ref_lens, np.int64), batch.tid, batch.pos, batch.flag, batch.mapq,
                   batch.l_seq.astype(np.int32), batch.nm, batch.nm_kind, batch.cigar_off, batch.cigar, z, z, z,
                   [b"r%d" % i for i in range(n)] if n < 100000 else [], "")


def compare(b: BamData, ff=(True, True, False), fp=None, excl=75, mask=None, check_depth=(), chunks=1, paths_out=None):
    off = O.FlagFilter(*ff)
    ofp = None
    filt = FilterConfig(*ff)
    if fp is not None:
        ofp = O.FilterParameters(off, **fp)
        fs, fpair = O.filter_mode(ofp)
        assert fs and not fpair, "ABI covers the single-read branch"
        filt = FilterConfig(*ff, filter_single=True, min_mapq=ofp.min_mapq,
                            min_aligned_length=ofp.min_aligned_length_single,
                            min_percent_identity=ofp.min_percent_identity_single,
                            min_aligned_percent=ofp.min_aligned_percent_single)
    exp, exp_hist, prim = O.integer_stats(b, off, ofp, excl, mask)
    batch = to_batch(b)
    with Session(0, filt, excl, want_hist=True, want_identity=True) as s:
        s.set_targets(b.ref_lens, mask)
        n = batch.n_records
        edges = np.linspace(0, n, chunks + 1).astype(int)
        for lo, hi in zip(edges[:-1], edges[1:]):
            s.push(batch.slice(lo, hi))
        st, summ = s.finish()
        if paths_out is not None:
            paths_out.update(s.last_paths())        # cov_last_paths: which branches of the pipeline this finish took
        hist = s.hist()
        assert summ.num_detected_primary_alignments == prim
        live = exp["seen"] == 1 if mask is None else (exp["seen"] == 1) & (np.asarray(mask) != 0)
        for f in ("n_primary", "n_pass", "n_nonsupp"):
            np.testing.assert_array_equal(st[f], exp[f], err_msg=f)
        for f in ("sum_nm", "sum_indel", "win_sum_d", "win_sum_d2", "win_covered", "full_covered", "win_min_d",
                  "win_max_d"):
            np.testing.assert_array_equal(st[f][live], exp[f][live], err_msg=f)
            assert (st[f][~live] == 0).all(), f
        seen = exp["seen"] == 1
        np.testing.assert_array_equal(st["first_record"][seen], exp["first_record"][seen])
        np.testing.assert_array_equal(st["last_record"][seen], exp["last_record"][seen])
        # identity sums: bit-exact f64
        np.testing.assert_array_equal(st["sum_identity_primary"][live].view(np.uint64),
                                      exp["id_primary"][live].view(np.uint64))
        np.testing.assert_array_equal(st["sum_identity_nonsupp"][live].view(np.uint64),
                                      exp["id_nonsupp"][live].view(np.uint64))
        # histograms
        np.testing.assert_array_equal(st["hist_len"][live], exp["hist_len"][live])
        assert (st["hist_len"][~live] == 0).all()
        for t in np.nonzero(live)[0]:
            n_b = int(exp["hist_len"][t])
            np.testing.assert_array_equal(hist[int(st["hist_off"][t]):int(st["hist_off"][t]) + n_b],
                                          exp_hist[int(exp["hist_off"][t]):int(exp["hist_off"][t]) + n_b],
                                          err_msg="hist of contig %d" % t)
        for t in check_depth:
            order, _ = O.reader_stage(b, ofp)
            ud = O.contig_deltas(b, off, t, order)
            np.testing.assert_array_equal(s.depth(t), np.cumsum(ud, dtype=np.int64).astype(np.int32),
                                          err_msg="depth of contig %d" % t)
    return st


FIXTURES = [f for f in cases.FIXTURE_FILES if "unsorted" not in f]


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("excl", [0, 75])
def test_fixture_stats_and_depth(name, excl):
    b = load_fixture(name)
    touched = sorted(set(int(t) for t in b.tid if t >= 0))[:4]
    if name.startswith("1read"):   # unmapped mate without NM: default flags include it? it is unmapped -> fine
        pass
    compare(b, ff=(True, True, False), excl=excl, check_depth=touched)


@pytest.mark.parametrize("ff", [(True, False, False), (False, True, True), (True, True, True)])
def test_fixture_flag_filters(ff):
    for name in ["7seqs.reads_for_seq1_and_seq2.bam", "2seqs.bad_read.1.with_supplementary.bam",
                 "k141_2005182.bam", "eg2.bam"]:
        compare(load_fixture(name), ff=ff, excl=75)


def test_fixture_single_read_filter():
    for name, fp in [("2seqs.bad_read.1.bam", dict(min_percent_identity_single=0.99)),
                     ("mapq_test.sam", dict(min_mapq=51)),
                     ("k141_2005182.head11.bam", dict(min_percent_identity_single=float(np.float32(0.97001)))),
                     ("eg2.bam", dict(min_aligned_length_single=100, min_aligned_percent_single=0.9,
                                      min_percent_identity_single=0.95))]:
        compare(load_fixture(name), ff=(True, True, True), fp=fp, excl=75)


def test_unsorted_error():
    b = load_fixture("2seqs.bad_read.1.unsorted.bam")
    with Session(0, FilterConfig(), 75) as s:
        s.set_targets(b.ref_lens)
        s.push(to_batch(b))
        with pytest.raises(CovError) as ei:
            s.finish()
        assert ei.value.status == ERR_UNSORTED
        assert "BAM file appears to be unsorted" in ei.value.message


def test_nm_missing_and_pos_oob_errors():
    b = load_fixture("7seqs.reads_for_seq1_and_seq2.bam")
    bad = to_batch(b)
    bad.nm_kind = bad.nm_kind.copy(); bad.nm_kind[5] = 0
    with Session(0, FilterConfig(), 75) as s:
        s.set_targets(b.ref_lens); s.push(bad)
        with pytest.raises(CovError) as ei:
            s.finish()
        assert ei.value.status == ERR_NM_MISSING and "record 5" in ei.value.message
    bad = to_batch(b)
    bad.pos = bad.pos.copy(); bad.pos[3] = int(b.ref_lens[b.tid[3]]) + 7
    with Session(0, FilterConfig(), 75) as s:
        s.set_targets(b.ref_lens); s.push(bad)
        with pytest.raises(CovError) as ei:
            s.finish()
        assert ei.value.status == ERR_POS_OOB


@pytest.mark.parametrize("n_contigs,total,n_reads,chunks", [(40, 3_000_000, 60_000, 1), (300, 20_000_000, 400_000, 3)])
def test_synthetic_all_ops(n_contigs, total, n_reads, chunks):
    ref = synth.make_reference(n_contigs, total, seed=11, min_len=1500, max_len=400_000)
    batch = synth.make_reads(ref, n_reads, seed=12)
    b = to_bamdata(batch, ref.lengths, ref.names)
    ops = np.bincount(batch.cigar & 15, minlength=9)
    assert ops[7] and ops[8] and ops[3] and ops[2] and ops[1] and ops[4]   # = X N D I S all present
    compare(b, ff=(True, True, False), excl=75, check_depth=[0, 1, n_contigs // 2, n_contigs - 1], chunks=chunks)
    compare(b, ff=(False, False, False), excl=0,
            fp=dict(min_percent_identity_single=0.95, min_aligned_length_single=50), chunks=chunks)


def test_mask_matches_contig_names_mode():
    ref = synth.make_reference(30, 2_000_000, seed=3, min_len=1500, max_len=300_000)
    batch = synth.make_reads(ref, 30_000, seed=4)
    b = to_bamdata(batch, ref.lengths, ref.names)
    mask = (np.arange(30) % 3 != 0).astype(np.uint8)
    compare(b, ff=(True, False, False), excl=75, mask=mask)


def test_edge_cases_empty_and_tiny():
    ref_lens = np.asarray([100, 5, 149, 150, 151, 16384, 16385, 40000], dtype=np.int64)
    # no records at all
    empty = RecordBatch.from_arrays([], [], [], [], [], [], [], [0], [])
    with Session(0, FilterConfig(), 75, want_hist=True) as s:
        s.set_targets(ref_lens)
        st, summ = s.finish()
        assert summ.n_records == 0 and (st["n_pass"] == 0).all()
        s.push(empty)
        st, summ = s.finish()
        assert (st["win_sum_d"] == 0).all()
    # reads running off contig ends, contigs shorter than 2*excl, zero-length ops, deep pile at one base
    tid, pos, cig, coff = [], [], [], [0]

    def add(t, p, ops):
        tid.append(t); pos.append(p)
        cig.extend((l << 4) | "MIDNSHP=X".index(o) for l, o in ops)
        coff.append(len(cig))
    add(0, 0, [(100, "M")]); add(0, 50, [(80, "M")]); add(0, 99, [(1, "M"), (5, "S")])
    add(1, 0, [(5, "M")]); add(1, 4, [(3, "M")])
    add(2, 10, [(5, "S"), (20, "M"), (3, "I"), (0, "M"), (30, "M"), (10, "D"), (40, "=")])
    add(3, 0, [(150, "X")])
    for _ in range(3000):
        add(4, 75, [(1, "M")])
    add(5, 16383, [(1, "M")]); add(5, 16300, [(84, "M")][:1])
    add(6, 16380, [(2, "M"), (2, "N"), (1, "M")]); add(6, 16384, [(1, "M")])
    add(7, 100, [(30000, "M")]); add(7, 16000, [(500, "M"), (8000, "N"), (500, "M")]); add(7, 39999, [(1, "M")])
    n = len(tid)
    order = np.lexsort((pos, tid))
    tid = np.asarray(tid)[order]; posa = np.asarray(pos)[order]
    coff = np.asarray(coff); cig = np.asarray(cig, dtype=np.uint32)
    ncig = (coff[1:] - coff[:-1])[order]
    new_off = np.zeros(n + 1, dtype=np.uint32); np.cumsum(ncig, out=new_off[1:])
    new_cig = np.concatenate([cig[coff[i]:coff[i + 1]] for i in order])
    batch = RecordBatch.from_arrays(tid, posa, np.full(n, 99), np.full(n, 30), np.ones(n), np.ones(n),
                                    np.full(n, 150), new_off, new_cig)
    b = to_bamdata(batch, ref_lens)
    for excl in (0, 75, 8000):
        compare(b, ff=(True, True, False), excl=excl, check_depth=range(8))


@pytest.mark.parametrize("mode", ["parallel", "serial"])
def test_identity_sums_hot_contig_bit_exact(mode, monkeypatch):
    """ANIr sums over contigs with hundreds of thousands of reads: the exact-parallel path (integer sums per binade
    of the running total, verified against the exact sum) must reproduce the reference's serial f64 chain bit for bit."""
    monkeypatch.setenv("COVERM_IDENTITY", mode)
    ref = synth.make_reference(4, 3_000_000, seed=31, min_len=200_000, max_len=1_500_000)
    batch = synth.make_reads(ref, 900_000, seed=32)
    rng = np.random.default_rng(5)
    # make rounding ties and awkward values likely: NM = aligned/2, /4, 0, aligned (identity 0.5, 0.75, 1, 0)
    batch.nm = batch.nm.copy()
    sel = rng.random(batch.n_records) < 0.3
    batch.nm[sel] = rng.choice([0, 75, 150, 37, 1, 2, 3], size=int(sel.sum())).astype(np.uint32)
    b = to_bamdata(batch, ref.lengths, ref.names)
    st = compare(b, ff=(True, True, True), excl=75)
    assert st["n_pass"].max() > 150_000


def _long_read_batch(ref_lens, n_long, n_short, seed, huge_skip=False):
    """Long reads with hundreds of CIGAR operations (every op type), mixed with short reads; optional 2^24+ N skip."""
    rng = np.random.default_rng(seed)
    recs = []   # (tid, pos, [(len, op)])
    ops_mid = "MIDN=X"
    for _ in range(n_long):
        t = int(rng.integers(0, len(ref_lens)))
        L = int(ref_lens[t])
        n_ops = int(rng.choice([5, 40, 130, 400]))
        ops = [(int(rng.integers(1, 30)), "S")] if rng.random() < 0.3 else []
        ref_used = 0
        for k in range(n_ops):
            o = "M" if k % 2 == 0 else ops_mid[int(rng.integers(1, 6))]
            ln = int(rng.integers(1, 120)) if o in "M=X" else int(rng.integers(1, 12)) if o in "ID" else int(rng.integers(1, 3000))
            ops.append((ln, o))
            if o in "MDN=X": ref_used += ln
        if ref_used >= L - 1:
            continue
        recs.append((t, int(rng.integers(0, L - ref_used)), ops))
    for _ in range(n_short):
        t = int(rng.integers(0, len(ref_lens)))
        L = int(ref_lens[t])
        recs.append((t, int(rng.integers(0, L - 150)), [(150, "M")]))
    if huge_skip:   # a spliced-style record whose N operation alone exceeds 2^24 bases
        t = int(np.argmax(ref_lens))
        recs.append((t, 1000, [(100, "M"), ((1 << 24) + 12345, "N"), (80, "M"), (5, "D"), (20, "=")]))
    recs.sort(key=lambda r: (r[0], r[1]))
    tid = np.asarray([r[0] for r in recs]); pos = np.asarray([r[1] for r in recs])
    coff = np.zeros(len(recs) + 1, dtype=np.uint32)
    np.cumsum([len(r[2]) for r in recs], out=coff[1:])
    cig = np.asarray([(l << 4) | "MIDNSHP=X".index(o) for r in recs for l, o in r[2]], dtype=np.uint32)
    n = len(recs)
    flag = np.where(rng.random(n) < 0.05, 0x800 | 99, 99)
    return RecordBatch.from_arrays(tid, pos, flag, np.full(n, 40), rng.integers(0, 50, n), np.ones(n),
                                   np.full(n, 5000), coff, cig)


def test_long_reads_many_ops():
    """CIGARs beyond the fast walk's 128 operations, reads spanning tens of tiles, mixed with short reads."""
    ref_lens = np.asarray([300_000, 1_200_000, 80_000, 2_000_000], dtype=np.int64)
    batch = _long_read_batch(ref_lens, 1500, 20_000, seed=5)
    nops = np.diff(batch.cigar_off.astype(np.int64))
    assert (nops > 128).sum() > 100 and (nops > 3).sum() > 1000
    b = to_bamdata(batch, ref_lens)
    compare(b, ff=(True, True, False), excl=75, check_depth=range(4))
    compare(b, ff=(False, False, False), excl=0, fp=dict(min_percent_identity_single=0.5, min_aligned_length_single=1000))


def test_huge_skip_operation():
    """One operation of >= 2^24 bases forces the literal 64-bit walk for that record only."""
    ref_lens = np.asarray([50_000, 40_000_000, 70_000], dtype=np.int64)
    batch = _long_read_batch(ref_lens, 200, 5_000, seed=6, huge_skip=True)
    assert ((batch.cigar >> 4) >= (1 << 24)).sum() == 1
    b = to_bamdata(batch, ref_lens)
    compare(b, ff=(True, True, False), excl=75, check_depth=[0, 2])
    with Session(0, FilterConfig(), 0, want_hist=False) as s:   # depth of the large contig around the two blocks
        s.set_targets(ref_lens)
        s.push(batch)
        s.finish()
        d = s.depth(1)
    order, _ = O.reader_stage(b, None)
    ud = O.contig_deltas(b, O.FlagFilter(True, True, False), 1, order)
    np.testing.assert_array_equal(d, np.cumsum(ud, dtype=np.int64).astype(np.int32))


def test_bucket_regrowth_and_session_reuse():
    """One session, several samples: long reads (bucket buffer sized on a first pass and the pipeline repeated), then
    short reads, then a larger long-read sample (second growth); every result equals the oracle's, and the depth
    read-out after a repeated pass is exact."""
    ref_lens = np.asarray([400_000, 900_000, 150_000], dtype=np.int64)
    ref = synth.SynthReference(["c0", "c1", "c2"], ref_lens, np.zeros(3, np.int32), ["g"])
    samples = [_long_read_batch(ref_lens, 300, 2_000, seed=21), synth.make_reads(ref, 30_000, seed=22),
               _long_read_batch(ref_lens, 2_500, 10_000, seed=23)]
    off = O.FlagFilter(True, True, False)
    with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
        s.set_targets(ref_lens)
        for k, batch in enumerate(samples):
            b = to_bamdata(batch, ref_lens)
            exp, exp_hist, prim = O.integer_stats(b, off, None, 75, None)
            s.reset()
            s.push(batch)
            st, summ = s.finish()
            hist = s.hist()
            live = exp["seen"] == 1
            for f in ("n_pass", "sum_nm", "sum_indel", "win_sum_d", "win_sum_d2", "win_covered", "full_covered",
                      "win_min_d", "win_max_d", "hist_len"):
                np.testing.assert_array_equal(st[f][live], exp[f][live], err_msg="sample %d %s" % (k, f))
            for t in np.nonzero(live)[0]:
                n_b = int(exp["hist_len"][t])
                np.testing.assert_array_equal(hist[int(st["hist_off"][t]):int(st["hist_off"][t]) + n_b],
                                              exp_hist[int(exp["hist_off"][t]):int(exp["hist_off"][t]) + n_b])
            order, _ = O.reader_stage(b, None)
            np.testing.assert_array_equal(s.depth(1), np.cumsum(O.contig_deltas(b, off, 1, order), dtype=np.int64).astype(np.int32))


def _fuzz_case(seed):
    rng = np.random.default_rng(1000 + seed)
    n_ctg = int(rng.integers(1, 12))
    pool = [1, 5, 149, 150, 151, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 4096, 4097, 9000, 33000, 70000]
    ref_lens = np.asarray([pool[int(rng.integers(0, len(pool)))] for _ in range(n_ctg)], dtype=np.int64)
    n_rec = int(rng.integers(1, 3000))
    ops_all = "MIDNSHP=X"
    recs = []
    for _ in range(n_rec):
        t = int(rng.integers(0, n_ctg))
        L = int(ref_lens[t])
        style = rng.random()
        n_ops = 1 if style < 0.5 else int(rng.integers(2, 8)) if style < 0.9 else int(rng.integers(8, 40)) if style < 0.985 \
            else int(rng.integers(129, 200))
        ops, ref_used = [], 0
        budget = L - 1
        for k in range(n_ops):
            o = ops_all[int(rng.integers(0, 9))] if n_ops > 1 else "M"
            mx = 160 if n_ops < 8 else 12
            ln = int(rng.integers(0, mx)) if rng.random() < 0.97 else int(rng.integers(0, 3000))
            if o in "MDN=X":
                ln = min(ln, max(0, budget - ref_used))
                ref_used += ln
            ops.append((ln, o))
        pos = int(rng.integers(0, max(1, L - ref_used)))
        if rng.random() < 0.05 and ops[-1][1] in "M=X":          # run off the end of the contig: clipped, not an error
            ops[-1] = (ops[-1][0] + int(rng.integers(1, 400)), ops[-1][1])
            if pos + ref_used - 0 >= L: pos = max(0, L - ref_used - 1)
        recs.append([t, pos, ops])
    recs.sort(key=lambda r: (r[0], r[1]))
    if rng.random() < 0.3:                                          # positions out of order inside one contig: accepted
        t = recs[int(rng.integers(0, len(recs)))][0]
        idx = [i for i, r in enumerate(recs) if r[0] == t]
        perm = rng.permutation(len(idx))
        sub = [recs[idx[j]] for j in perm]
        for i, r in zip(idx, sub): recs[i] = r
    n = len(recs)
    tid = np.asarray([r[0] for r in recs]); pos = np.asarray([r[1] for r in recs])
    coff = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum([len(r[2]) for r in recs], out=coff[1:])
    cig = np.asarray([(l << 4) | ops_all.index(o) for r in recs for l, o in r[2]], dtype=np.uint32)
    base = rng.choice([99, 147, 83, 163, 97, 145, 0, 16], n)
    flag = base | np.where(rng.random(n) < 0.05, 0x100, 0) | np.where(rng.random(n) < 0.05, 0x800, 0) \
        | np.where(rng.random(n) < 0.04, 0x4, 0)
    nm = rng.integers(0, 30, n)
    batch = RecordBatch.from_arrays(tid, pos, flag, rng.integers(0, 61, n), nm, np.ones(n), rng.integers(1, 300, n), coff, cig)
    return ref_lens, batch, rng


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_small_cases(seed):
    """Random small inputs aimed at the index/bucket/walk machinery: contig lengths around tile boundaries, zero-length
    operations, every CIGAR op, CIGARs above the 16- and 128-operation thresholds, reads clipped by the contig end,
    position-unsorted contigs, all flag combinations, random filters, end exclusions and masks."""
    ref_lens, batch, rng = _fuzz_case(seed)
    b = to_bamdata(batch, ref_lens)
    ff = tuple(bool(x) for x in rng.integers(0, 2, 3))
    excl = int(rng.choice([0, 10, 75, 500]))
    mask = (rng.random(len(ref_lens)) < 0.8).astype(np.uint8) if rng.random() < 0.3 else None
    fp = None
    if rng.random() < 0.35 and mask is None:
        fp = dict(min_percent_identity_single=float(np.float32(rng.choice([0.0, 0.5, 0.9]))),
                  min_aligned_length_single=int(rng.choice([0, 20, 100])),
                  min_aligned_percent_single=float(np.float32(rng.choice([0.0, 0.3]))))
        if fp["min_percent_identity_single"] == 0.0 and fp["min_aligned_length_single"] == 0 and fp["min_aligned_percent_single"] == 0.0:
            fp = None
    depth_of = [t for t in range(min(3, len(ref_lens))) if mask is None or mask[t]]    # a masked contig has no depth here
    compare(b, ff=ff, fp=fp, excl=excl, mask=mask, check_depth=depth_of, chunks=int(rng.integers(1, 4)))


def test_stream_kernel_all_tiles(monkeypatch):
    """COVERM_PILEUP=stream: k_pileup_stream (round 1's default, now the slow-tile kernel behind k_pileup_fast) over every
    tile must agree with the oracle on the same inputs as the default pair of kernels."""
    monkeypatch.setenv("COVERM_PILEUP", "stream")
    for name in ["7seqs.reads_for_seq1_and_seq2.bam", "k141_2005182.bam", "eg2.bam"]:
        compare(load_fixture(name), ff=(True, True, False), excl=75)
    ref = synth.make_reference(40, 3_000_000, seed=11, min_len=1500, max_len=400_000)
    b = to_bamdata(synth.make_reads(ref, 60_000, seed=12), ref.lengths, ref.names)
    compare(b, ff=(True, True, False), excl=75, check_depth=[0, 1, 39], chunks=2)
    for seed in range(0, 48, 5):
        ref_lens, batch, rng = _fuzz_case(seed)
        compare(to_bamdata(batch, ref_lens), ff=(True, True, False), excl=int(rng.choice([0, 75])),
                check_depth=range(min(3, len(ref_lens))))


def test_fast_kernel_two_table_variant(monkeypatch):
    """COVERM_FAST_TABLES=2: k_pileup_fast2t (two u16 count tables, 384 LDS histogram bins) instead of the default k_pileup_fast (one table
    of biased deltas, 512 bins) — the same body, so the same inputs as the other kernels' tests, plus piles whose depths lie between the
    two bin counts."""
    monkeypatch.setenv("COVERM_FAST_TABLES", "2")
    for name in ["7seqs.reads_for_seq1_and_seq2.bam", "k141_2005182.bam", "eg2.bam"]:
        compare(load_fixture(name), ff=(True, True, False), excl=75)
    ref = synth.make_reference(40, 3_000_000, seed=11, min_len=1500, max_len=400_000)
    b = to_bamdata(synth.make_reads(ref, 60_000, seed=12), ref.lengths, ref.names)
    compare(b, ff=(True, True, False), excl=75, check_depth=[0, 1, 39], chunks=2)
    for seed in range(0, 48, 5):
        ref_lens, batch, rng = _fuzz_case(seed)
        compare(to_bamdata(batch, ref_lens), ff=(True, True, False), excl=int(rng.choice([0, 75])),
                check_depth=range(min(3, len(ref_lens))))
    _piles_between_the_bin_counts()


@pytest.mark.parametrize("kernel", ["lean", "7"])
def test_prep_kernel_compilations(kernel, monkeypatch):
    """The two implementations of k_prep — k_prep_lean + k_prep_generic (the default: a wave walks consecutive records, common steps in the
    loop, the others listed for the second launch) and k_prep7s (COVERM_PREP_KERNEL=7: round 5's body, every step per lane) — over every
    shape: flag filters, the reader-stage filter, a target mask, identity sums, long CIGARs, chunks that end inside a step."""
    if kernel == "7":
        monkeypatch.setenv("COVERM_PREP_KERNEL", kernel)
    else:
        monkeypatch.delenv("COVERM_PREP_KERNEL", raising=False)
    for name in ["7seqs.reads_for_seq1_and_seq2.bam", "2seqs.bad_read.1.with_supplementary.bam", "k141_2005182.bam", "eg2.bam"]:
        compare(load_fixture(name), ff=(True, True, False), excl=75)
        compare(load_fixture(name), ff=(False, True, True), excl=0)
    compare(load_fixture("eg2.bam"), ff=(True, True, True), excl=75,
            fp=dict(min_aligned_length_single=100, min_aligned_percent_single=0.9, min_percent_identity_single=0.95))
    ref = synth.make_reference(40, 3_000_000, seed=11, min_len=1500, max_len=400_000)
    b = to_bamdata(synth.make_reads(ref, 60_000, seed=12), ref.lengths, ref.names)
    compare(b, ff=(True, True, False), excl=75, check_depth=[0, 1, 39], chunks=2)
    compare(b, ff=(False, False, False), excl=0, fp=dict(min_percent_identity_single=0.95, min_aligned_length_single=50), chunks=3)
    mask = (np.arange(40) % 3 != 0).astype(np.uint8)
    compare(b, ff=(True, True, False), excl=75, mask=mask)
    ref_lens = np.asarray([300_000, 1_200_000, 80_000, 2_000_000], dtype=np.int64)
    compare(to_bamdata(_long_read_batch(ref_lens, 300, 5_000, seed=5), ref_lens), ff=(True, True, False), excl=75, check_depth=range(4))
    for seed in range(0, 48, 7):
        ref_lens, batch, rng = _fuzz_case(seed)
        compare(to_bamdata(batch, ref_lens), ff=(True, True, False), excl=int(rng.choice([0, 75])), check_depth=range(min(3, len(ref_lens))))


def test_assemblies_of_many_short_contigs():
    """The regime the reference is used in (its own fixtures carry 54 579 and 176 600 references, /root/reference src/contig.rs:513-520):
    (a) fewer than 128 records per contig — k_prep_generic walks every step, every step has several contig borders, the histogram layout
    spans many blocks of contigs; (b) ~250 records per contig — k_prep_lean's loop takes the steps inside a contig and lists the others."""
    ref = synth.make_reference(200_000, 260_000_000, seed=21, min_len=1000, max_len=40_000)      # 200 000 contigs
    b = to_bamdata(synth.make_reads(ref, 2_400_000, seed=22), ref.lengths, ref.names)
    compare(b, ff=(True, True, False), excl=75, check_depth=[0, 1, 199_999], chunks=2)
    compare(b, ff=(False, True, True), excl=0, fp=dict(min_percent_identity_single=0.95, min_aligned_length_single=50))
    ref = synth.make_reference(4_000, 30_000_000, seed=23, min_len=1000, max_len=60_000)
    b = to_bamdata(synth.make_reads(ref, 1_000_000, seed=24), ref.lengths, ref.names)
    compare(b, ff=(True, True, False), excl=75, check_depth=[0, 3_999])
    mask = (np.arange(4_000) % 3 != 0).astype(np.uint8)
    compare(b, ff=(True, True, False), excl=75, mask=mask)


def _piles_between_the_bin_counts():
    """Contigs whose depth climbs through 384 and 512: interior tiles leave the stripped loop (candidates >= bins) at different depths in
    the two variants, and the general loop's segments cross from the LDS bins into the arena."""
    rng = np.random.default_rng(78)
    ref_lens = np.asarray([30_000, 12_000], dtype=np.int64)
    n0, n1 = 75_000, 9_000
    tid = np.concatenate([np.zeros(n0, np.int32), np.ones(n1, np.int32)])
    # contig 0: density rising along the contig (depth 0 at the start to ~650 at the end); contig 1: flat, ~110
    u = np.sort(rng.random(n0))
    pos0 = (np.sqrt(u) * 29_000).astype(np.int32)
    pos1 = np.sort(rng.integers(0, 11_800, n1)).astype(np.int32)
    pos = np.concatenate([pos0, pos1])
    n = len(tid)
    cig = ((rng.integers(100, 150, n).astype(np.uint32)) << 4)
    batch = RecordBatch.from_arrays(tid, pos, np.zeros(n, np.uint16), np.full(n, 30, np.uint8), rng.integers(0, 4, n),
                                    np.ones(n, np.uint8), np.full(n, 150), np.arange(n + 1, dtype=np.uint32), cig)
    for excl in (0, 75):
        compare(to_bamdata(batch, ref_lens), ff=(True, True, False), excl=excl, check_depth=[0, 1])


def test_fast_kernel_piles_between_the_bin_counts():
    _piles_between_the_bin_counts()


def test_fast_kernel_deep_and_interleaved_tiles():
    """A pile deeper than k_pileup_fast's u16 count tables allow (> 32767 candidate runs in one tile: handed to
    k_pileup_stream through the slow-tile list) and one just below the limit, which stays on the fast kernel with depths
    far above its 512-bin LDS histogram (general loop, histogram overflow into the arena)."""
    rng = np.random.default_rng(77)
    ref_lens = np.asarray([40_000, 9_000, 25_000], dtype=np.int64)
    n0 = 40_000
    tid = np.concatenate([np.zeros(n0, np.int32), np.full(30_000, 1, np.int32), np.full(3_000, 2, np.int32)])
    pos = np.concatenate([np.sort(rng.integers(5_000, 5_400, n0)), np.sort(rng.integers(2_000, 2_300, 30_000)),
                          np.sort(rng.integers(0, 24_000, 3_000))]).astype(np.int32)
    n = len(tid)
    cig = ((rng.integers(60, 160, n).astype(np.uint32)) << 4)
    batch = RecordBatch.from_arrays(tid, pos, np.zeros(n, np.uint16), np.full(n, 30, np.uint8), rng.integers(0, 4, n),
                                    np.ones(n, np.uint8), np.full(n, 150), np.arange(n + 1, dtype=np.uint32), cig)
    for excl in (0, 75):
        compare(to_bamdata(batch, ref_lens), ff=(True, True, False), excl=excl, check_depth=[0, 1])


@pytest.mark.parametrize("neg_frac", [0.02, 0.6])
def test_identity_sums_with_negative_identities(neg_frac):
    """NM > aligned length gives a negative per-read identity (the reference adds it all the same, contig.rs:208-211).
    With a few of them the running sum dips and re-crosses binades; with most of them it goes negative, where the
    integer fast path must stand aside."""
    ref = synth.make_reference(3, 2_000_000, seed=41, min_len=300_000, max_len=1_000_000)
    batch = synth.make_reads(ref, 500_000, seed=42)
    rng = np.random.default_rng(6)
    batch.nm = batch.nm.copy()
    sel = (rng.random(batch.n_records) < neg_frac) & (batch.nm_kind == 1)
    batch.nm[sel] = rng.integers(151, 4000, int(sel.sum())).astype(np.uint32)
    b = to_bamdata(batch, ref.lengths, ref.names)
    st = compare(b, ff=(True, True, True), excl=0)
    assert (st["sum_identity_primary"] < 0).any() == (neg_frac > 0.5)


def test_single_identity_stream_flags():
    """COV_WANT_IDENTITY_PRIMARY_ONLY / _NONSUPP_ONLY: the requested sum is bit-identical to the two-stream run, the
    other one comes back 0."""
    ref = synth.make_reference(6, 2_000_000, seed=51, min_len=100_000, max_len=900_000)
    batch = synth.make_reads(ref, 300_000, seed=52)
    res = {}
    for mode in (True, "primary", "nonsupp"):
        with Session(0, FilterConfig(True, True, True), 75, want_hist=False, want_identity=mode) as s:
            s.set_targets(ref.lengths)
            s.push(batch)
            res[mode] = s.finish()[0]
    both = res[True]
    assert (both["sum_identity_primary"] != both["sum_identity_nonsupp"]).any()
    np.testing.assert_array_equal(res["primary"]["sum_identity_primary"].view(np.uint64), both["sum_identity_primary"].view(np.uint64))
    np.testing.assert_array_equal(res["nonsupp"]["sum_identity_nonsupp"].view(np.uint64), both["sum_identity_nonsupp"].view(np.uint64))
    assert (res["primary"]["sum_identity_nonsupp"] == 0).all() and (res["nonsupp"]["sum_identity_primary"] == 0).all()
    for f in ("n_pass", "win_sum_d", "win_sum_d2", "sum_nm"):
        np.testing.assert_array_equal(res["primary"][f], both[f])


def _device_tensors(batch, lo=0, hi=None):
    """cov_batch fields as torch device tensors; [lo, hi) is a VIEW into the full arrays, so cigar_off[0] != 0."""
    import torch
    hi = batch.n_records if hi is None else hi
    dev = torch.device("cuda", 0)
    full = {k: torch.from_numpy(np.ascontiguousarray(getattr(batch, k))).to(dev) for k in
            ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")}
    view = {k: (v[lo:hi] if k not in ("cigar_off", "cigar") else v) for k, v in full.items()}
    view["cigar_off"] = full["cigar_off"][lo:hi + 1]
    torch.cuda.synchronize()
    return view, hi - lo


def _finish_all(s):
    st, summ = s.finish()
    return st, s.hist(), int(summ.num_detected_primary_alignments)


def test_push_batch_device_matches_host_push():
    """cov_push_batch_device: adopted in place (zero copy, cigar offsets not starting at 0), then materialised when a
    second batch follows; short and long-read (bucket + repeated pass) inputs; all equal to pushing from the host."""
    ref = synth.make_reference(20, 4_000_000, seed=61, min_len=20_000, max_len=900_000)
    short = synth.make_reads(ref, 120_000, seed=62)
    ref_lens = np.asarray([300_000, 1_200_000, 80_000], dtype=np.int64)
    longb = _long_read_batch(ref_lens, 600, 4_000, seed=63)
    for batch, lens in ((short, ref.lengths), (longb, ref_lens)):
        n = batch.n_records
        cut = n // 3
        with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
            s.set_targets(lens)
            s.push(batch)
            want = _finish_all(s)
            want_depth = s.depth(1)
        # whole batch adopted
        t_all, n_all = _device_tensors(batch)
        with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
            s.set_targets(lens)
            s.push_device(t_all, n_all)
            got = _finish_all(s)
            assert got[0].tobytes() == want[0].tobytes() and (got[1] == want[1]).all() and got[2] == want[2]
            np.testing.assert_array_equal(s.depth(1), want_depth)
        # a view that starts in the middle of the CIGAR array, followed by a host batch -> materialised, offsets rebased
        t_tail, n_tail = _device_tensors(batch, cut, n)
        with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
            s.set_targets(lens)
            s.push(batch.slice(0, cut))
            s.push_device(t_tail, n_tail)
            got = _finish_all(s)
            assert got[0].tobytes() == want[0].tobytes() and (got[1] == want[1]).all() and got[2] == want[2]
        t_head, n_head = _device_tensors(batch, 0, cut)
        with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
            s.set_targets(lens)
            s.push_device(t_head, n_head)          # adopted ...
            s.push(batch.slice(cut, n))            # ... then copied into the owned store in front of this one
            got = _finish_all(s)
            assert got[0].tobytes() == want[0].tobytes() and (got[1] == want[1]).all() and got[2] == want[2]
        # adopted view alone: equals the host push of the same slice
        with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
            s.set_targets(lens)
            s.push(batch.slice(cut, n))
            w2 = _finish_all(s)
        with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
            s.set_targets(lens)
            s.push_device(t_tail, n_tail)
            g2 = _finish_all(s)
            # record indices (first/last_record) are relative to the pushed batch in both cases
            assert g2[0].tobytes() == w2[0].tobytes() and (g2[1] == w2[1]).all()
