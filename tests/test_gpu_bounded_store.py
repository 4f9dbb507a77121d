"""Bounded record store (covermhip.h "bounded record store"): a sample larger than the store's cap goes through in pieces — the
complete contigs leave for the host, the contig in flight moves to the front — and every result equals the oracle's over the whole
sample.  The reference has no size limit because it holds one contig at a time and flushes on the tid change
(/root/reference src/contig.rs:128-155); the caps here are set far below the default (2^31) so that a 400 k-read sample spills
dozens of times."""
import ctypes as C
import os

import numpy as np
import pytest

from coverm_amd import bam as cbam
from coverm_amd import synth
from coverm_amd.engine import FilterConfig, RecordBatch, Session
from coverm_amd.native import CovError, ERR_NM_MISSING, ERR_UNSORTED
from oracle import oracle as O
from tests import binary
from tests.test_gpu_abi_parity import _long_read_batch, compare, to_bamdata, to_batch
from tests.knobs import set_knobs, with_knobs

pytestmark = pytest.mark.gpu

CAPS = dict(store_cap_records=50000, store_cap_cigar=200000)      # COVERM_KNOBS names (csrc/knobs.h)


@pytest.fixture
def small_caps(monkeypatch):
    set_knobs(monkeypatch, **CAPS)       # read by cov_create


def spills(s):
    L = s._lib
    L.cov_store_spills.restype = C.c_uint32
    L.cov_store_spills.argtypes = [C.c_void_p]
    return int(L.cov_store_spills(s._h))


def short_sample(n_reads=400_000, n_contigs=300, seed=71):
    ref = synth.make_reference(n_contigs, 30_000_000, seed=seed, min_len=1500, max_len=600_000)
    return ref, synth.make_reads(ref, n_reads, seed=seed + 1)


def test_push_path_short_reads_spill_and_equal_the_oracle(small_caps):
    ref, batch = short_sample()
    b = to_bamdata(batch, ref.lengths, ref.names)
    compare(b, excl=75, chunks=57)
    # the same through one session, to see that it really went through in pieces
    with Session(0, FilterConfig(), 75, want_hist=True) as s:
        s.set_targets(ref.lengths)
        for lo in range(0, batch.n_records, 9973):
            s.push(batch.slice(lo, min(batch.n_records, lo + 9973)))
        st, summ = s.finish()
        assert spills(s) >= 5
        assert int(summ.n_records) == batch.n_records
        with pytest.raises(CovError):
            s.depth(0)                   # needs every record: refused after a spill, not answered from a part
        s.reset()
        assert spills(s) == 0


def test_push_path_long_reads_spill_on_the_cigar_cap(small_caps):
    ref = synth.make_reference(40, 12_000_000, seed=75, min_len=200_000, max_len=900_000)
    batch = _long_read_batch(ref.lengths, 6_000, 30_000, seed=76)
    assert int(batch.cigar_off[-1]) > 3 * 200_000
    compare(to_bamdata(batch, ref.lengths), excl=0, chunks=23)
    compare(to_bamdata(batch, ref.lengths), ff=(True, False, True), excl=75, chunks=5)


def test_filter_and_mask_with_spills(small_caps):
    ref, batch = short_sample(200_000, 150, seed=81)
    b = to_bamdata(batch, ref.lengths, ref.names)
    fp = dict(min_aligned_length_single=50, min_percent_identity_single=0.95, min_aligned_percent_single=0.0, min_mapq=10,
              min_aligned_length_pair=0, min_percent_identity_pair=0.0, min_aligned_percent_pair=0.0)
    compare(b, ff=(True, True, False), fp=fp, excl=75, chunks=31)
    mask = (np.arange(len(ref.lengths)) % 3 != 0).astype(np.uint8)
    compare(b, excl=0, mask=mask, chunks=17)


def test_unconsidered_records_of_other_references_between_the_pieces(small_caps):
    """Unmapped mates carry their partner's reference — or any other: records the scan skips may sit anywhere, also where a piece
    ends.  Only the order of the CONSIDERED records matters (contig.rs:118-132)."""
    ref, batch = short_sample(250_000, 120, seed=85)
    rng = np.random.default_rng(5)
    tid = batch.tid.copy(); flag = batch.flag.copy()
    pick = rng.random(batch.n_records) < 0.03
    flag[pick] |= 0x4
    tid[pick] = rng.integers(-1, len(ref.lengths), int(pick.sum()))
    nb = RecordBatch.from_arrays(tid, batch.pos, flag, batch.mapq, batch.nm, batch.nm_kind, batch.l_seq, batch.cigar_off, batch.cigar)
    compare(to_bamdata(nb, ref.lengths, ref.names), excl=75, chunks=41)


def test_a_reference_that_comes_back_after_a_spill_is_unsorted(small_caps):
    ref, batch = short_sample(150_000, 60, seed=91)
    order = np.argsort(np.where(batch.tid == 3, 10_000, batch.tid), kind="stable")      # reference 3's records behind everything else
    b = to_bamdata(batch, ref.lengths, ref.names)
    moved = RecordBatch.from_arrays(b.tid[order], b.pos[order], b.flag[order], b.mapq[order], b.nm[order], b.nm_kind[order], b.l_seq[order],
                                    np.concatenate([[0], np.cumsum((batch.cigar_off[1:] - batch.cigar_off[:-1])[order])]).astype(np.uint32),
                                    np.concatenate([batch.cigar[batch.cigar_off[i]:batch.cigar_off[i + 1]] for i in order]))
    with Session(0, FilterConfig(), 75, want_hist=True) as s:
        s.set_targets(ref.lengths)
        with pytest.raises(CovError) as ei:
            for lo in range(0, moved.n_records, 20_000):
                s.push(moved.slice(lo, min(moved.n_records, lo + 20_000)))
            s.finish()
        assert ei.value.status == ERR_UNSORTED and spills(s) >= 1


def test_error_records_are_counted_from_the_first_record_of_the_sample(small_caps):
    ref, batch = short_sample(180_000, 80, seed=95)
    bad = 151_234
    while (batch.flag[bad] & 0x104) or batch.tid[bad] < 0:
        bad += 1
    nmk = batch.nm_kind.copy(); nmk[bad] = 0
    nb = RecordBatch.from_arrays(batch.tid, batch.pos, batch.flag, batch.mapq, batch.nm, nmk, batch.l_seq, batch.cigar_off, batch.cigar)
    with Session(0, FilterConfig(), 75) as s:
        s.set_targets(ref.lengths)
        with pytest.raises(CovError) as ei:
            for lo in range(0, nb.n_records, 15_000):
                s.push(nb.slice(lo, min(nb.n_records, lo + 15_000)))
            s.finish()
        assert ei.value.status == ERR_NM_MISSING and "record %d)" % bad in ei.value.message


def _bam(tmp_path, ref, batch, name):
    path = os.path.join(str(tmp_path), name + ".bam")
    cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=1, threads=8)
    return path


def test_device_ingest_through_the_binary_with_small_windows_and_caps(tmp_path):
    """caps at 50 k records / 200 k CIGAR words over a 400 k-read + long-read sample, table text == oracle."""
    ref, batch = short_sample(400_000, 300, seed=101)
    longs = synth.make_long_reads(ref, 1_200, seed=103, mean_len=8_000)
    both = merge_sorted(batch, longs)
    b = to_bamdata(both, ref.lengths, ref.names)
    path = _bam(tmp_path, ref, both, "bounded")
    env = with_knobs(dict(COVERM_CLI_TIMING="1"), ingest_round_blocks=128, **CAPS)
    args = dict(methods=["mean", "trimmed_mean", "covered_fraction", "covered_bases", "variance", "length", "count", "reads_per_base", "rpkm", "tpm", "anir"])
    want = O.run_cli("contig", [path], bams=[b], **args)
    r = binary.run_full("contig", [path], env=env, **args)
    assert r.stdout == want
    assert "bounded store: spill" in r.stderr
    assert binary.run("contig", [path], **args) == want                       # and without the caps
    # the CPU reader's pushes (no device ingest) and two tid spans meeting in cov_gather, each spilling on its own
    assert binary.run("contig", [path], env=with_knobs(dict(COVERM_NO_GPU_INGEST="1"), **CAPS), **args) == want
    assert binary.run("contig", [path], env=env, devices="0,0", **args) == want
    args = dict(methods=["relative_abundance", "mean", "variance"], separator="~")
    assert binary.run("genome", [path], env=env, **args) == O.run_cli("genome", [path], bams=[b], **args)


def test_reset_in_the_middle_of_an_ingest_that_already_spilled(tmp_path, monkeypatch):
    """cov_reset on a session whose open device ingest has already spilled part of its file.  The abort it runs first
    answers COV_ERR_STATE ("cov_reset the session") — which IS this call: it must go on and clear the store, the spill and the merged state, and
    the session must then take a sample as if nothing had happened.  Drives the raw cov_ingest_* calls of include/covermhip.h (what
    covh_bam_gpu_ingest does) with the file's own BGZF block table."""
    import zlib
    set_knobs(monkeypatch, ingest_round_blocks=64, **CAPS)
    ref, batch = short_sample(220_000, 120, seed=141)
    path = _bam(tmp_path, ref, batch, "reset")
    raw = open(path, "rb").read()

    class Blk(C.Structure):
        _fields_ = [("in_off", C.c_uint64), ("out_off", C.c_uint64), ("in_len", C.c_uint32), ("isize", C.c_uint32), ("crc", C.c_uint32), ("pad", C.c_uint32)]
    blocks, ends, off, out = [], [], 0, 0
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04" and raw[off + 12:off + 16] == b"BC\x02\x00"      # our writer's 18-byte BGZF header
        bsize = int.from_bytes(raw[off + 16:off + 18], "little") + 1
        isize = int.from_bytes(raw[off + bsize - 4:off + bsize], "little")
        if isize:
            blocks.append((off + 18, out, bsize - 26, isize, int.from_bytes(raw[off + bsize - 8:off + bsize - 4], "little")))
            ends.append(off + bsize)
        out += isize
        off += bsize
    head = b""
    for b in blocks[:64]:      # the BAM header: magic, l_text, text, n_ref, references
        head += zlib.decompress(raw[b[0]:b[0] + b[2]], -15)
    l_text = int.from_bytes(head[4:8], "little"); p = 8 + l_text
    n_ref = int.from_bytes(head[p:p + 4], "little"); p += 4
    for _ in range(n_ref):
        p += 4 + int.from_bytes(head[p:p + 4], "little") + 4
    first_record = p

    with Session(0, FilterConfig(), 75, want_hist=True) as s:
        L = s._lib
        L.cov_ingest_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
        L.cov_ingest_slot_wait.argtypes = [C.c_void_p, C.c_int]
        L.cov_ingest_feed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32]
        s.set_targets(ref.lengths)
        assert L.cov_ingest_begin(s._h, len(raw), first_record, 1) == 0
        buf = C.create_string_buffer(raw, len(raw))
        step = 48                                     # blocks per piece: a window (64 blocks) closes every other piece
        fed, k = 0, 0
        for lo in range(0, len(blocks) * 3 // 4, step):      # three quarters of the file: the ingest stays open
            hi = min(lo + step, len(blocks))
            arr = (Blk * (hi - lo))(*[Blk(*b, 0) for b in blocks[lo:hi]])
            assert L.cov_ingest_slot_wait(s._h, k % 4) == 0
            assert L.cov_ingest_feed(s._h, k % 4, C.byref(buf, fed), fed, ends[hi - 1] - fed, arr, hi - lo) == 0, L.cov_last_error(s._h)
            fed = ends[hi - 1]; k += 1
        import time
        t0 = time.time()
        while spills(s) == 0 and time.time() - t0 < 20:      # the windows are verified and extracted as further calls drain them
            assert L.cov_ingest_slot_wait(s._h, 0) == 0
            lo = hi; hi = min(lo + 8, len(blocks))
            if lo >= hi:
                break
            arr = (Blk * (hi - lo))(*[Blk(*b, 0) for b in blocks[lo:hi]])
            assert L.cov_ingest_feed(s._h, 0, C.byref(buf, fed), fed, ends[hi - 1] - fed, arr, hi - lo) == 0
            fed = ends[hi - 1]
        assert spills(s) >= 1, "the ingest did not spill: the test needs smaller caps"
        s.reset()                                      # COV_OK (Session.reset raises otherwise)
        assert spills(s) == 0
        # the session is as good as new: the whole sample, pushed, equals the oracle
        s.push(batch)
        st, summ = s.finish()
        hist = s.hist()
        assert int(summ.n_records) == batch.n_records
    with Session(0, FilterConfig(), 75, want_hist=True) as s2:      # a fresh session on the same sample
        s2.set_targets(ref.lengths)
        s2.push(batch)
        st2, summ2 = s2.finish()
        assert st.tobytes() == st2.tobytes() and (hist == s2.hist()).all() and int(summ.n_considered) == int(summ2.n_considered)
    compare(to_bamdata(batch, ref.lengths, ref.names), excl=75, chunks=3)


def merge_sorted(a, b):
    """Two record batches merged into one coordinate-sorted batch."""
    tid = np.concatenate([a.tid, b.tid]); pos = np.concatenate([a.pos, b.pos])
    key = np.where(tid < 0, 1 << 40, tid.astype(np.int64) << 32) + pos
    order = np.argsort(key, kind="stable")
    na, nb = np.diff(a.cigar_off.astype(np.int64)), np.diff(b.cigar_off.astype(np.int64))
    nops = np.concatenate([na, nb])[order]
    starts = np.concatenate([a.cigar_off[:-1].astype(np.int64), b.cigar_off[:-1].astype(np.int64) + int(a.cigar_off[-1])])[order]
    cig_all = np.concatenate([a.cigar[:int(a.cigar_off[-1])], b.cigar[int(b.cigar_off[0]):int(b.cigar_off[-1])]])
    idx = np.repeat(starts - np.concatenate([[0], np.cumsum(nops)[:-1]]), nops) + np.arange(int(nops.sum()))
    cat = lambda f: np.concatenate([getattr(a, f), getattr(b, f)])[order]
    return RecordBatch.from_arrays(tid[order], pos[order], cat("flag"), cat("mapq"), cat("nm"), cat("nm_kind"), cat("l_seq"),
                                   np.concatenate([[0], np.cumsum(nops)]).astype(np.uint32), cig_all[idx])
