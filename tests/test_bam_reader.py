"""C++ BGZF/BAM/SAM reader (csrc/host_bam.cpp) against the oracle's pure-Python reader and writer."""
import glob
import os

import numpy as np
import pytest

from coverm_amd import bam as cbam
from coverm_amd import synth
from oracle import bamio
from tests.fixtures import load_fixture
from tests.golden import cases
from tests.knobs import set_knobs

REF_DATA = "/root/reference/tests/data"
FIELDS = ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "cigar_off", "cigar")


def same(af, d: bamio.BamData):
    assert af.ref_names == d.ref_names
    np.testing.assert_array_equal(af.ref_lens, d.ref_lens)
    for f in FIELDS:
        np.testing.assert_array_equal(getattr(af.records, f), getattr(d, f), err_msg=f)
    np.testing.assert_array_equal(af.records.l_seq, d.l_seq.astype(np.uint32))
    np.testing.assert_array_equal(af.mtid, d.mtid)
    if d.qname:
        assert af.qname == d.qname


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference fixtures only exist in the build container")
def test_reference_fixture_files_decode_identically():
    paths = sorted(glob.glob(REF_DATA + "/*.bam") + glob.glob(REF_DATA + "/*.sam"))
    assert len(paths) >= 20
    for p in paths:
        same(cbam.read_alignment_file(p, threads=4), bamio.read_alignment_file(p))


def synth_bamdata(n_reads, seed, n_contigs=12):
    ref = synth.make_reference(n_contigs, 400_000, seed=seed, min_len=1500, max_len=80_000)
    b = synth.make_reads(ref, n_reads, seed=seed + 1)
    n = b.n_records
    z = np.zeros(n, np.int32)
    return bamio.BamData(ref.names, ref.lengths, b.tid, b.pos, b.flag, b.mapq, b.l_seq.astype(np.int32), b.nm,
                         b.nm_kind, b.cigar_off, b.cigar, z - 1, z, z, [b"read%d" % i for i in range(n)], "")


@pytest.mark.parametrize("block,with_seq,nm_type,threads", [(0xFF00, True, "C", 1), (777, True, "S", 3),
                                                            (4096, False, "I", 8)])
def test_roundtrip_through_written_bam(tmp_path, block, with_seq, nm_type, threads):
    d = synth_bamdata(3000, seed=5)
    d.nm_kind = d.nm_kind.copy()
    d.nm_kind[7] = bamio.NM_BADTYPE      # written as type 'c' -> the reference would panic on it
    d.nm[d.nm_kind != bamio.NM_UNSIGNED] = 0
    p = str(tmp_path / "x.bam")
    bamio.write_bam(p, d, level=1, with_seq=with_seq, block=block, nm_type=nm_type)
    af = cbam.read_alignment_file(p, threads=threads)
    if not with_seq:
        d.l_seq = np.zeros_like(d.l_seq)
    same(af, d)
    same(af, bamio.read_bam(p))


def test_golden_fixtures_reencoded(tmp_path):
    """Every decoded reference fixture, re-encoded to BAM by the test writer, reads back identically."""
    for name in cases.FIXTURE_FILES:
        d = load_fixture(name)
        if d.n_records > 2000 or len(d.ref_names) > 1000:
            continue
        p = str(tmp_path / (name + ".re.bam"))
        bamio.write_bam(p, d, block=1500)
        same(cbam.read_alignment_file(p, threads=2), d)


def test_empty_and_errors(tmp_path):
    d = synth_bamdata(0, seed=9)
    p = str(tmp_path / "empty.bam")
    bamio.write_bam(p, d)
    af = cbam.read_alignment_file(p)
    assert af.records.n_records == 0 and len(af.ref_names) == 12
    with pytest.raises(IOError):
        cbam.read_alignment_file(str(tmp_path / "missing.bam"))
    bad = str(tmp_path / "bad.bam")
    with open(p, "rb") as fh:
        raw = bytearray(fh.read())
    raw[40] ^= 0xFF
    with open(bad, "wb") as fh:
        fh.write(raw)
    with pytest.raises(IOError):
        cbam.read_alignment_file(bad)


def test_cpp_writer_roundtrip(tmp_path):
    ref = synth.make_reference(9, 300_000, seed=2, min_len=1500, max_len=80_000)
    b = synth.make_reads(ref, 20_000, seed=3)
    b.nm = b.nm.copy(); b.nm[5] = 300; b.nm[6] = 70000    # exercise NM types S and I
    for with_seq in (True, False):
        p = str(tmp_path / ("w%d.bam" % with_seq))
        cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=with_seq, threads=4)
        d = bamio.read_bam(p)                      # oracle reader
        af = cbam.read_alignment_file(p, threads=4)  # product reader
        same(af, d)
        for f in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "cigar_off", "cigar"):
            np.testing.assert_array_equal(getattr(af.records, f), getattr(b, f), err_msg=f)
        np.testing.assert_array_equal(af.records.l_seq, b.l_seq if with_seq else np.zeros_like(b.l_seq))


def test_parallel_boundary_detection_matches_serial(tmp_path):
    """Bodies above 2 MiB take the speculative multi-segment record-boundary path; it must equal the serial hop."""
    ref = synth.make_reference(30, 5_000_000, seed=8, min_len=5000, max_len=800_000)
    b = synth.make_reads(ref, 120_000, seed=9)
    p = str(tmp_path / "big.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=True, threads=4)
    assert os.path.getsize(p) > 1 << 20
    one = cbam.read_alignment_file(p, threads=1)
    for thr in (2, 8, 16):
        many = cbam.read_alignment_file(p, threads=thr)
        for f in FIELDS + ("l_seq",):
            np.testing.assert_array_equal(getattr(many.records, f), getattr(one.records, f), err_msg=f)
        assert many.qname == one.qname
    np.testing.assert_array_equal(one.records.pos, b.pos)
    np.testing.assert_array_equal(one.records.cigar, b.cigar)


def test_malformed_inputs_are_errors_not_crashes(tmp_path):
    """Truncated BGZF stream, truncated record, wrong magic, empty and missing files: an IOError with a message."""
    from coverm_amd import bam as cbam
    b = load_fixture("7seqs.reads_for_seq1_and_seq2.bam")
    good = str(tmp_path / "good.bam")
    bamio.write_bam(good, b, block=700)
    raw = open(good, "rb").read()
    cases = {
        "missing": None,
        "empty": b"",
        "garbage": bytes(range(256)) * 8,
        "cut_block": raw[:len(raw) // 2],                  # ends inside a BGZF block
        "bad_crc": raw[:200] + bytes([raw[200] ^ 0xff]) + raw[201:],
    }
    for name, data in cases.items():
        p = str(tmp_path / (name + ".bam"))
        if data is not None:
            open(p, "wb").write(data)
        with pytest.raises(IOError) as ei:
            cbam.read_alignment_file(p, threads=3)
        assert str(ei.value), name
    af = cbam.read_alignment_file(good, threads=2)
    assert af.records.n_records == b.n_records


# ---------------------------------------------------------------------------------- streamed reader (covh_bam_stream_*)
def _same_records(a, b):
    for f in FIELDS + ("l_seq",):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)


@pytest.mark.parametrize("window_kb,threads,with_seq", [(64, 1, 1), (64, 4, 2), (300, 8, 2), (32768, 4, 0)])
def test_streamed_reader_equals_whole_file_reader(tmp_path, monkeypatch, window_kb, threads, with_seq):
    """Window by window (windows far smaller than the file, so records and BGZF blocks straddle every boundary) the
    streamed reader must hand out exactly the records covh_bam_open decodes, in order."""
    set_knobs(monkeypatch, stream_window_kb=window_kb)
    ref = synth.make_reference(30, 5_000_000, seed=8, min_len=5000, max_len=800_000)
    b = synth.make_reads(ref, 60_000, seed=9)
    p = str(tmp_path / "s.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=with_seq, threads=4)
    whole = cbam.read_alignment_file(p, threads=2, want_names=False)
    st = {}
    names, lens, rec = cbam.read_streamed(p, threads=threads, stats=st)
    assert names == whole.ref_names
    np.testing.assert_array_equal(lens, whole.ref_lens)
    _same_records(rec, whole.records)
    np.testing.assert_array_equal(rec.pos, b.pos)
    assert st["n_records"] == b.n_records
    if window_kb < 1000:
        assert st["peak_bytes"] < 64 << 20     # bounded: nowhere near the inflated size of the file


def test_streamed_reader_fixtures_and_big_header(tmp_path, monkeypatch):
    """Reference fixtures re-encoded (tiny BGZF blocks; eg2 has a 54 579-sequence header that spans many windows)."""
    set_knobs(monkeypatch, stream_window_kb=64)
    for name in ["7seqs.reads_for_seq1_and_seq2.bam", "k141_2005182.bam", "eg2.bam", "2seqs.reads_for_seq1.with_unmapped.bam"]:
        d = load_fixture(name)
        p = str(tmp_path / (name + ".re.bam"))
        bamio.write_bam(p, d, block=1500 if d.n_records < 2000 else 0xFF00)
        names, lens, rec = cbam.read_streamed(p, threads=3)
        assert names == d.ref_names
        for f in FIELDS:
            np.testing.assert_array_equal(getattr(rec, f), getattr(d, f), err_msg=name + ":" + f)


@pytest.mark.parametrize("spans", [2, 3, 8])
def test_streamed_spans_partition_the_file(tmp_path, monkeypatch, spans):
    """span k of n: every record in exactly one span, spans cut at tid changes, concatenation == the whole file; records
    without a reference (tid -1, at the end of a sorted BAM) go to the last span."""
    set_knobs(monkeypatch, stream_window_kb=128)
    ref = synth.make_reference(40, 6_000_000, seed=18, min_len=5000, max_len=800_000)
    b = synth.make_reads(ref, 80_000, seed=19)
    n_un = 500                                      # unplaced unmapped reads at the end
    import dataclasses
    b = dataclasses.replace(
        b, tid=np.concatenate([b.tid, np.full(n_un, -1, np.int32)]), pos=np.concatenate([b.pos, np.full(n_un, -1, np.int32)]),
        flag=np.concatenate([b.flag, np.full(n_un, 4, np.uint16)]), mapq=np.concatenate([b.mapq, np.zeros(n_un, np.uint8)]),
        nm=np.concatenate([b.nm, np.zeros(n_un, np.uint32)]), nm_kind=np.concatenate([b.nm_kind, np.zeros(n_un, np.uint8)]),
        l_seq=np.concatenate([b.l_seq, np.full(n_un, 150, np.uint32)]),
        cigar_off=np.concatenate([b.cigar_off, np.full(n_un, b.cigar_off[-1], np.uint32)]))
    p = str(tmp_path / "sp.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=4)
    parts = [cbam.read_streamed(p, threads=2, span_index=k, span_count=spans)[2] for k in range(spans)]
    assert sum(x.n_records for x in parts) == b.n_records
    np.testing.assert_array_equal(np.concatenate([x.tid for x in parts]), b.tid)
    np.testing.assert_array_equal(np.concatenate([x.pos for x in parts]), b.pos)
    np.testing.assert_array_equal(np.concatenate([x.cigar for x in parts]), b.cigar)
    tids = [set(np.unique(x.tid).tolist()) for x in parts]
    for i in range(spans):
        for j in range(i + 1, spans):
            assert not (tids[i] & tids[j]), "a contig is split between spans %d and %d" % (i, j)
    assert sum(1 for x in parts if x.n_records) >= 2
    assert -1 in tids[-1] or all(-1 not in t for t in tids)


def test_streamed_reader_errors(tmp_path, monkeypatch):
    set_knobs(monkeypatch, stream_window_kb=64)
    b = load_fixture("7seqs.reads_for_seq1_and_seq2.bam")
    good = str(tmp_path / "good.bam")
    bamio.write_bam(good, b, block=700)
    raw = open(good, "rb").read()
    for name, data in {"empty": b"", "garbage": bytes(range(256)) * 8, "cut_block": raw[:len(raw) // 2],
                       "bad_crc": raw[:200] + bytes([raw[200] ^ 0xff]) + raw[201:], "cut_record": None}.items():
        p = str(tmp_path / (name + ".bam"))
        if name == "cut_record":      # whole BGZF blocks, but the BAM stream ends inside a record
            d = bamio.read_bam(good)
            full = str(tmp_path / "full.bam")
            bamio.write_bam(full, d, block=300)
            r2 = open(full, "rb").read()
            # drop the last data block (keep the EOF marker)
            blocks, q = [], 0
            while q < len(r2):
                bs = int.from_bytes(r2[q + 16:q + 18], "little") + 1
                blocks.append(r2[q:q + bs]); q += bs
            data = b"".join(blocks[:-2]) + blocks[-1]
        open(p, "wb").write(data)
        with pytest.raises(IOError) as ei:
            cbam.read_streamed(p, threads=2)
        assert str(ei.value), name
    with pytest.raises(IOError):
        cbam.read_streamed(str(tmp_path / "missing.bam"))


def test_long_cigar_restored_from_cg_tag(tmp_path):
    """A CIGAR of more than 65535 operations is stored as `<l_seq>S<ref_len>N` + CG:B,I; htslib (and so the reference,
    contig.rs:168) sees the real CIGAR.  Both readers must resolve it."""
    n_ops = 70_001
    ops = np.empty(n_ops, np.uint32)
    ops[0::2] = (3 << 4) | 0          # 3M
    ops[1::2] = (1 << 4) | 2          # 1D
    ref_span = int(((ops >> 4)[(ops & 15) != 1]).sum())
    l_seq = int(((ops >> 4)[(ops & 15) == 0]).sum())
    names, lens = ["big", "other"], [ref_span + 1000, 5000]
    import struct
    import zlib
    def rec(tid, pos, cigar, lseq, aux, name=b"q"):
        core = struct.pack("<iiBBHHHiiii", tid, pos, len(name) + 1, 30, 4680, len(cigar), 0, lseq, -1, -1, 0)
        body = core + name + b"\0" + struct.pack("<%dI" % len(cigar), *cigar) + b"\x11" * ((lseq + 1) // 2) + b"\xff" * lseq + aux
        return struct.pack("<i", len(body)) + body
    cg = b"CGBI" + struct.pack("<I", n_ops) + ops.tobytes()
    placeholder = [(l_seq << 4) | 4, (ref_span << 4) | 3]
    r1 = rec(0, 100, placeholder, l_seq, b"NMC\x05" + cg)
    r2 = rec(0, 200, [(50 << 4) | 0], 50, b"NMC\x01", b"plain")
    r3 = rec(1, 10, [(20 << 4) | 4, (30 << 4) | 3], 20, b"NMC\x02", b"looks_like_it_but_no_CG")   # stays as stored
    text = b""
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 2)
    for n, l in zip(names, lens):
        hdr += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    data = hdr + r1 + r2 + r3
    out = b""
    for s0 in range(0, len(data), 0xff00):
        chunk = data[s0:s0 + 0xff00]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    p = str(tmp_path / "cg.bam")
    open(p, "wb").write(out)
    af = cbam.read_alignment_file(p, threads=2)
    _, _, st = cbam.read_streamed(p, threads=2)
    for r in (af.records, st):
        assert r.n_records == 3
        np.testing.assert_array_equal(r.cigar_off, [0, n_ops, n_ops + 1, n_ops + 3])
        np.testing.assert_array_equal(r.cigar[:n_ops], ops)
        np.testing.assert_array_equal(r.nm, [5, 1, 2])
        np.testing.assert_array_equal(r.nm_kind, [1, 1, 1])


def test_cpu_span_reader_refuses_a_file_that_is_not_sorted_by_reference(tmp_path):
    """A span drops its neighbours' records trusting the file's order: tids that decrease must end in the reference's error
    (contig.rs:129-132), not in silently missing records."""
    from coverm_amd import synth
    from tests.fixtures import swap_halves
    ref = synth.make_reference(40, 6_000_000, seed=18, min_len=5000, max_len=800_000)
    b = synth.make_reads(ref, 60_000, seed=31)
    cut = int(np.searchsorted(b.tid, 20))
    p = str(tmp_path / "unsorted.bam")
    cbam.write_bam(p, ref.names, ref.lengths, swap_halves(b, cut), with_seq=1, threads=2)
    seen = 0
    for i in range(2):
        try:
            list(cbam.stream_batches(p, 2, span_index=i, span_count=2))
        except IOError as e:
            assert "appears to be unsorted" in str(e)
            seen += 1
    assert seen >= 1
    # the whole-file reader does not judge the order (cov_finish does, in file order with the other per-record errors)
    assert sum(x.n_records for x in list(cbam.stream_batches(p, 2))[1:]) == b.n_records


def test_an_assemblys_reference_dictionary(tmp_path):
    """300 000 references with names of 1 to 40 characters: the header is ~7 MB inflated over ~110 BGZF blocks, so the header reader asks
    `is the dictionary complete?` several times (once per MiB of compressed bytes) and inflates a chunk's blocks on several threads;
    names and lengths must be the writer's (HeaderView::target_names / target_len, contig.rs:145), through covh_bam_read_header (the device
    ingest's header), the streamed reader and the whole-file reader."""
    import ctypes as C
    from coverm_amd import native
    rng = np.random.default_rng(17)
    n = 300_000
    alphabet = np.frombuffer(b"ACGTNacgtn_.|0123456789k", dtype=np.uint8)
    lens_name = rng.integers(1, 41, n)
    names = ["".join(map(chr, alphabet[rng.integers(0, len(alphabet), int(k))])) + "_%d" % i for i, k in enumerate(lens_name)]
    ref_lens = rng.integers(200, 50_000, n).astype(np.int64)
    batch = synth.make_reads(synth.make_reference(50, 500_000, seed=3, min_len=2000, max_len=40_000), 2_000, seed=4)
    batch.tid[:] = np.sort(rng.integers(0, n, batch.n_records)).astype(np.int32)
    batch.pos[:] = 0
    p = str(tmp_path / "assembly.bam")
    cbam.write_bam(p, names, ref_lens, batch, with_seq=0, threads=4)
    L = native.lib()
    L.covh_bam_read_header.restype = C.c_void_p
    L.covh_bam_read_header.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.covh_bam_header_free.argtypes = [C.c_void_p]
    L.covh_bam_header_n_targets.argtypes = [C.c_void_p]
    L.covh_bam_header_n_targets.restype = C.c_uint32
    L.covh_bam_header_target_name.argtypes = [C.c_void_p, C.c_uint32]
    L.covh_bam_header_target_name.restype = C.c_char_p
    L.covh_bam_header_target_len.argtypes = [C.c_void_p, C.c_uint32]
    L.covh_bam_header_target_len.restype = C.c_uint64
    err = C.create_string_buffer(512)
    hd = L.covh_bam_read_header(p.encode(), err, 512)
    assert hd, err.value
    try:
        assert L.covh_bam_header_n_targets(hd) == n
        for t in list(range(0, n, 9973)) + [n - 1]:
            assert L.covh_bam_header_target_name(hd, t).decode() == names[t]
            assert L.covh_bam_header_target_len(hd, t) == int(ref_lens[t])
    finally:
        L.covh_bam_header_free(hd)
    got_names, got_lens, rec = cbam.read_streamed(p, threads=3)
    assert got_names == names and np.array_equal(np.asarray(got_lens, np.int64), ref_lens) and rec.n_records == batch.n_records
    whole = cbam.read_alignment_file(p, threads=2, want_names=False)
    assert whole.ref_names == names and whole.records.n_records == batch.n_records
    # a dictionary cut short (the file ends inside it) is an error, not a shorter dictionary
    raw = open(p, "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(raw[:len(raw) // 3])
    assert not L.covh_bam_read_header(cut.encode(), err, 512)
