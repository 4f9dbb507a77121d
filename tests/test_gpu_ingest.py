"""Device ingest (cov_ingest_*, csrc/ingest_kernels.hip.h): the GPU's inflate + BAM parse must put exactly the records into the
session's store that the CPU reader decodes, for every DEFLATE block type, across staging-piece boundaries, and must hand the
file back (IngestFallback) rather than guess when something is irregular."""
import os
import struct
import zlib

import numpy as np
import pytest

from coverm_amd import bam as cbam
from coverm_amd import synth
from coverm_amd.engine import FilterConfig, Session
from oracle import bamio
from tests.fixtures import load_fixture, swap_halves
from tests.knobs import set_knobs, with_knobs

pytestmark = pytest.mark.gpu
FIELDS = ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")


@pytest.fixture(autouse=True, params=["default", "lane-per-block", "lz-rounds"])
def inflate_version(request, monkeypatch):
    """Every test of this file runs with the kernels that ship (k_inflate_wave: one wave per BGZF block; k_lz_stage: the matches of a
    batch resolved in LDS), with the second inflate implementation (k_inflate: one lane per block) and with the second match
    resolution (k_lz_resolve: rounds through global memory).  cov_ingest_begin reads the switches, so sessions of one process may differ.
    (k_inflate_wave's builds with the 8-byte sink and with one unit per lock-step left the library in round 6; the CPU emulation of
    csrc/inflate_wave_core.h still runs both, tests/test_inflate_wave_core.py.)"""
    monkeypatch.delenv("COVERM_INFLATE_V", raising=False)
    monkeypatch.delenv("COVERM_LZ_V", raising=False)
    if request.param == "lane-per-block":
        monkeypatch.setenv("COVERM_INFLATE_V", "1")
    elif request.param == "lz-rounds":
        monkeypatch.setenv("COVERM_LZ_V", "1")
    return request.param


def binary_stderr(mode, paths, **kw):
    """stderr of a coverm-amd run with its timing lines on (they say which ingest path a file took)"""
    import subprocess
    from tests import binary
    r = subprocess.run(binary.argv(mode, paths, **kw), capture_output=True, text=True, timeout=300, env=dict(os.environ, COVERM_CLI_TIMING="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


def _check(path, threads=4, **kw):
    whole = cbam.read_alignment_file(path, threads=2, want_names=False)
    with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
        names, lens, n, timing = cbam.gpu_ingest(s, path, threads=threads, **kw)
        assert names == whole.ref_names
        np.testing.assert_array_equal(lens, whole.ref_lens)
        assert n == whole.records.n_records
        got = cbam.session_records(s)
        for f in FIELDS:
            np.testing.assert_array_equal(getattr(got, f), getattr(whole.records, f), err_msg=f)
        st_gpu, su_gpu = s.finish()
        h_gpu = s.hist()
    with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
        s.set_targets(whole.ref_lens)
        s.push(whole.records)
        st, su = s.finish()
        h = s.hist()
    assert st_gpu.tobytes() == st.tobytes() and (h_gpu == h).all()
    assert su_gpu.num_detected_primary_alignments == su.num_detected_primary_alignments
    return whole


@pytest.mark.parametrize("with_seq,piece_kb", [(2, 0), (2, 64), (1, 200), (0, 64)])
def test_device_ingest_equals_cpu_reader(tmp_path, monkeypatch, with_seq, piece_kb):
    """Product writer output (libdeflate / zlib level 1: dynamic-Huffman blocks), whole pieces and 64 KiB pieces (every BGZF
    block and many headers straddle a staging piece)."""
    if piece_kb:
        set_knobs(monkeypatch, ingest_piece_kb=piece_kb)
    ref = synth.make_reference(40, 6_000_000, seed=18, min_len=5000, max_len=800_000)
    b = synth.make_reads(ref, 120_000, seed=19)
    p = str(tmp_path / "s.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=with_seq, threads=4)
    w = _check(p)
    np.testing.assert_array_equal(w.records.pos, b.pos)


@pytest.mark.parametrize("level,block", [(0, 0xFF00), (9, 0xFF00), (6, 700), (1, 90)])
def test_device_inflate_block_types(tmp_path, level, block):
    """Stored blocks (level 0), long-match streams (level 9), and tiny BGZF blocks, which zlib emits with FIXED Huffman codes."""
    d = load_fixture("7seqs.reads_for_seq1_and_seq2.bam") if block < 1000 else load_fixture("eg2.bam")
    p = str(tmp_path / "x.bam")
    bamio.write_bam(p, d, level=level, block=block)
    _check(p)


def test_device_ingest_inflated_stream_is_bit_exact(tmp_path):
    ref = synth.make_reference(12, 900_000, seed=41, min_len=1500, max_len=200_000)
    b = synth.make_reads(ref, 30_000, seed=43)
    p = str(tmp_path / "s.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=2)
    raw = open(p, "rb").read()
    exp, q = b"", 0
    while q < len(raw):
        bs = int.from_bytes(raw[q + 16:q + 18], "little") + 1
        exp += zlib.decompress(raw[q + 18:q + bs - 8], -15)
        q += bs
    import ctypes as C
    from coverm_amd import native
    with Session(0, FilterConfig(), 75) as s:
        cbam.gpu_ingest(s, p, threads=2)
        L = native.lib()
        L.cov_ingest_copy_inflated.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        out = np.zeros(len(exp), np.uint8)
        assert L.cov_ingest_copy_inflated(s._h, 0, len(exp), out.ctypes.data) == 0
        assert out.tobytes() == exp


def test_device_ingest_hands_irregular_files_back(tmp_path):
    """A BGZF block whose CRC-32 does not match (beyond the part the host inflates for the header): IngestFallback and nothing
    appended; with the check switched off the (intact) payload goes through.  A CG:B,I long-CIGAR placeholder is NOT irregular: the
    device ingest resolves it like htslib does, and the binary's table over such a file equals the oracle's."""
    ref = synth.make_reference(12, 900_000, seed=41, min_len=1500, max_len=200_000)
    b = synth.make_reads(ref, 60_000, seed=43)
    good = str(tmp_path / "good.bam")
    cbam.write_bam(good, ref.names, ref.lengths, b, with_seq=2, threads=2)
    raw = bytearray(open(good, "rb").read())
    assert len(raw) > 3 << 20
    q = 0
    while q < (2 << 20) + (1 << 19):
        q += int.from_bytes(raw[q + 16:q + 18], "little") + 1
    bs = int.from_bytes(raw[q + 16:q + 18], "little") + 1
    raw[q + bs - 8] ^= 0x5A
    bad = str(tmp_path / "bad_crc.bam")
    open(bad, "wb").write(bytes(raw))
    with Session(0, FilterConfig(), 75) as s:
        with pytest.raises(cbam.IngestFallback):
            cbam.gpu_ingest(s, bad, threads=2)
        assert cbam.session_records(s).n_records == 0
        cbam.gpu_ingest(s, bad, threads=2, check_crc=False)
        assert cbam.session_records(s).n_records == b.n_records
    # long-CIGAR placeholder: `<l_seq>S<ref_len>N` + CG:B,I  (the CPU readers resolve it, tests/test_bam_reader.py)
    n_ops = 70_001
    ops = np.empty(n_ops, np.uint32)
    ops[0::2] = (3 << 4) | 0
    ops[1::2] = (1 << 4) | 2
    ref_span = int(((ops >> 4)[(ops & 15) != 1]).sum())
    l_seq = int(((ops >> 4)[(ops & 15) == 0]).sum())

    def rec(tid, pos, cigar, lseq, aux, name=b"q"):
        core = struct.pack("<iiBBHHHiiii", tid, pos, len(name) + 1, 30, 4680, len(cigar), 0, lseq, -1, -1, 0)
        body = core + name + b"\0" + struct.pack("<%dI" % len(cigar), *cigar) + b"\x11" * ((lseq + 1) // 2) + b"\xff" * lseq + aux
        return struct.pack("<i", len(body)) + body
    cg = b"CGBI" + struct.pack("<I", n_ops) + ops.tobytes()
    data = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 4) + b"big\0" + struct.pack("<i", ref_span + 1000)
    data += rec(0, 100, [(l_seq << 4) | 4, (ref_span << 4) | 3], l_seq, b"NMC\x05" + cg) + rec(0, 200, [(50 << 4) | 0], 50, b"NMC\x01", b"plain")
    out = b""
    for s0 in range(0, len(data), 0xff00):
        chunk = data[s0:s0 + 0xff00]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    p = str(tmp_path / "cg.bam")
    open(p, "wb").write(out)
    # ... and so does the device ingest: k_bam_hop counts the CG words, k_bam_extract copies them (contig.rs:166-168 sees htslib's swap)
    w = _check(p, threads=2)
    assert w.records.n_records == 2 and int(w.records.cigar_off[1]) == n_ops
    from oracle import oracle as O
    from tests import binary
    args = dict(methods=["mean", "covered_bases", "variance", "count"], contig_end_exclusion=0)
    ob = bamio.read_alignment_file(p)       # the oracle's reader keeps the placeholder: give it the CIGAR htslib would hand the reference
    placeholder = int(ob.cigar_off[1])
    ob.cigar = np.concatenate([ops, ob.cigar[placeholder:]]).astype(np.uint32)
    ob.cigar_off = np.concatenate([[0], ob.cigar_off[1:].astype(np.int64) - placeholder + n_ops]).astype(np.uint32)
    want = O.run_cli("contig", [p], bams=[ob], **args)
    assert binary.run("contig", [p], **args) == want
    assert "device ingest" in binary_stderr("contig", [p], **args)


@pytest.mark.parametrize("mode,round_blocks,carry_kb,cwin_kb,piece_kb", [("short", 64, 64, 0, 0), ("long", 64, 1024, 0, 0), ("short", 128, 4, 0, 0),
                                                                         ("carry_overflow", 64, 8, 0, 0), ("short", 256, 64, 512, 64),
                                                                         ("long", 4096, 1024, 300, 200)])
def test_device_ingest_in_many_windows(tmp_path, mode, round_blocks, carry_kb, cwin_kb, piece_kb):
    """The inflated stream exists one window (= one inflate round) at a time; records cut by a window's end are carried into
    the next.  With 64-block windows a 10 MB file takes ~10 windows (the three window buffers, four parse-state sets and both
    token buffers all come round several times); long reads make nearly every window end inside a record; a record larger than
    the carry buffer hands the file back.  cwin_kb bounds the compressed bytes of a round instead (rounds then close on bytes, not
    on block counts), with 64 KiB staging pieces so that blocks complete several feeds after their first bytes arrived."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = with_knobs(os.environ, ingest_round_blocks=round_blocks, ingest_carry_kb=carry_kb, ingest_cwin_kb=cwin_kb or None,
                     ingest_piece_kb=piece_kb or None)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "ingest_windows_worker.py"), mode, str(tmp_path)], capture_output=True, text=True,
                       env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "WINDOWS_OK " + mode in r.stdout


@pytest.mark.parametrize("count", [2, 3, 5])
def test_device_ingest_of_tid_spans_partitions_the_file(tmp_path, count):
    """covh_bam_gpu_ingest_span: each span reads only its part of the file (bisection for its end), the device drops the neighbours'
    records at both ends; the spans' records, in span order, are exactly the file's records, and equal the CPU span reader's."""
    ref = synth.make_reference(40, 6_000_000, seed=18, min_len=5000, max_len=800_000)
    b = synth.make_reads(ref, 150_000, seed=31)
    p = str(tmp_path / "s.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=4)
    whole = cbam.read_alignment_file(p, threads=2, want_names=False).records
    parts = []
    for i in range(count):
        with Session(0, FilterConfig(), 75) as s:
            names, lens, n, _ = cbam.gpu_ingest(s, p, threads=3, span=(i, count))
            got = cbam.session_records(s)
            assert got.n_records == n
            cpu = list(cbam.stream_batches(p, 2, span_index=i, span_count=count))[1:]
            assert sum(x.n_records for x in cpu) == n
            if n:
                np.testing.assert_array_equal(got.pos, np.concatenate([x.pos for x in cpu]))
            parts.append(got)
    assert sum(x.n_records for x in parts) == whole.n_records
    nonempty = [x for x in parts if x.n_records]
    assert len(nonempty) >= 2
    for f in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq"):
        np.testing.assert_array_equal(np.concatenate([getattr(x, f) for x in nonempty]), getattr(whole, f), err_msg=f)
    np.testing.assert_array_equal(np.concatenate([x.cigar for x in nonempty]), whole.cigar)


def test_device_ingest_given_up_midway_leaves_the_session_usable(tmp_path):
    """A BGZF block with an extra subfield besides BC, megabytes into the file: the driver stops feeding after uploads, inflate rounds
    and extractions are already queued (cov_ingest_abort waits for them), reports IngestFallback, and the same session then takes the
    CPU reader's records and produces what a fresh session produces (the queued work must not race the push)."""
    ref = synth.make_reference(12, 900_000, seed=41, min_len=1500, max_len=200_000)
    b = synth.make_reads(ref, 90_000, seed=47)
    good = str(tmp_path / "good.bam")
    cbam.write_bam(good, ref.names, ref.lengths, b, with_seq=2, threads=2)
    raw = open(good, "rb").read()
    q = 0
    while q < (3 << 20):
        q += int.from_bytes(raw[q + 16:q + 18], "little") + 1
    bs = int.from_bytes(raw[q + 16:q + 18], "little") + 1
    # same block with a second (empty) subfield "XX": XLEN 6 -> 10, BSIZE + 4
    blk = raw[q:q + 10] + struct.pack("<H", 10) + b"BC\x02\0" + struct.pack("<H", bs + 4 - 1) + b"XX\0\0" + raw[q + 18:q + bs]
    odd = str(tmp_path / "extra_subfield.bam")
    open(odd, "wb").write(raw[:q] + blk + raw[q + bs:])
    whole = cbam.read_alignment_file(odd, threads=2, want_names=False)
    assert whole.records.n_records == b.n_records
    with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
        with pytest.raises(cbam.IngestFallback) as ei:
            cbam.gpu_ingest(s, odd, threads=2)
        assert "subfield" in str(ei.value)
        assert cbam.session_records(s).n_records == 0
        s.push(whole.records)
        st, su = s.finish()
        h = s.hist()
        # and the session still ingests a regular file afterwards
        s.reset()
        cbam.gpu_ingest(s, good, threads=2)
        st3, su3 = s.finish()
        h3 = s.hist()
    with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
        s.set_targets(whole.ref_lens)
        s.push(whole.records)
        st2, su2 = s.finish()
        h2 = s.hist()
    assert st.tobytes() == st2.tobytes() == st3.tobytes() and (h == h2).all() and (h == h3).all()


def test_spans_of_a_file_that_is_not_sorted_by_reference_are_refused(tmp_path):
    """A span drops its neighbours' records trusting the file's order; a file whose tids decrease must end in the reference's
    "appears to be unsorted" error (contig.rs:129-132) in span mode too, from the device ingest and from the CPU span reader —
    never in silently missing records."""
    ref = synth.make_reference(40, 6_000_000, seed=18, min_len=5000, max_len=800_000)
    b = synth.make_reads(ref, 150_000, seed=31)
    cut = int(np.searchsorted(b.tid, 20))
    assert 0 < cut < b.n_records
    swapped = swap_halves(b, cut)
    p = str(tmp_path / "unsorted.bam")
    cbam.write_bam(p, ref.names, ref.lengths, swapped, with_seq=2, threads=4)
    seen_dev = seen_cpu = 0
    for i in range(2):
        with Session(0, FilterConfig(), 75) as s:
            try:
                cbam.gpu_ingest(s, p, threads=3, span=(i, 2))
            except IOError as e:
                assert "appears to be unsorted" in str(e)
                seen_dev += 1
        try:
            list(cbam.stream_batches(p, 2, span_index=i, span_count=2))
        except IOError as e:
            assert "appears to be unsorted" in str(e)
            seen_cpu += 1
    assert seen_dev >= 1 and seen_cpu >= 1


def test_spans_leave_a_file_only_the_reference_would_accept_to_one_device(tmp_path):
    """contig.rs:118-132 compares the tids of MAPPED records that passed the flag filters only: an unmapped record that carries a
    lower tid than its predecessor does not make the file unsorted for the reference.  The span readers refuse any decrease (above), so
    `--devices` with fewer files than devices sends such a file through one device whole: same table as one device, as the oracle."""
    from oracle import oracle as O
    from tests import binary
    ref = synth.make_reference(30, 3_000_000, seed=21, min_len=5000, max_len=400_000)
    b = synth.make_reads(ref, 90_000, seed=22)
    k = int(np.searchsorted(b.tid, 15)) + 5
    assert 0 < k < b.n_records - 1 and b.tid[k] >= 15
    b.tid[k] = 2                                   # one record of an early contig in the middle of the file ...
    b.pos[k] = 10
    b.flag[k] = np.uint16(int(b.flag[k]) | 4)      # ... unmapped: skipped before the order check
    p = str(tmp_path / "stray_unmapped.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=4)
    args = dict(methods=["mean", "variance", "count"])
    want = O.run_cli("contig", [p], bams=[bamio.read_alignment_file(p)], **args)
    assert binary.run("contig", [p], **args) == want
    assert binary.run("contig", [p], devices="0,0", **args) == want


def test_four_feeders_through_staging_slots_and_through_the_mapped_file(tmp_path):
    """Four device ingests at once (four spans of one file on device 0: one GPU here, a functional check): through staging slots — the
    default for any number of feeders since round 6's measurement with eight (profiles/r06_eight_feeders_io.json: eight up-front
    registrations do not run beside one another), filled from a mapping of the file with non-temporal stores or by pread — and with the file mapped and its spans registered with the device once, up front
    (COVERM_INGEST_IO=mmap-upfront).  The table must be the one-device one either way."""
    from oracle import oracle as O
    from tests import binary
    ref = synth.make_reference(80, 8_000_000, seed=31, min_len=5000, max_len=400_000)
    b = synth.make_reads(ref, 300_000, seed=32)
    p = str(tmp_path / "four_feeders.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=4)
    args = dict(methods=["mean", "trimmed_mean", "variance", "count"])
    want = O.run_cli("contig", [p], bams=[bamio.read_alignment_file(p)], **args)
    r = binary.run_full("contig", [p], env={"COVERM_CLI_TIMING": "1", "COVERM_INGEST_IO": "mmap-upfront"}, devices="0,0,0,0", **args)
    assert r.stdout == want
    assert r.stderr.count("bytes from the mapped file (registered up front)") == 4
    r = binary.run_full("contig", [p], env={"COVERM_CLI_TIMING": "1"}, devices="0,0,0,0", **args)
    assert r.stdout == want and r.stderr.count("bytes from staging slots (copied from the mapping)") == 4
    r = binary.run_full("contig", [p], env={"COVERM_CLI_TIMING": "1"}, devices="0,0", **args)
    assert r.stdout == want and r.stderr.count("bytes from staging slots (copied from the mapping)") == 2
    r = binary.run_full("contig", [p], env={"COVERM_CLI_TIMING": "1", "COVERM_INGEST_IO": "pread"}, devices="0,0,0,0", **args)
    assert r.stdout == want and r.stderr.count("bytes from staging slots (pread)") == 4


RAW = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raw")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(RAW) if f.endswith(".bam")) if os.path.isdir(RAW) else [])
def test_reference_fixture_files_unmodified_through_the_device_ingest(name):
    """The reference's own BAM files as htslib wrote them (tests/golden/raw, byte for byte): the device's inflate + parse must
    deliver exactly the records of the decoded fixture (tests/golden/fixtures/*.npz, made from the same file by make_golden.py
    with the pure-Python reader), and the inflated stream must equal zlib's."""
    path = os.path.join(RAW, name)
    if name.endswith("unsorted.bam") or not os.path.exists(os.path.join(os.path.dirname(RAW), "fixtures", name + ".npz")):
        exp = bamio.read_bam(path)
    else:
        exp = load_fixture(name)
    raw = open(path, "rb").read()
    stream, q = b"", 0
    while q < len(raw):
        bs = int.from_bytes(raw[q + 16:q + 18], "little") + 1
        stream += zlib.decompress(raw[q + 18:q + bs - 8], -15)
        q += bs
    import ctypes as C
    from coverm_amd import native
    unsorted = bool((np.diff(np.where(np.asarray(exp.tid) < 0, 0x7fffffff, np.asarray(exp.tid)).astype(np.int64)) < 0).any())
    with Session(0, FilterConfig(), 75) as s:
        if unsorted:      # keys that decrease: the device pair filter may not take "same tid" for "same run of a reference", so with mates the file is handed back
            with pytest.raises(cbam.IngestFallback):
                cbam.gpu_ingest(s, path, threads=2, want_mates=True)
        names, lens, n, _ = cbam.gpu_ingest(s, path, threads=2, want_mates=not unsorted)
        assert names == list(exp.ref_names) and n == len(exp.tid)
        got = cbam.session_records(s)
        L = native.lib()
        L.cov_ingest_copy_inflated.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        out = np.zeros(len(stream), np.uint8)
        assert L.cov_ingest_copy_inflated(s._h, 0, len(stream), out.ctypes.data) == 0
        assert out.tobytes() == stream
    for f, g in (("tid", exp.tid), ("pos", exp.pos), ("flag", exp.flag), ("mapq", exp.mapq), ("nm", exp.nm), ("nm_kind", exp.nm_kind), ("l_seq", exp.l_seq)):
        np.testing.assert_array_equal(getattr(got, f), np.asarray(g).astype(getattr(got, f).dtype), err_msg=f)
    np.testing.assert_array_equal(got.cigar_off, exp.cigar_off)
    np.testing.assert_array_equal(got.cigar, exp.cigar)


def _reblock(raw: bytes, sizes, level=1, wrong_crc_at=None):
    """The inflated stream of the BGZF file `raw`, cut into blocks of the given inflated sizes (cycled), each deflated on its own; the
    CRC-32 of block number `wrong_crc_at` gets one bit flipped.  -> (file bytes, number of blocks)"""
    data, q = b"", 0
    while q < len(raw):
        bs = int.from_bytes(raw[q + 16:q + 18], "little") + 1
        data += zlib.decompress(raw[q + 18:q + bs - 8], -15)
        q += bs
    out, off, k = bytearray(), 0, 0
    while off < len(data):
        n = min(sizes[k % len(sizes)], len(data) - off)
        chunk = data[off:off + n]
        co = zlib.compressobj(level if n < 60_000 else max(level, 1), zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        if len(comp) + 26 > 65536:                      # (random bytes at the largest sizes: store them)
            co = zlib.compressobj(0, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
        crc = zlib.crc32(chunk) ^ (0x0400 if k == wrong_crc_at else 0)
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", crc, n)
        off += n
        k += 1
    out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    return bytes(out), k


RAGGED = [1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 25, 63, 64, 65, 503, 504, 505, 511, 512, 513, 519, 520, 1023, 1024, 1025, 1031, 4095, 4096, 4097,
          32767, 65279, 65280, 65281]


def test_crc_of_ragged_blocks(tmp_path):
    """k_crc32_wave (and, in the lane-per-block combination, k_crc32) over blocks of every length class of its scheme — below the bytewise
    limit of 16, around a word, around a row of 512 bytes, around two rows, around htslib's 0xff00 — starting at whatever alignment the
    lengths in front leave (the sizes are odd on purpose): every block's CRC is accepted, the records equal the CPU reader's, and a file
    with ONE wrong trailer per length class is handed back (bgzf.c:inflate_block behind bam_generator.rs:125-129 fails such a file)."""
    ref = synth.make_reference(12, 900_000, seed=51, min_len=1500, max_len=200_000)
    b = synth.make_reads(ref, 40_000, seed=53)
    good = str(tmp_path / "good.bam")
    cbam.write_bam(good, ref.names, ref.lengths, b, with_seq=2, threads=2)
    raw = open(good, "rb").read()
    body, n_blocks = _reblock(raw, RAGGED)
    assert n_blocks > 2 * len(RAGGED)
    p = str(tmp_path / "ragged.bam")
    open(p, "wb").write(body)
    w = _check(p, threads=2)
    assert w.records.n_records == b.n_records
    # one wrong trailer at a time, in the 26th cycle of the sizes (~2.3 MB into the file: the host inflates the first MiB itself, for the header)
    with Session(0, FilterConfig(), 75) as s:
        for j in (0, 3, 6, 8, 12, 14, 16, 19, 20, 23, 27, 29, 30, 31, 32):
            bad = str(tmp_path / "bad.bam")
            open(bad, "wb").write(_reblock(raw, RAGGED, wrong_crc_at=25 * len(RAGGED) + j)[0])
            with pytest.raises(cbam.IngestFallback):
                cbam.gpu_ingest(s, bad, threads=2)
            assert cbam.session_records(s).n_records == 0, RAGGED[j]
