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
