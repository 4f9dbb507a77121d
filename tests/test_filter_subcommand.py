"""`coverm-amd filter` (csrc/host_cli.cpp run_filter -> covh_bam_filter_file) against the oracle's restatement of
ReferenceSortedBamFilter::read (oracle.reader_filter, pinned on filter.rs's own tests in tests/test_oracle_golden.py) on the reference's
fixture BAMs: the output must hold exactly the records the filter returns, in its order, byte for byte, under the input's header —
what `coverm filter` does with a bam::Writer (bin/coverm.rs:408-472).  The file goes through in windows (bounded memory): the output must
not depend on where they end.  Host code only: runs without a GPU."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from oracle import bamio, oracle as O
from tests.golden import cases
from tests.knobs import with_knobs

HERE = os.path.dirname(os.path.abspath(__file__))
RAW = os.path.join(HERE, "golden", "raw")
BIN = os.path.join(os.path.dirname(HERE), "coverm_amd", "coverm-amd")

pytestmark = pytest.mark.skipif(not os.path.exists(BIN), reason="coverm-amd not built")


def inflate_bam(path):
    """-> (header bytes, [record bytes]) of a BAM file, every BGZF block's CRC-32 checked"""
    buf = open(path, "rb").read()
    out, p = [], 0
    while p < len(buf):
        assert buf[p:p + 4] == b"\x1f\x8b\x08\x04"
        bsize = struct.unpack_from("<H", buf, p + 16)[0] + 1
        data = zlib.decompress(buf[p + 18:p + bsize - 8], -15)
        crc, isize = struct.unpack_from("<II", buf, p + bsize - 8)
        assert zlib.crc32(data) == crc and len(data) == isize
        out.append(data)
        p += bsize
    assert buf[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])     # the EOF marker
    u = b"".join(out)
    assert u[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", u, 4)[0]
    q = 8 + l_text
    n_ref = struct.unpack_from("<i", u, q)[0]
    q += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", u, q)[0]
        q += 4 + l_name + 4
    header, recs = u[:q], []
    while q < len(u):
        bs = struct.unpack_from("<i", u, q)[0]
        recs.append(u[q:q + 4 + bs])
        q += 4 + bs
    assert q == len(u)
    return header, recs


def flags_of(case, inverse):
    v = []
    imp, supp, sec = case["ff"]
    if not imp:
        v.append("--proper-pairs-only")
    if not supp:
        v.append("--exclude-supplementary")
    if sec:
        v.append("--include-secondary")
    ls, ps, cs = case["single"]
    lp, pp, cp = case["pair"]
    v += ["--min-read-aligned-length", str(ls), "--min-read-percent-identity", repr(ps), "--min-read-aligned-percent", repr(cs),
          "--min-read-aligned-length-pair", str(lp), "--min-read-percent-identity-pair", repr(pp), "--min-read-aligned-percent-pair", repr(cp),
          "--min-mapq", str(case["mapq"])]
    if inverse:
        v.append("--inverse")
    return v


ALL = [(c, False) for c in cases.FILTER_CASES if c["bam"].endswith(".bam")] + [(c, True) for c in cases.FILTER_INVERSE_CASES] + \
      [(c, True) for c in cases.FILTER_CASES if c["bam"].endswith(".bam") and c["id"] != "filter_hello_world"]


@pytest.mark.parametrize("case,inverse", ALL, ids=[c["id"] + ("/inverse" if inv else "") for c, inv in ALL])
def test_filter_writes_the_records_the_reader_filter_returns(tmp_path, case, inverse):
    src = os.path.join(RAW, case["bam"])
    if not os.path.exists(src):
        pytest.skip("raw fixture not present")
    out = str(tmp_path / "out.bam")
    r = subprocess.run([BIN, "filter", "-b", src, "-o", out, "-t", "3"] + flags_of(case, inverse), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    b = bamio.read_alignment_file(src)
    fp = O.FilterParameters(O.FlagFilter(*case["ff"]), case["single"][0], float(np.float32(case["single"][1])), float(np.float32(case["single"][2])),
                            case["mapq"], case["pair"][0], float(np.float32(case["pair"][1])), float(np.float32(case["pair"][2])))
    order = [int(i) for i in O.reader_filter(b, fp, filter_out=not inverse)]
    h_in, r_in = inflate_bam(src)
    h_out, r_out = inflate_bam(out)
    assert h_out == h_in                                    # Header::from_template(reader.header()): the same names, lengths and text
    assert len(r_in) == b.n_records
    assert r_out == [r_in[i] for i in order]                # byte for byte: names, bases, qualities, tags
    # and the reference's own expectation where its test states one
    if not inverse or case in cases.FILTER_INVERSE_CASES:
        names = [b.qname[i].decode() for i in order]
        if "count" in case:
            assert len(names) == case["count"]
        elif case["exhaustive"]:
            assert names == case["qnames"]
        else:
            assert names[:len(case["qnames"])] == case["qnames"]


def test_filter_without_thresholds_and_its_errors(tmp_path):
    """tests/test_cmdline.rs:97-135: no thresholds = the pair branch with nothing to fail (`1<TAB>99<TAB>seq1` is there); with
    --min-read-percent-identity-pair 0.99 --proper-pairs-only it is not.  One output per input; a missing input is the reference's message."""
    src = os.path.join(RAW, "2seqs.bad_read.1.bam")
    if not os.path.exists(src):
        pytest.skip("raw fixture not present")

    def view(path):      # qname, flag, reference name — the first three columns of `samtools view`
        b = bamio.read_alignment_file(path)
        return ["%s\t%d\t%s" % (b.qname[i].decode(), int(b.flag[i]), b.ref_names[int(b.tid[i])] if b.tid[i] >= 0 else "*") for i in range(b.n_records)]
    out = str(tmp_path / "all.bam")
    assert subprocess.run([BIN, "filter", "-b", src, "-o", out], capture_output=True).returncode == 0
    assert "1\t99\tseq1" in view(out)
    out2 = str(tmp_path / "strict.bam")
    assert subprocess.run([BIN, "filter", "--min-read-percent-identity-pair", "0.99", "-b", src, "-o", out2, "--proper-pairs-only"], capture_output=True).returncode == 0
    assert "1\t99\tseq1" not in view(out2) and len(view(out2)) > 0
    r = subprocess.run([BIN, "filter", "-b", src, src, "-o", out], capture_output=True, text=True)
    assert r.returncode != 0 and "The number of input BAM files must be the same as the number output" in r.stderr
    r = subprocess.run([BIN, "filter", "-b", str(tmp_path / "nope.bam"), "-o", out], capture_output=True, text=True)
    assert r.returncode != 0 and "Unable to find BAM file" in r.stderr


@pytest.mark.parametrize("opt,val", [("--min-mapq", "abc"), ("--min-mapq", "300"), ("--min-mapq", "-1"), ("--min-read-aligned-length", "12x"),
                                     ("--min-read-percent-identity", "abc"), ("--min-read-aligned-percent-pair", ""), ("--threads", "1.5")])
def test_numbers_that_are_not_numbers_end_the_run(tmp_path, opt, val):
    """clap's typed value parsers (cli.rs: u8 / u16 / u32 / f32 arguments) refuse such values; they are not read as 0."""
    src = os.path.join(RAW, "2seqs.bad_read.1.bam")
    if not os.path.exists(src):
        pytest.skip("raw fixture not present")
    out = str(tmp_path / "o.bam")
    r = subprocess.run([BIN, "filter", "-b", src, "-o", out, opt, val], capture_output=True, text=True)
    assert r.returncode != 0 and "invalid value '%s' for '%s'" % (val, opt) in r.stderr and not os.path.exists(out)
    r = subprocess.run([BIN, "contig", "-b", src, "-m", "mean", opt, val], capture_output=True, text=True)
    assert r.returncode != 0 and "invalid value '%s' for '%s'" % (val, opt) in r.stderr


@pytest.mark.parametrize("window_kb", [1, 7, 300])
def test_output_does_not_depend_on_the_windows(tmp_path, window_kb):
    """covh_bam_filter_file streams the file in windows of BGZF blocks (64 MiB by default): with windows of 1 / 7 / 300 KiB — the header,
    most blocks and many records cut by a window's end, first mates parked across dozens of windows — the output file is the same, byte
    for byte, in the single-read branch, the pair branch and their inverses."""
    from coverm_amd import bam as cbam, synth
    ref = synth.make_reference(12, 3_000_000, seed=5, min_len=2000, max_len=900_000)
    b = synth.make_reads(ref, 40_000, seed=6)
    src = str(tmp_path / "pairs.bam")
    cbam.write_bam(src, ref.names, ref.lengths, b, with_seq=3, threads=2)      # records 2k / 2k + 1 on one reference are mates
    srcs = [src] + [os.path.join(RAW, n) for n in ("2seqs.bad_read.1.bam", "7seqs.reads_for_seq1_and_seq2.bam") if os.path.exists(os.path.join(RAW, n))]
    modes = [["--min-read-percent-identity", "97", "--min-read-aligned-length", "50"],
             ["--min-read-percent-identity", "97", "--inverse"],
             ["--min-read-percent-identity-pair", "97", "--proper-pairs-only"],
             ["--min-read-aligned-length-pair", "250", "--min-mapq", "20", "--inverse"],
             []]
    for si, sp in enumerate(srcs):
        for mi, mode in enumerate(modes):
            want, got = str(tmp_path / ("w%d_%d.bam" % (si, mi))), str(tmp_path / ("g%d_%d.bam" % (si, mi)))
            env = with_knobs(os.environ, filter_window_kb=None)
            r = subprocess.run([BIN, "filter", "-b", sp, "-o", want, "-t", "2"] + mode, capture_output=True, text=True, env=env, timeout=120)
            assert r.returncode == 0, r.stderr[-1000:]
            r = subprocess.run([BIN, "filter", "-b", sp, "-o", got, "-t", "3"] + mode, capture_output=True, text=True,
                               env=with_knobs(env, filter_window_kb=window_kb), timeout=300)
            assert r.returncode == 0, r.stderr[-1000:]
            assert open(got, "rb").read() == open(want, "rb").read(), (sp, mode)
            if si == 0 and mi == 2:         # and the pair branch really selects pairs here
                h, recs = inflate_bam(got)
                assert len(recs) > 1000 and len(recs) % 2 == 0


def test_truncated_input_is_an_error_not_a_short_output(tmp_path):
    src = os.path.join(RAW, "2seqs.bad_read.1.bam")
    if not os.path.exists(src):
        pytest.skip("raw fixture not present")
    raw = open(src, "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(raw[:len(raw) * 2 // 3])
    r = subprocess.run([BIN, "filter", "-b", cut, "-o", str(tmp_path / "o.bam")], capture_output=True, text=True)
    assert r.returncode != 0 and ("truncated" in r.stderr or "BGZF" in r.stderr)


def _adversarial(seed):
    """Records built to walk every branch of ReferenceSortedBamFilter::read: few read names (so names repeat three and more times on one
    reference), mates pointing at other references, references that come back later in the file, unmapped / secondary / supplementary /
    improper flags everywhere, short and long alignments, mapq 255, NM from 0 to more than the aligned length."""
    rng = np.random.default_rng(seed)
    n_ref = int(rng.integers(1, 5))
    ref_lens = rng.integers(500, 5000, n_ref).astype(np.int64)
    n = int(rng.integers(1, 400))
    runs = rng.integers(0, n_ref, max(1, n // int(rng.integers(3, 40)) + 1))          # a reference may come back: the parked set is cleared each time
    tid = np.repeat(runs, rng.integers(1, 60, len(runs)))[:n].astype(np.int32)
    n = len(tid)
    flag = np.zeros(n, np.uint16)
    flag |= np.where(rng.random(n) < 0.75, 0x2, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.06, 0x4, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.05, 0x100, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.05, 0x800, 0).astype(np.uint16)
    tid = np.where((flag & 0x4) != 0, np.where(rng.random(n) < 0.5, -1, tid), tid).astype(np.int32)
    mtid = np.where(rng.random(n) < 0.85, tid, rng.integers(-1, n_ref, n)).astype(np.int32)
    names = [b"q%d" % k for k in rng.integers(0, max(2, n // int(rng.integers(2, 5))), n)]
    l_seq = rng.integers(1, 200, n).astype(np.int32)
    ops, off = [], [0]
    for i in range(n):
        k = int(rng.integers(0, 5))
        for _ in range(k):
            ops.append((int(rng.integers(1, 120)) << 4) | int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 7, 8])))
        off.append(len(ops))
    mapq = rng.choice([0, 3, 20, 40, 60, 255], n).astype(np.uint8)
    nm = rng.integers(0, 30, n).astype(np.uint32)
    pos = np.zeros(n, np.int32)
    z = np.zeros(n, np.int32)
    return bamio.BamData(["r%d" % k for k in range(n_ref)], ref_lens, tid, pos, flag, mapq, l_seq, nm, np.ones(n, np.uint8),
                         np.asarray(off, np.uint32), np.asarray(ops, np.uint32), mtid, z, z, names, "")


@pytest.mark.parametrize("seed", range(24))
def test_random_adversarial_files_against_the_oracle(tmp_path, seed):
    d = _adversarial(seed)
    src = str(tmp_path / "in.bam")
    bamio.write_bam(src, d, level=1)
    b = bamio.read_alignment_file(src)
    h_in, r_in = inflate_bam(src)
    rng = np.random.default_rng(1000 + seed)
    for trial in range(4):
        pair = trial % 2 == 1
        inverse = trial >= 2
        ff = (bool(rng.random() < 0.5), bool(rng.random() < 0.7), bool(rng.random() < 0.3))
        case = dict(ff=ff, single=(int(rng.choice([0, 30, 80])), float(rng.choice([0.0, 0.9])), float(rng.choice([0.0, 0.5]))) if not pair else (0, 0.0, 0.0),
                    pair=(int(rng.choice([0, 60, 150])), float(rng.choice([0.0, 0.9])), float(rng.choice([0.0, 0.5]))) if pair else (0, 0.0, 0.0),
                    mapq=int(rng.choice([255, 255, 10, 30])))
        if pair and case["pair"] == (0, 0.0, 0.0) and case["mapq"] == 255:
            case["pair"] = (60, 0.0, 0.0)
        if not pair and case["single"] == (0, 0.0, 0.0):
            case["single"] = (30, 0.0, 0.0)
        out = str(tmp_path / ("o%d.bam" % trial))
        r = subprocess.run([BIN, "filter", "-b", src, "-o", out, "-t", "2"] + flags_of(case, inverse), capture_output=True, text=True, timeout=60,
                           env=with_knobs(os.environ, filter_window_kb=int(rng.choice([1, 64, 65536]))))
        assert r.returncode == 0, r.stderr[-1000:]
        fp = O.FilterParameters(O.FlagFilter(*ff), case["single"][0], float(np.float32(case["single"][1])), float(np.float32(case["single"][2])),
                                case["mapq"], case["pair"][0], float(np.float32(case["pair"][1])), float(np.float32(case["pair"][2])))
        order = [int(i) for i in O.reader_filter(b, fp, filter_out=not inverse)]
        h_out, r_out = inflate_bam(out)
        assert h_out == h_in
        assert r_out == [r_in[i] for i in order], (seed, trial, case, inverse)
