"""The standalone C++ CLI (coverm_amd/coverm-amd): BAM file on disk -> C++ reader -> C ABI -> kernels ->
C++ host layer -> stdout, compared with the reference's own CLI expectations (tests/test_cmdline.rs)."""
import os
import subprocess

import pytest

from oracle import bamio
from tests.fixtures import FIXDIR, load_fixture
from tests.golden import cases
from tests.knobs import with_knobs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "coverm_amd", "coverm-amd")


def argv_of(case, paths):
    a = case["args"]
    v = [BIN, case["mode"], "-b"] + paths
    if "methods" in a: v += ["-m"] + list(a["methods"])
    if "output_format" in a: v += ["--output-format", a["output_format"]]
    if "separator" in a: v += ["-s", a["separator"]]
    if a.get("single_genome"): v += ["--single-genome"]
    if "min_covered_fraction" in a: v += ["--min-covered-fraction", str(a["min_covered_fraction"])]
    if "min_mapq" in a: v += ["--min-mapq", str(a["min_mapq"])]
    if a.get("proper_pairs_only"): v += ["--proper-pairs-only"]
    if "genome_definition" in a: v += ["--genome-definition", os.path.join(FIXDIR, a["genome_definition"])]
    return v


def test_binary_exists_and_prints_usage():
    assert os.path.exists(BIN), "build with python -m coverm_amd.build"
    p = subprocess.run([BIN], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr


RAW = os.path.join(ROOT, "tests", "golden", "raw")


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["reference_file", "reencoded"])
@pytest.mark.parametrize("case", cases.CLI_CASES, ids=[c["id"] for c in cases.CLI_CASES])
def test_cli_binary_golden(case, tmp_path, source):
    """source = reference_file: the reference's own fixture BAMs byte for byte (tests/golden/raw: htslib's zlib-6 BGZF streams);
    reencoded: the decoded fixture written again in 3000-byte BGZF blocks (many blocks, records straddling them)."""
    paths = []
    for b in case["bams"]:
        stem = os.path.splitext(b)[0]
        p = str(tmp_path / (stem + ".bam"))     # the stoit name is the file stem
        raw = os.path.join(RAW, b)
        if source == "reference_file":
            if not os.path.exists(raw):
                pytest.skip("%s is not among the raw fixture files" % b)
            import shutil
            shutil.copy(raw, p)
        else:
            bamio.write_bam(p, load_fixture(b), block=3000)
        paths.append(p)
    r = subprocess.run(argv_of(case, paths), capture_output=True, text=True, timeout=300)
    if case["match"] == "error":
        assert r.returncode != 0 and case["expected"] in r.stderr
        return
    assert r.returncode == 0, r.stderr
    out = r.stdout
    if case["match"] == "is":
        assert out == case["expected"]
    elif case["match"] == "contains":
        assert case["expected"] in out
    elif case["match"] == "contains_all":
        for e in case["expected"]:
            assert e in out
    else:
        so = out.split("\n"); se = case["expected"].split("\n")
        assert [so[0]] + sorted(so[1:]) == [se[0]] + sorted(se[1:])
    if case["mode"] == "contig":
        assert "reads mapped out of" in r.stderr


@pytest.mark.gpu
def test_cli_binary_multi_sample_pipeline(tmp_path):
    """Config 4 shape (several BAMs -> one dense table) through the decoder-ahead pipeline and the reused session,
    including a filtered (single-read mode) pass; checked against the oracle's CLI text."""
    import numpy as np

    from coverm_amd import bam as cbam, synth
    from oracle import oracle as O
    from oracle.bamio import BamData
    ref = synth.make_reference(80, 5_000_000, seed=61, min_len=1500, max_len=300_000)
    paths, bs = [], []
    for k in range(4):
        batch = synth.make_reads(ref, 40_000 + 15_000 * k, seed=70 + k)
        p = str(tmp_path / ("sample%d.bam" % k))
        cbam.write_bam(p, ref.names, ref.lengths, batch, with_seq=(k % 2 == 0))
        paths.append(p)
        z = np.zeros(batch.n_records, np.int32)
        bs.append(BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq,
                          batch.l_seq.astype(np.int32), batch.nm, batch.nm_kind, batch.cigar_off, batch.cigar, z, z, z,
                          [], ""))
    r = subprocess.run([BIN, "contig", "-b"] + paths + ["-m", "mean", "variance", "rpkm", "-t", "8"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == O.run_cli("contig", paths, bams=bs, methods=["mean", "variance", "rpkm"])
    assert r.stderr.count("reads mapped out of") == 4
    r = subprocess.run([BIN, "contig", "-b"] + paths + ["-m", "trimmed_mean", "anir", "--min-read-percent-identity",
                                                         "97", "--min-read-aligned-length", "60", "-t", "8"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == O.run_cli("contig", paths, bams=bs, methods=["trimmed_mean", "anir"],
                                 min_read_percent_identity=97, min_read_aligned_length=60)
    # a missing file in the middle of the list is reported, not hung on
    r = subprocess.run([BIN, "contig", "-b", paths[0], str(tmp_path / "absent.bam"), paths[1], "-m", "mean"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "absent.bam" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.GENE_CLI_CASES, ids=[c["id"] for c in cases.GENE_CLI_CASES])
def test_cli_binary_per_gene_golden(case, tmp_path):
    """--gff through the standalone binary (C++ GFF parser + gene driver over cov_copy_depth); text == the oracle's."""
    from oracle import oracle as O
    paths = []
    for b in case["bams"]:
        p = str(tmp_path / (os.path.splitext(b)[0] + ".bam"))
        bamio.write_bam(p, load_fixture(b), block=3000)
        paths.append(p)
    a = dict(case["args"])
    v = [BIN, case["mode"], "-b"] + paths + ["--gff", os.path.join(FIXDIR, a["gff"]), "-m"] + list(a["methods"])
    if "output_format" in a: v += ["--output-format", a["output_format"]]
    if "contig_end_exclusion" in a: v += ["--contig-end-exclusion", str(a["contig_end_exclusion"])]
    if "min_covered_fraction" in a: v += ["--min-covered-fraction", str(a["min_covered_fraction"])]
    if a.get("no_zeros"): v += ["--no-zeros"]
    if "genome_definition" in a: v += ["--genome-definition", os.path.join(FIXDIR, a["genome_definition"])]
    r = subprocess.run(v, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for e in case["expected"]:
        assert e in r.stdout, r.stdout
    oa = dict(a)
    for k in ("gff", "genome_definition"):
        if k in oa: oa[k] = os.path.join(FIXDIR, oa[k])
    assert r.stdout == O.run_cli(case["mode"], case["bams"], bams=[load_fixture(b) for b in case["bams"]], **oa)


def test_binary_argument_errors():
    """Argument handling that needs no GPU: unknown flags, missing values, no BAM, methods that refuse a covered-fraction
    threshold (coverm.rs:1480-1503), metabat with --gff (coverm.rs:492-495)."""
    for argv, needle in ([["contig", "--frobnicate"], "unknown argument"],
                         [["contig", "-b"], "--bam-files is required"],
                         [["genome", "-m", "mean"], "--bam-files is required"],
                         [["contig", "-b", "x.bam", "--min-mapq"], "missing value"]):
        p = subprocess.run([BIN] + argv, capture_output=True, text=True)
        assert p.returncode != 0 and needle in p.stderr, (argv, p.stderr)
    p = subprocess.run([BIN, "contig", "-b", "/nonexistent/x.bam", "-m", "metabat", "--gff", "/nonexistent/g.gff"],
                       capture_output=True, text=True)
    assert p.returncode != 0 and "metabat method cannot be used with --gff" in p.stderr


@pytest.mark.gpu
def test_cli_binary_genome_definition_with_comments(tmp_path):
    """genome_parsing.rs:189-198: text after the contig name is ignored; the binary's own parser agrees with the plain file."""
    p = str(tmp_path / "7seqs.reads_for_seq1_and_seq2.bam")
    bamio.write_bam(p, load_fixture("7seqs.reads_for_seq1_and_seq2.bam"), block=3000)
    outs = []
    for d in ("7seqs.definition", "7seqs.definition_with_comments"):
        r = subprocess.run([BIN, "genome", "-b", p, "--genome-definition", os.path.join(FIXDIR, d)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1] and "genome2\t53.167923\n" in outs[0]
    bad = tmp_path / "bad.tsv"
    bad.write_text("g1\tgenome2~seq1\n\ng2\tgenome5~seq2\n")
    r = subprocess.run([BIN, "genome", "-b", p, "--genome-definition", str(bad)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "not a genome name and contig name separated by a tab" in r.stderr


def _bamdata(ref, batch):
    import numpy as np
    from oracle.bamio import BamData
    z = np.zeros(batch.n_records, np.int32)
    return BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq.astype(np.int32), batch.nm,
                   batch.nm_kind, batch.cigar_off, batch.cigar, z, z, z, [], "")


@pytest.mark.gpu
def test_cli_binary_streamed_equals_whole_file_and_oracle(tmp_path):
    """The streamed ingest (many small windows -> many pushes) gives the same table as --no-stream and as the oracle; all
    methods incl. histogram- and identity-based ones."""
    from coverm_amd import bam as cbam, synth
    from oracle import oracle as O
    ref = synth.make_reference(60, 6_000_000, seed=31, min_len=1500, max_len=500_000)
    batch = synth.make_reads(ref, 150_000, seed=32)
    p = str(tmp_path / "s.bam")
    cbam.write_bam(p, ref.names, ref.lengths, batch, with_seq=2)
    methods = ["mean", "trimmed_mean", "covered_fraction", "covered_bases", "variance", "length", "count", "reads_per_base", "rpkm", "tpm", "anir"]
    env = with_knobs(dict(os.environ, COVERM_CLI_TIMING="1", COVERM_NO_GPU_INGEST="1"), stream_window_kb=256)
    a = subprocess.run([BIN, "contig", "-b", p, "-t", "6", "-m"] + methods, capture_output=True, text=True, timeout=300, env=env)
    b = subprocess.run([BIN, "contig", "-b", p, "-t", "6", "--no-stream", "-m"] + methods, capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stderr
    assert b.returncode == 0, b.stderr
    assert "streamed" in a.stderr and "whole file" not in a.stderr
    assert a.stdout == b.stdout
    assert a.stdout == O.run_cli("contig", [p], bams=[_bamdata(ref, batch)], methods=methods)
    # default: device ingest (GPU inflate + parse), small staging pieces so that blocks straddle them
    env2 = with_knobs(dict(os.environ, COVERM_CLI_TIMING="1"), ingest_piece_kb=512)
    c = subprocess.run([BIN, "contig", "-b", p, "-t", "6", "-m"] + methods, capture_output=True, text=True, timeout=300, env=env2)
    assert c.returncode == 0, c.stderr
    assert "device ingest" in c.stderr and c.stdout == a.stdout
    # single-read filter path, streamed
    a = subprocess.run([BIN, "contig", "-b", p, "-t", "6", "-m", "mean", "variance", "--min-read-percent-identity", "98", "--min-read-aligned-length", "100"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert a.returncode == 0, a.stderr
    assert a.stdout == O.run_cli("contig", [p], bams=[_bamdata(ref, batch)], methods=["mean", "variance"], min_read_percent_identity=98,
                                 min_read_aligned_length=100)


@pytest.mark.gpu
def test_cli_binary_each_bam_uses_its_own_header(tmp_path):
    """contig.rs:29-32: every BAM is scanned against its own header.  Two BAMs with the same NUMBER of references but
    different names and lengths: the streaming taker (sparse output) reports each against its own header; the cached taker
    (dense output) exits with the reference's message (coverage_takers.rs:140-148)."""
    from coverm_amd import bam as cbam, synth
    from oracle import oracle as O
    refa = synth.make_reference(12, 900_000, seed=41, min_len=1500, max_len=200_000)
    refb = synth.make_reference(12, 1_400_000, seed=42, min_len=1500, max_len=300_000)
    refb.names = ["other_" + n for n in refb.names]
    ba, bb = synth.make_reads(refa, 20_000, seed=43), synth.make_reads(refb, 25_000, seed=44)
    pa, pb = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    cbam.write_bam(pa, refa.names, refa.lengths, ba, with_seq=1)
    cbam.write_bam(pb, refb.names, refb.lengths, bb, with_seq=0)
    r = subprocess.run([BIN, "contig", "-b", pa, pb, "-m", "mean", "covered_fraction", "--output-format", "sparse"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == O.run_cli("contig", [pa, pb], bams=[_bamdata(refa, ba), _bamdata(refb, bb)], methods=["mean", "covered_fraction"],
                                 output_format="sparse")
    assert "other_" in r.stdout
    r = subprocess.run([BIN, "contig", "-b", pa, pb, "-m", "mean"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "Found a difference amongst the reference sets used for mapping" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["contig", "genome"])
def test_cli_binary_span_sharded_devices(tmp_path, mode):
    """--devices with fewer BAMs than devices: the BAM is cut into tid spans, one session + stream reader per span, result
    blocks gathered on the first device (cov_gather).  Device 0 listed three times = three ranks on the one GPU of the test
    box (RCCL refuses that, so the blocks move by device copies; the span / merge logic is the same).  Output == one device."""
    from coverm_amd import bam as cbam, synth
    ref = synth.make_reference(90, 8_000_000, seed=51, min_len=1500, max_len=500_000)
    batch = synth.make_reads(ref, 200_000, seed=52)
    p = str(tmp_path / "s.bam")
    cbam.write_bam(p, ref.names, ref.lengths, batch, with_seq=2)
    extra = ["-m", "mean", "trimmed_mean", "variance", "count", "anir"] if mode == "contig" else ["-s", "~", "-m", "relative_abundance", "mean", "variance"]
    env = with_knobs(os.environ, stream_window_kb=512)
    one = subprocess.run([BIN, mode, "-b", p, "-t", "4"] + extra, capture_output=True, text=True, timeout=300, env=env)
    three = subprocess.run([BIN, mode, "-b", p, "-t", "6", "--devices", "0,0,0"] + extra, capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0, one.stderr
    assert three.returncode == 0, three.stderr
    assert three.stdout == one.stdout
    assert [l for l in three.stderr.splitlines() if "reads mapped out of" in l] == [l for l in one.stderr.splitlines() if "reads mapped out of" in l]


@pytest.mark.gpu
def test_cli_binary_rccl_gather_world_one_and_sample_parallel(tmp_path):
    """`--devices 0` with COVERM_FORCE_RCCL: the gather runs through librccl (ncclCommInitAll + ncclGather, world size 1) —
    the code path an 8-GPU node takes, minus the peers.  Several BAMs on a repeated device list = sample-parallel lanes."""
    from coverm_amd import bam as cbam, synth
    ref = synth.make_reference(30, 2_000_000, seed=61, min_len=1500, max_len=300_000)
    paths = []
    for k in range(3):
        b = synth.make_reads(ref, 30_000 + 5_000 * k, seed=62 + k)
        p = str(tmp_path / ("s%d.bam" % k))
        cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=1)
        paths.append(p)
    base = subprocess.run([BIN, "contig", "-b"] + paths + ["-m", "mean", "variance"], capture_output=True, text=True, timeout=300)
    assert base.returncode == 0, base.stderr
    lanes = subprocess.run([BIN, "contig", "-b"] + paths + ["-m", "mean", "variance", "--devices", "0,0", "-t", "4"], capture_output=True, text=True, timeout=300)
    assert lanes.returncode == 0, lanes.stderr
    assert lanes.stdout == base.stdout
    # world-size-1 RCCL gather through the span path (one BAM, device list of one, forced)
    import ctypes as C
    import numpy as np
    from coverm_amd import native
    from coverm_amd.engine import FilterConfig, Session
    L = native.lib()
    L.cov_gather.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32]
    L.cov_gathered.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(native.CovSummary)]
    b = synth.make_reads(ref, 50_000, seed=70)
    os.environ["COVERM_FORCE_RCCL"] = "1"
    try:
        with Session(0, FilterConfig(), 75, want_hist=True) as s:
            s.set_targets(ref.lengths)
            s.push(b)
            st, summ = s.finish()
            arr = (C.c_void_p * 1)(s._h)
            rc = L.cov_gather(arr, 1, 0)
            assert rc == 0, L.cov_last_error(s._h)
            st2 = np.zeros(len(ref.lengths), dtype=native.CONTIG_STATS_DTYPE)
            summ2 = native.CovSummary()
            assert L.cov_gathered(s._h, 0, st2.ctypes.data, C.byref(summ2)) == 0
            assert st2.tobytes() == st.tobytes()
            assert summ2.num_detected_primary_alignments == summ.num_detected_primary_alignments and summ2.n_records == b.n_records
    finally:
        del os.environ["COVERM_FORCE_RCCL"]
