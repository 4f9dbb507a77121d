"""Per-gene coverage (--gff; src/genes.rs): product host driver (C++ covh_gene_coverage + tests/harness_cli.py) against the
reference's golden vectors and against the oracle on synthetic genes.

CPU variant: the contig depth arrays come from the oracle, so GFF parsing, gene resolution, per-gene statistics, read
assignment and printing are checked without a GPU.  GPU variant (-m gpu): depth from the HIP engine (cov_copy_depth)."""
import contextlib
import io
import os

import numpy as np
import pytest

from coverm_amd import host, synth
from tests import harness_cli as cli
from tests.harness_cli import AlignmentFile
from coverm_amd.engine import RecordBatch, make_config
from coverm_amd.host import CoverageEstimator as E
from oracle import oracle as O
from oracle.bamio import BamData
from tests.fixtures import FIXDIR, load_fixture
from tests.golden import cases


def af_of(b: BamData, path) -> AlignmentFile:
    rec = RecordBatch.from_arrays(b.tid, b.pos, b.flag, b.mapq, b.nm, b.nm_kind, b.l_seq, b.cigar_off, b.cigar)
    return AlignmentFile(path, b.ref_names, b.ref_lens, rec, b.qname if len(b.qname) else None, b.mtid)


def oracle_depth_provider(bam_of):
    """Depth provider for CPU tests: what cov_copy_depth would return, computed by the oracle."""
    @contextlib.contextmanager
    def provider(af, records, filt, device=0):
        b = bam_of(af)
        off = O.FlagFilter(filt.include_improper_pairs, filt.include_supplementary, filt.include_secondary)
        ofp = None
        if filt.filter_single:
            ofp = O.FilterParameters(off, filt.min_aligned_length, filt.min_percent_identity, filt.min_aligned_percent, filt.min_mapq)
        # `records` is what the scan sees: rebuild a BamData over exactly those records (pair mode has reordered them)
        z = np.zeros(records.n_records, np.int32)
        bb = BamData(b.ref_names, b.ref_lens, records.tid, records.pos, records.flag, records.mapq,
                     records.l_seq.astype(np.int32), records.nm, records.nm_kind, records.cigar_off, records.cigar, z, z, z, [], "")
        order, prim = O.reader_stage(bb, ofp)

        def depth_of(tid):
            return np.cumsum(O.contig_deltas(bb, off, tid, order), dtype=np.int64).astype(np.int32)
        yield depth_of, prim, make_config(0, filt, 0)
    return provider


def _est(spec):
    return E.new_estimator_mean(*spec[1:]) if spec[0] == "mean" else E.new_estimator_read_count()


def _run_api_case(case, provider):
    b = load_fixture(case["bam"])
    af = af_of(b, "tests/data/" + case["bam"])
    taker = host.CoverageTaker.new_single_float_coverage_streaming_coverage_printer()
    genes = host.Genes.from_list(case["genes"])
    fp = cli.FilterParameters(cli.FlagFilter(True, False, False))
    records, filt, prim = cli.reader_stage(af, fp)
    mode, g_of, names = 0, None, None
    if case["namer"] is not None:
        names = sorted(set(case["namer"].values()))
        g_of = np.asarray([names.index(case["namer"][n]) if n in case["namer"] else -1 for n in af.ref_names], np.int32)
        mode = 3
    with provider(af, records, filt, 0) as prov:
        depth_of, prim_dev, cfg = prov[:3]
        rm = host.gene_coverage(af.ref_names, af.ref_lens, genes, af.stoit_name, records, cfg, depth_of, prim_dev, taker,
                                [_est(case["est"])], case["print_zeros"], mode, "~", g_of, names,
                                prov[3] if len(prov) > 3 else None)
    assert taker.text() == case["expected"]
    return rm


def _run_cli_case(case, depth_provider):
    files = [af_of(load_fixture(b), "tests/data/" + b) for b in case["bams"]]
    args = dict(case["args"])
    for k in ("gff", "genome_definition"):
        if k in args:
            args[k] = os.path.join(FIXDIR, args[k])
    out = cli.run(case["mode"], files, depth_provider=depth_provider, **args)
    for e in case["expected"]:
        assert e in out, out
    # and the whole text equals the oracle's
    assert out == O.run_cli(case["mode"], case["bams"], bams=[load_fixture(b) for b in case["bams"]], **args)


FIXTURE_PROVIDER = oracle_depth_provider(lambda af: load_fixture(os.path.basename(af.path)))


def test_gff_parsing_cpp():
    assert host.Genes.read_gff(os.path.join(FIXDIR, "2seqs.gff")).as_list() == cases.GFF_PARSE_EXPECTED
    with pytest.raises(IOError):
        host.Genes.read_gff(os.path.join(FIXDIR, "absent.gff"))


def test_gff_parsing_odd_lines(tmp_path):
    """GTF-style attributes, auto ids, skipped lines (too few columns, bad coordinates, start 0, end < start), CRLF,
    feature filter, '+' prefixed numbers — C++ parser == oracle parser (genes.rs:42-161)."""
    p = tmp_path / "odd.gff"
    p.write_text("##gff-version 3\n"
                 "c1\tsrc\tCDS\t10\t90\t.\t+\t0\tgene_id \"g 1\"; transcript_id \"t1\"\n"
                 "c1\tsrc\tgene\t+5\t50\t.\t+\t.\tName=abc;ID=\n"
                 "c2\tsrc\tgene\t1\t20\t.\t+\t.\n"
                 "c2\tsrc\tgene\t0\t20\t.\t+\t.\tID=zero\n"
                 "c2\tsrc\tgene\t30\t20\t.\t+\t.\tID=rev\n"
                 "c2\tsrc\tgene\tx\t20\t.\t+\t.\tID=nan\n"
                 "c2\tsrc\tgene\t5\r\n"
                 "c3\tsrc\tgene\t7\t8\t.\t-\t.\tlocus_tag=LT_1;Parent=p \r\n"
                 "\n# comment\n"
                 "c4\tsrc\tgene\t1\t2\t.\t-\t.\tfoo=bar\n")
    for ft in (None, "gene", "CDS"):
        assert host.Genes.read_gff(str(p), ft).as_list() == O.read_gff(str(p), ft)
    got = host.Genes.read_gff(str(p)).as_list()
    assert ("g 1", "c1", 9, 90) in got and ("abc", "c1", 4, 50) in got and ("c2_gene_1", "c2", 0, 20) in got
    assert ("LT_1", "c3", 6, 8) in got and ("c4_gene_2", "c4", 0, 2) in got and len(got) == 5


@pytest.mark.parametrize("case", cases.GENE_API_CASES, ids=[c["id"] for c in cases.GENE_API_CASES])
def test_gene_api_golden_cpu(case):
    _run_api_case(case, FIXTURE_PROVIDER)


@pytest.mark.parametrize("case", cases.GENE_CLI_CASES, ids=[c["id"] for c in cases.GENE_CLI_CASES])
def test_gene_cli_golden_cpu(case):
    _run_cli_case(case, FIXTURE_PROVIDER)


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.GENE_API_CASES, ids=[c["id"] for c in cases.GENE_API_CASES])
def test_gene_api_golden_gpu(case):
    _run_api_case(case, cli.device_depth)


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.GENE_CLI_CASES, ids=[c["id"] for c in cases.GENE_CLI_CASES])
def test_gene_cli_golden_gpu(case):
    _run_cli_case(case, cli.device_depth)


def _synthetic(tmp_path, seed, n_reads=40_000):
    ref = synth.make_reference(25, 1_500_000, seed=seed, min_len=1500, max_len=200_000)
    batch = synth.make_reads(ref, n_reads, seed=seed + 1)
    z = np.zeros(batch.n_records, np.int32)
    b = BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq.astype(np.int32), batch.nm,
                batch.nm_kind, batch.cigar_off, batch.cigar, z, z, z, [], "")
    rng = np.random.default_rng(seed)
    lines = ["##gff-version 3"]
    for c, (name, L) in enumerate(zip(ref.names, ref.lengths)):
        if c % 7 == 3:
            continue                                           # contigs without genes
        for k in range(int(rng.integers(1, 12))):
            s = int(rng.integers(1, L))
            e = min(int(L) + (50 if rng.random() < 0.1 else 0), s + int(rng.integers(1, 3000)))   # some run past the end
            lines.append("%s\tsyn\t%s\t%d\t%d\t.\t+\t.\tID=%s_g%d" % (name, "gene" if k % 3 else "CDS", s, e, name.replace("~", "_"), k))
    lines.append("not_in_header\tsyn\tgene\t1\t100\t.\t+\t.\tID=stray")
    gff = tmp_path / "syn.gff"
    gff.write_text("\n".join(lines) + "\n")
    return ref, b, str(gff)


ALL_METHODS = ["mean", "trimmed_mean", "covered_fraction", "covered_bases", "variance", "length", "count", "reads_per_base",
               "anir", "rpkm", "tpm"]


def _synthetic_cases(tmp_path, depth_provider_of):
    ref, b, gff = _synthetic(tmp_path, 71)
    af = af_of(b, "data/syn.bam")
    prov = depth_provider_of(b)
    for kw in (dict(mode="contig", methods=ALL_METHODS, output_format="sparse"),
               dict(mode="contig", methods=["mean", "variance"], contig_end_exclusion=0, gff_feature_type="CDS", no_zeros=True),
               dict(mode="genome", methods=["mean", "covered_fraction", "count"], separator="~", min_covered_fraction=0),
               dict(mode="genome", methods=["relative_abundance", "tpm"], single_genome=True, output_format="sparse"),
               dict(mode="contig", methods=["mean", "anir"], min_read_percent_identity=97, min_read_aligned_length=60,
                    proper_pairs_only=True, output_format="sparse"),
               # per-gene depth histograms (PileupCounts through genes.rs like any other estimator, coverm.rs:1358,1438-1446)
               dict(mode="contig", methods=["coverage_histogram"]),
               dict(mode="genome", methods=["coverage_histogram"], separator="~", contig_end_exclusion=10, min_covered_fraction=0)):
        kw = dict(kw)
        mode = kw.pop("mode")
        got = cli.run(mode, [af], gff=gff, depth_provider=prov, **kw)
        assert got == O.run_cli(mode, ["data/syn.bam"], bams=[b], gff=gff, **kw), kw
        assert got.count("\n") > 20


def test_gene_synthetic_matches_oracle_cpu(tmp_path):
    _synthetic_cases(tmp_path, lambda b: oracle_depth_provider(lambda af: b))


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["device", "host"])
def test_gene_synthetic_matches_oracle_gpu(tmp_path, where, monkeypatch):
    """device: per-gene reductions by cov_interval_stats_compute over the depth kept in HBM; host: over cov_copy_depth."""
    if where == "host":
        monkeypatch.setenv("COVERM_GENES_ON_HOST", "1")
    _synthetic_cases(tmp_path, lambda b: cli.device_depth)


@pytest.mark.gpu
def test_interval_stats_abi(tmp_path):
    """cov_interval_stats_compute against numpy on the depth read back by cov_copy_depth: windows, empty windows,
    histograms (incl. bins beyond the LDS window), intervals touching contig ends; then a second finish + recompute."""
    import ctypes as C
    from coverm_amd import native
    from coverm_amd.engine import FilterConfig, Session
    ref = synth.make_reference(6, 600_000, seed=81, min_len=20_000, max_len=300_000)
    batch = synth.make_reads(ref, 60_000, seed=82)
    # a deep pile so that some depths exceed the 512 bins kept in LDS: the first 3000 records of contig 0 start together
    batch.pos = batch.pos.copy()
    batch.pos[:3000] = batch.pos[0]
    assert (batch.tid[:3000] == 0).all()
    rng = np.random.default_rng(9)

    class IV(C.Structure):
        _fields_ = [("tid", C.c_uint32), ("pad", C.c_uint32), ("start", C.c_uint64), ("end", C.c_uint64)]

    class ST(C.Structure):
        _fields_ = [("win_sum_d", C.c_uint64), ("win_sum_d2", C.c_uint64), ("win_covered", C.c_uint64), ("full_covered", C.c_uint64),
                    ("win_min_d", C.c_uint32), ("win_max_d", C.c_uint32), ("hist_len", C.c_uint32), ("pad", C.c_uint32),
                    ("hist_off", C.c_uint64)]
    L = native.lib()
    L.cov_interval_stats_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
    L.cov_fetch_interval_hist.argtypes = [C.c_void_p, C.c_void_p]
    with Session(0, FilterConfig(), 0, want_hist=False) as s:
        s.set_targets(ref.lengths)
        s.push(batch)
        for rnd in range(2):
            s.finish()
            depth = [s.depth(t) for t in range(len(ref.lengths))]
            ivs = []
            for t, Lc in enumerate(ref.lengths):
                ivs += [(t, 0, int(Lc)), (t, int(Lc) - 10, int(Lc)), (t, 0, 1), (t, max(0, int(batch.pos[0]) - 50), min(int(Lc), int(batch.pos[0]) + 400))]
                for _ in range(40):
                    a = int(rng.integers(0, Lc - 1)); b = min(int(Lc), a + int(rng.integers(1, 5000)))
                    ivs.append((t, a, b))
            arr = (IV * len(ivs))(*[IV(t, 0, a, b) for t, a, b in ivs])
            for excl in (0, 75, 3000):
                out = (ST * len(ivs))()
                tot = C.c_uint64(0)
                assert L.cov_interval_stats_compute(s._h, arr, len(ivs), excl, 1, out, C.byref(tot)) == 0
                hist = np.zeros(max(1, tot.value), np.uint64)
                assert L.cov_fetch_interval_hist(s._h, hist.ctypes.data) == 0
                for (t, a, b), o in zip(ivs, out):
                    d = depth[t][a:b].astype(np.int64)
                    assert o.full_covered == int((d > 0).sum())
                    if 2 * excl < b - a:
                        w = d[excl:len(d) - excl]
                        assert (o.win_sum_d, o.win_sum_d2, o.win_covered) == (int(w.sum()), int((w * w).sum()), int((w > 0).sum()))
                        assert (o.win_min_d, o.win_max_d, o.hist_len) == (int(w.min()), int(w.max()), int(w.max()) + 1)
                        np.testing.assert_array_equal(hist[o.hist_off:o.hist_off + o.hist_len], np.bincount(w, minlength=int(w.max()) + 1).astype(np.uint64))
                    else:
                        assert (o.win_sum_d, o.win_covered, o.hist_len) == (0, 0, 0)
            s.reset()
            s.push(batch)
    assert max(int(x.max()) for x in depth) > 1000


@pytest.mark.gpu
def test_gene_coverage_through_the_binary_takes_the_device_ingest(tmp_path):
    """`coverm-amd --gff` over a BAM file: the device inflates and parses the file, the records the gene driver needs come back from
    the session's store (no whole-file decode on the host) — text == the oracle's, and == the run that decodes on the host
    (COVERM_GENES_DECODE_ON_HOST=1).  240 k reads in several BGZF blocks per contig; single-read and pair-mode filters included."""
    import subprocess
    from coverm_amd import bam as cbam
    from tests import binary
    ref = synth.make_reference(25, 1_500_000, seed=83, min_len=1500, max_len=200_000)
    batch = synth.make_reads(ref, 240_000, seed=84)
    n = batch.n_records
    i = np.arange(n, dtype=np.int64)
    m = i ^ 1
    same = (m < n) & (batch.tid[np.minimum(m, n - 1)] == batch.tid)          # what the writer's with_seq = 3 makes mates of
    b = BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq.astype(np.int32), batch.nm, batch.nm_kind,
                batch.cigar_off, batch.cigar, batch.tid.copy(), np.zeros(n, np.int32), np.zeros(n, np.int32),
                [b"n%d" % k for k in np.where(same, i & ~1, i)], "")
    path = str(tmp_path / "syn.bam")
    cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=3, threads=4)
    rng = np.random.default_rng(5)
    lines = ["##gff-version 3"]
    for c, (name, L) in enumerate(zip(ref.names, ref.lengths)):
        for k in range(int(rng.integers(1, 9))):
            s = int(rng.integers(1, L))
            lines.append("%s\tsyn\tgene\t%d\t%d\t.\t+\t.\tID=%s_g%d" % (name, s, min(int(L), s + int(rng.integers(1, 3000))), name.replace("~", "_"), k))
    gff = tmp_path / "syn.gff"
    gff.write_text("\n".join(lines) + "\n")
    for kw in (dict(methods=["mean", "variance", "covered_fraction", "count", "anir"]),
               dict(methods=["mean", "trimmed_mean"], min_read_percent_identity=97, proper_pairs_only=True, output_format="sparse"),
               dict(methods=["mean", "count"], min_read_percent_identity_pair=95, proper_pairs_only=True)):
        want = O.run_cli("contig", [path], bams=[b], gff=str(gff), **kw)
        r = subprocess.run(binary.argv("contig", [path], gff=str(gff), **kw), capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1"), timeout=120)
        assert r.returncode == 0, r.stderr
        assert "--gff over the device ingest" in r.stderr
        assert r.stdout == want, kw
        assert binary.run("contig", [path], gff=str(gff), env={"COVERM_GENES_DECODE_ON_HOST": "1"}, **kw) == want
        assert want.count("\n") > 20
