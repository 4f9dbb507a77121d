"""Host layer (C++ estimators / scan drivers / takers / printers + tests/harness_cli.py) against the
reference's golden vectors.

CPU variant: the per-contig integer statistics are supplied by the oracle (so the host logic is
checked without a GPU).  GPU variant (-m gpu): the same cases with statistics from the HIP engine,
i.e. the full product path fixture -> C ABI -> kernels -> host layer -> exact reference text.
"""
import os

import numpy as np
import pytest

from coverm_amd import host, native
from tests import harness_cli as cli
from tests.harness_cli import AlignmentFile
from coverm_amd.engine import RecordBatch
from coverm_amd.host import CoverageEstimator as E
from coverm_amd.host import CoverageTaker, SampleResult
from oracle import oracle as O
from tests.fixtures import FIXDIR, load_fixture
from tests.golden import cases
from tests.knobs import set_knobs


def alignment_file(name) -> AlignmentFile:
    b = load_fixture(name)
    rec = RecordBatch.from_arrays(b.tid, b.pos, b.flag, b.mapq, b.nm, b.nm_kind, b.l_seq, b.cigar_off, b.cigar)
    return AlignmentFile("tests/data/" + name, b.ref_names, b.ref_lens, rec, b.qname, b.mtid)


def oracle_sample(af, fp, contig_end_exclusion, want_hist, want_identity, mask=None, device=0) -> SampleResult:
    """Statistics provider for CPU tests: the oracle computes what the device would return."""
    b = load_fixture(os.path.basename(af.path))
    off = O.FlagFilter(fp.flag_filters.include_improper_pairs, fp.flag_filters.include_supplementary,
                       fp.flag_filters.include_secondary)
    ofp = O.FilterParameters(off, fp.min_aligned_length_single, fp.min_percent_identity_single,
                             fp.min_aligned_percent_single, fp.min_mapq, fp.min_aligned_length_pair,
                             fp.min_percent_identity_pair, fp.min_aligned_percent_pair)
    st, hist, prim = O.integer_stats(b, off, ofp if ofp.doing_filtering() else None, contig_end_exclusion, mask)
    out = np.zeros(len(st), dtype=native.CONTIG_STATS_DTYPE)
    for f in ("n_primary", "n_pass", "n_nonsupp", "sum_nm", "sum_indel", "win_sum_d", "win_sum_d2", "win_covered",
              "full_covered", "first_record", "last_record", "win_min_d", "win_max_d", "hist_len", "hist_off"):
        out[f] = st[f]
    out["sum_identity_primary"] = st["id_primary"]
    out["sum_identity_nonsupp"] = st["id_nonsupp"]
    return SampleResult(af.stoit_name, out, hist if want_hist else None, prim)


def make_est(spec):
    k = spec[0]
    return {"mean": lambda: E.new_estimator_mean(spec[1], spec[2], spec[3]),
            "variance": lambda: E.new_estimator_variance(spec[1], spec[2]),
            "trimmed_mean": lambda: E.new_estimator_trimmed_mean(spec[1], spec[2], spec[3], spec[4]),
            "pileup_counts": lambda: E.new_estimator_pileup_counts(spec[1], spec[2]),
            "covered_fraction": lambda: E.new_estimator_covered_fraction(spec[1]),
            "covered_bases": lambda: E.new_estimator_covered_bases(spec[1]),
            "rpkm": lambda: E.new_estimator_rpkm(spec[1]), "tpm": lambda: E.new_estimator_tpm(spec[1]),
            "length": E.new_estimator_length, "read_count": E.new_estimator_read_count,
            "reads_per_base": E.new_estimator_reads_per_base, "anir": E.new_estimator_anir}[k]()


def excl_of(specs):
    for s in specs:
        if s[0] in ("mean", "variance", "pileup_counts"):
            return s[2]
        if s[0] == "trimmed_mean":
            return s[4]
    return 0


def run_api_case(case, provider):
    files = [alignment_file(b) for b in case["bams"]]
    est = [make_est(e) for e in case["est"]]
    excl = excl_of(case["est"])
    want_hist, want_id = host.wants(est)
    taker = (CoverageTaker.new_single_float_coverage_streaming_coverage_printer() if case["taker"] == "stream"
             else CoverageTaker.new_pileup_coverage_coverage_printer())
    fp = cli.FilterParameters(cli.FlagFilter(*case["ff"]))
    names, lens = files[0].ref_names, files[0].ref_lens
    mask = g_of = None
    if case["api"] == "names":
        genomes, c2g = case["geco"]
        g_of = np.asarray([c2g.get(n, -1) for n in names], dtype=np.int32)
        mask = (g_of >= 0).astype(np.uint8)
    samples = [provider(af, fp, excl, want_hist, want_id, mask=mask) for af in files]
    if case["api"] == "contig":
        rm = host.contig_coverage(names, lens, samples, taker, est, case["print_zero"])
    elif case["api"] == "sep":
        rm = host.mosdepth_genome_coverage(names, lens, samples, case["sep"], taker, case["print_zero"], est,
                                           case["single"])
    else:
        rm = host.mosdepth_genome_coverage_with_contig_names(names, lens, samples, genomes, g_of, taker,
                                                             case["print_zero"], est)
    assert taker.text() == case["expected"]
    if "reads_mapped" in case:
        assert [(r.num_mapped_reads, r.num_reads) for r in rm] == case["reads_mapped"]


def _sorted_table(s):
    lines = s.split("\n")
    return [lines[0]] + sorted(lines[1:])


def run_cli_case(case, provider):
    files = [alignment_file(b) for b in case["bams"]]
    args = dict(case["args"])
    if "genome_definition" in args:
        args["genome_definition"] = os.path.join(FIXDIR, args["genome_definition"])
    if case["match"] == "error":
        with pytest.raises((O.OracleError, native.CovError)) as ei:
            cli.run(case["mode"], files, sample_provider=provider, **args)
        if isinstance(ei.value, native.CovError):
            assert ei.value.status == native.ERR_UNSORTED and case["expected"] in ei.value.message
        return
    out = cli.run(case["mode"], files, sample_provider=provider, **args)
    if case["match"] == "is":
        assert out == case["expected"]
    elif case["match"] == "contains":
        assert case["expected"] in out
    elif case["match"] == "contains_all":
        for e in case["expected"]:
            assert e in out
    else:
        assert _sorted_table(out) == _sorted_table(case["expected"])


@pytest.mark.parametrize("case", cases.API_CASES, ids=[c["id"] for c in cases.API_CASES])
def test_host_api_golden_cpu(case):
    run_api_case(case, oracle_sample)


@pytest.mark.parametrize("case", cases.CLI_CASES, ids=[c["id"] for c in cases.CLI_CASES])
def test_host_cli_golden_cpu(case):
    run_cli_case(case, oracle_sample)


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.API_CASES, ids=[c["id"] for c in cases.API_CASES])
def test_api_golden_gpu(case):
    run_api_case(case, cli.device_sample)


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.CLI_CASES, ids=[c["id"] for c in cases.CLI_CASES])
def test_cli_golden_gpu(case):
    run_cli_case(case, cli.device_sample)


@pytest.mark.parametrize("case", [c for c in cases.FILTER_CASES if c.get("mode") == (False, True) or
                                  (c.get("mode") is None) or c.get("mode") == (True, True)],
                         ids=lambda c: c["id"])
def test_pair_mode_order(case):
    """Host-side mate pairing (filter.rs:117-228) reproduces the reference's qname order."""
    af = alignment_file(case["bam"])
    fp = cli.FilterParameters(cli.FlagFilter(*case["ff"]), case["single"][0], case["single"][1], case["single"][2],
                              case["mapq"], case["pair"][0], case["pair"][1], case["pair"][2])
    fs, fpairs = fp.filter_mode()
    assert fpairs
    if case.get("mode") is not None:
        assert (fs, fpairs) == case["mode"]
    order = cli.pair_mode_order(af, fp)
    if "count" in case:
        assert len(order) == case["count"]
        return
    got = [af.qname[i].decode() for i in order]
    if case["exhaustive"]:
        assert got == case["qnames"]
    else:
        assert got[:len(case["qnames"])] == case["qnames"]


def test_format_matches_rust_display():
    assert host.format_f32(1.2) == "1.2"
    assert host.format_f32(500000.0) == "500000"
    assert host.format_f32(np.float32(0.011293635)) == "0.011293635"
    assert host.format_f32(np.float32(0.00035077872)) == "0.00035077872"
    assert host.format_f32(0.0) == "0"
    assert host.format_f32(np.float32(1e-7)) == "0.0000001"
    assert host.format_f64(900000.0357627869) == "900000.0357627869"
    assert host.format_f32(np.float32(17538.936)) == "17538.936"
    for v in np.random.default_rng(0).random(200).astype(np.float32) * np.float32(1000):
        assert host.format_f32(v) == O.fmt_f32(v)


def test_library_exports_every_declared_symbol():
    """Every function declared in include/*.h is exported by libcovermhip.so (parsed from the headers, so a declaration
    without a definition cannot slip through)."""
    import re
    L = native.lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for h in ("covermhip.h", "coverm_host.h"):
        text = open(os.path.join(root, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)                       # comments mention function names too
        text = re.sub(r"typedef[^;{]*\([^;]*;", " ", text)                         # function-pointer typedefs
        declared |= set(re.findall(r"\b(covh?_[a-z0-9_]+)\s*\(", text))
    assert len(declared) > 45 and {"cov_finish", "cov_interval_stats_compute", "covh_gene_coverage", "covh_pair_mode_order",
                                   "cov_host_alloc", "covh_bam_open"} <= declared
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    for name in native.EXPORTS:
        assert hasattr(L, name)
    assert L.cov_abi_version() == 1


def _paired_sample(n_pairs, seed):
    """Coordinate-sorted synthetic proper pairs with names: 85 % both mates on one contig, the rest split over two
    contigs (never paired), a few names used by two pairs (the parked-set state machine), a few with missing mates."""
    import numpy as np
    from oracle.bamio import BamData
    rng = np.random.default_rng(seed)
    ref_lens = np.asarray([50_000, 120_000, 8_000, 300_000, 20_000], dtype=np.int64)
    rows = []   # tid, pos, flag, mapq, nm, lseq, cig, mtid, name
    for p in range(n_pairs):
        name = b"p%d" % (p if rng.random() > 0.02 else int(rng.integers(0, max(1, p))))
        t1 = int(rng.integers(0, 5))
        t2 = t1 if rng.random() < 0.85 else int(rng.integers(0, 5))
        for k, (t, mt) in enumerate(((t1, t2), (t2, t1))):
            if k == 1 and rng.random() < 0.03:
                continue                                    # mate absent from the file
            L = int(ref_lens[t])
            ln = int(rng.integers(60, 151))
            cig = [(ln << 4)] if rng.random() < 0.8 else [((ln // 2) << 4), (int(rng.integers(1, 4)) << 4) | int(rng.integers(1, 3)), ((ln - ln // 2) << 4)]
            flag = (99 if k == 0 else 147) if rng.random() < 0.9 else (97 if k == 0 else 145)   # 10 % not proper pairs
            if rng.random() < 0.02: flag |= 0x100
            rows.append((t, int(rng.integers(0, L - 200)), flag, int(rng.integers(0, 61)), int(rng.integers(0, 12)), ln + int(rng.integers(0, 20)), cig, mt, name))
    rows.sort(key=lambda r: (r[0], r[1]))
    n = len(rows)
    coff = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum([len(r[6]) for r in rows], out=coff[1:])
    z = np.zeros(n, np.int32)
    return BamData(["c%d" % i for i in range(5)], ref_lens, np.asarray([r[0] for r in rows], np.int32),
                   np.asarray([r[1] for r in rows], np.int32), np.asarray([r[2] for r in rows], np.uint16),
                   np.asarray([r[3] for r in rows], np.uint8), np.asarray([r[5] for r in rows], np.int32),
                   np.asarray([r[4] for r in rows], np.uint32), np.ones(n, np.uint8), coff,
                   np.asarray([w for r in rows for w in r[6]], np.uint32), np.asarray([r[7] for r in rows], np.int32), z, z,
                   [r[8] for r in rows], "")


@pytest.mark.parametrize("threads", [1, 7])
@pytest.mark.parametrize("params", [dict(min_percent_identity_pair=0.95), dict(min_aligned_length_pair=200, min_mapq=20),
                                    dict(min_percent_identity_single=0.9, min_aligned_percent_pair=0.8),
                                    dict(min_mapq=30, proper=True)])
def test_pair_stage_cpp_matches_oracle_at_scale(params, threads):
    """The threaded C++ pair stage (csrc/host_filter.cpp) returns exactly the oracle's order on 40 k synthetic pairs."""
    import numpy as np
    from coverm_amd.engine import RecordBatch
    from oracle import oracle as O
    params = dict(params)
    proper = params.pop("proper", False)
    b = _paired_sample(40_000, seed=3)
    ofp = O.FilterParameters(O.FlagFilter(not proper, True, False), **params)
    fs, fpairs = O.filter_mode(ofp)
    assert fpairs
    want, _ = O.reader_stage(b, ofp)
    rec = RecordBatch.from_arrays(b.tid, b.pos, b.flag, b.mapq, b.nm, b.nm_kind, b.l_seq, b.cigar_off, b.cigar)
    af = cli.AlignmentFile("x.bam", b.ref_names, b.ref_lens, rec, b.qname, b.mtid)
    fp = cli.FilterParameters(cli.FlagFilter(not proper, True, False), **params)
    got = cli.pair_mode_order(af, fp, threads=threads)
    assert len(got) > 1000
    np.testing.assert_array_equal(got, np.asarray(want, dtype=np.int64))


def test_contig_scan_over_thousands_of_contigs_is_the_same_in_parallel(monkeypatch):
    """covh_contig_coverage computes the coverages of >= 1024 contigs on a few worker threads and feeds the taker in order afterwards:
    the text must equal the one-thread result and the oracle's CLI text (zero rows, reads-mapped line inputs, RPKM / TPM included)."""
    from coverm_amd import synth
    from oracle.bamio import BamData
    ref = synth.make_reference(2600, 9_000_000, seed=9, min_len=1500, max_len=20_000)
    batch = synth.make_reads(ref, 40_000, seed=10)
    z = np.zeros(batch.n_records, np.int32)
    b = BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq.astype(np.int32), batch.nm, batch.nm_kind,
                batch.cigar_off, batch.cigar, z, z, z, [], "")
    af = AlignmentFile("data/many.bam", ref.names, ref.lengths, batch)

    def provider(af_, fp, excl, want_hist, want_identity, mask=None, device=0):
        off = O.FlagFilter(fp.flag_filters.include_improper_pairs, fp.flag_filters.include_supplementary, fp.flag_filters.include_secondary)
        st, hist, prim = O.integer_stats(b, off, None, excl, mask)
        out = np.zeros(len(st), dtype=native.CONTIG_STATS_DTYPE)
        for f in ("n_primary", "n_pass", "n_nonsupp", "sum_nm", "sum_indel", "win_sum_d", "win_sum_d2", "win_covered", "full_covered", "first_record",
                  "last_record", "win_min_d", "win_max_d", "hist_len", "hist_off"):
            out[f] = st[f]
        out["sum_identity_primary"] = st["id_primary"]; out["sum_identity_nonsupp"] = st["id_nonsupp"]
        return SampleResult(af_.stoit_name, out, hist if want_hist else None, prim)
    for kw in (dict(methods=["mean", "trimmed_mean", "covered_fraction", "variance", "rpkm", "tpm", "anir", "count"]),
               dict(methods=["mean", "variance"], no_zeros=True, output_format="sparse")):
        set_knobs(monkeypatch, finalise_threads=4)
        par = cli.run("contig", [af], sample_provider=provider, **kw)
        set_knobs(monkeypatch, finalise_threads=1)
        ser = cli.run("contig", [af], sample_provider=provider, **kw)
        assert par == ser == O.run_cli("contig", ["data/many.bam"], bams=[b], **kw)
        assert par.count("\n") > 1000
