"""Pins the CPU oracle (oracle/) against every golden vector the reference's own tests hold for
the BAM -> pileup -> per-contig / per-genome path (tests/golden/cases.py cites each file:line)."""
import io
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.fixtures import FIXDIR, load_fixture, stoit
from tests.golden import cases


def make_est(spec):
    k = spec[0]
    if k == "mean": return O.est_mean(spec[1], spec[2], spec[3])
    if k == "variance": return O.est_variance(spec[1], spec[2])
    if k == "trimmed_mean": return O.est_trimmed_mean(spec[1], spec[2], spec[3], spec[4])
    if k == "pileup_counts": return O.est_pileup_counts(spec[1], spec[2])
    if k == "covered_fraction": return O.est_covered_fraction(spec[1])
    if k == "covered_bases": return O.est_covered_bases(spec[1])
    if k == "rpkm": return O.est_rpkm(spec[1])
    if k == "tpm": return O.est_tpm(spec[1])
    if k == "length": return O.est_length()
    if k == "read_count": return O.est_read_count()
    if k == "reads_per_base": return O.est_reads_per_base()
    if k == "anir": return O.est_anir()
    raise ValueError(k)


@pytest.mark.parametrize("case", cases.API_CASES, ids=[c["id"] for c in cases.API_CASES])
def test_oracle_api_golden(case):
    bams = [load_fixture(b) for b in case["bams"]]
    names = [stoit(b) for b in case["bams"]]
    s = io.StringIO()
    taker = O.StreamingTaker(s) if case["taker"] == "stream" else O.PileupTaker(s)
    ff = O.FlagFilter(*case["ff"])
    est = [make_est(e) for e in case["est"]]
    if case["api"] == "contig":
        rm = O.contig_coverage(bams, names, taker, est, case["print_zero"], ff)
    elif case["api"] == "sep":
        rm = O.genome_coverage_separator(bams, names, case["sep"], taker, case["print_zero"], est, ff,
                                         case["single"])
    else:
        genomes, c2g = case["geco"]
        rm = O.genome_coverage_with_contig_names(bams, names, genomes, c2g, taker, case["print_zero"], ff, est)
    assert s.getvalue() == case["expected"]
    if "reads_mapped" in case:
        assert [(r.num_mapped_reads, r.num_reads) for r in rm] == case["reads_mapped"]


@pytest.mark.parametrize("case", cases.FILTER_CASES, ids=[c["id"] for c in cases.FILTER_CASES])
def test_oracle_filter_golden(case):
    b = load_fixture(case["bam"])
    fp = O.FilterParameters(O.FlagFilter(*case["ff"]), case["single"][0], case["single"][1], case["single"][2],
                            case["mapq"], case["pair"][0], case["pair"][1], case["pair"][2])
    if case.get("mode") is not None:
        assert O.filter_mode(fp) == case["mode"]
    order, prim = O.reader_stage(b, fp)
    if "count" in case:
        assert len(order) == case["count"]
        return
    got = [b.qname[i].decode() for i in order]
    if case["exhaustive"]:
        assert got == case["qnames"]
    else:
        assert got[:len(case["qnames"])] == case["qnames"]


@pytest.mark.parametrize("case", cases.FILTER_INVERSE_CASES, ids=[c["id"] for c in cases.FILTER_INVERSE_CASES])
def test_oracle_filter_inverse_golden(case):
    """filter.rs's own tests with filter_out = false: what `coverm filter --inverse` returns."""
    b = load_fixture(case["bam"])
    fp = O.FilterParameters(O.FlagFilter(*case["ff"]), case["single"][0], case["single"][1], case["single"][2],
                            case["mapq"], case["pair"][0], case["pair"][1], case["pair"][2])
    got = [b.qname[i].decode() for i in O.reader_filter(b, fp, filter_out=False)]
    if case["exhaustive"]:
        assert got == case["qnames"]
    else:
        assert got[:len(case["qnames"])] == case["qnames"]


@pytest.mark.parametrize("case", cases.FILTER_CASES, ids=[c["id"] for c in cases.FILTER_CASES])
def test_oracle_reader_filter_equals_reader_stage_when_filtering_out(case):
    """The record-by-record restatement (both values of filter_out) against the reader stage the coverage path uses, filter_out = true."""
    b = load_fixture(case["bam"])
    fp = O.FilterParameters(O.FlagFilter(*case["ff"]), case["single"][0], case["single"][1], case["single"][2],
                            case["mapq"], case["pair"][0], case["pair"][1], case["pair"][2])
    order, prim = O.reader_stage(b, fp)
    assert list(O.reader_filter(b, fp, True)) == list(order)


def _sorted_table(s):
    lines = s.split("\n")
    return [lines[0]] + sorted(lines[1:])


@pytest.mark.parametrize("case", cases.CLI_CASES, ids=[c["id"] for c in cases.CLI_CASES])
def test_oracle_cli_golden(case):
    bams = [load_fixture(b) for b in case["bams"]]
    args = dict(case["args"])
    if "genome_definition" in args:
        args["genome_definition"] = os.path.join(FIXDIR, args["genome_definition"])
    if case["match"] == "error":
        with pytest.raises(O.OracleError) as ei:
            O.run_cli(case["mode"], case["bams"], bams=bams, **args)
        assert ei.value.kind == "unsorted" and case["expected"] in O.UNSORTED_MESSAGE
        return
    out = O.run_cli(case["mode"], case["bams"], bams=bams, **args)
    if case["match"] == "is":
        assert out == case["expected"]
    elif case["match"] == "contains":
        assert case["expected"] in out
    elif case["match"] == "contains_all":
        for e in case["expected"]:
            assert e in out
    elif case["match"] == "table":
        assert _sorted_table(out) == _sorted_table(case["expected"])


def test_rust_float_display():
    assert O.fmt_f32(np.float32(1.2)) == "1.2"
    assert O.fmt_f32(np.float32(500000.0)) == "500000"
    assert O.fmt_f32(np.float32(0.011293635)) == "0.011293635"
    assert O.fmt_f32(np.float32(0.00035077872)) == "0.00035077872"
    assert O.fmt_f32(0.0) == "0"
    assert O.fmt_f64(900000.0357627869) == "900000.0357627869"
    assert O.fmt_f32(np.float32(1e-7)) == "0.0000001"


# ---- per-gene coverage (--gff)
def test_oracle_gff_parsing():
    assert O.read_gff(os.path.join(FIXDIR, "2seqs.gff")) == cases.GFF_PARSE_EXPECTED
    assert O.read_gff(os.path.join(FIXDIR, "2seqs.gff"), "CDS") == []


def _gene_est(spec):
    return {"mean": lambda: O.est_mean(*spec[1:]), "count": lambda: O.est_read_count()}[spec[0]]()


@pytest.mark.parametrize("case", cases.GENE_API_CASES, ids=[c["id"] for c in cases.GENE_API_CASES])
def test_oracle_gene_api_golden(case):
    import io
    b = load_fixture(case["bam"])
    out = io.StringIO()
    namer = (lambda c: case["namer"].get(c)) if case["namer"] is not None else None
    O.gene_coverage([b], [os.path.splitext(case["bam"])[0]], O.StreamingTaker(out), [_gene_est(case["est"])], case["genes"],
                    namer, case["print_zeros"], O.FlagFilter(True, False, False))
    assert out.getvalue() == case["expected"]


@pytest.mark.parametrize("case", cases.GENE_CLI_CASES, ids=[c["id"] for c in cases.GENE_CLI_CASES])
def test_oracle_gene_cli_golden(case):
    bams = [load_fixture(b) for b in case["bams"]]
    args = dict(case["args"])
    for k in ("gff", "genome_definition"):
        if k in args:
            args[k] = os.path.join(FIXDIR, args[k])
    out = O.run_cli(case["mode"], case["bams"], bams=bams, **args)
    for e in case["expected"]:
        assert e in out, out
