"""The standalone C++ CLI (coverm_amd/coverm-amd): BAM file on disk -> C++ reader -> C ABI -> kernels ->
C++ host layer -> stdout, compared with the reference's own CLI expectations (tests/test_cmdline.rs)."""
import os
import subprocess

import pytest

from oracle import bamio
from tests.fixtures import FIXDIR, load_fixture
from tests.golden import cases

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


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.CLI_CASES, ids=[c["id"] for c in cases.CLI_CASES])
def test_cli_binary_golden(case, tmp_path):
    paths = []
    for b in case["bams"]:
        stem = os.path.splitext(b)[0]
        p = str(tmp_path / (stem + ".bam"))     # the stoit name is the file stem
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
