"""BASELINE.json configs as parity cases: the PRODUCT BINARY's text (coverm-amd: BAM file -> device ingest -> kernels -> C++ host
layer -> printer) == the oracle's CLI text on seeded synthetic data written to real BAM files, and size-independent properties at a
large size where the oracle would be too slow to be the checker."""
import os

import numpy as np
import pytest

from coverm_amd import bam as cbam
from coverm_amd import synth
from coverm_amd.engine import FilterConfig, Session
from oracle import oracle as O
from oracle.bamio import BamData
from tests import binary

pytestmark = pytest.mark.gpu

ALL_CONTIG_METHODS = ["mean", "trimmed_mean", "covered_fraction", "covered_bases", "variance", "length", "count",
                      "reads_per_base", "anir", "rpkm", "tpm"]


def make(tmp_path, n_contigs, total, n_reads, seed, name="synth", with_seq=1):
    """Synthetic sample as (reference, records, oracle view, path of the BAM file the binary reads)."""
    ref = synth.make_reference(n_contigs, total, seed=seed, min_len=1500, max_len=400_000)
    batch = synth.make_reads(ref, n_reads, seed=seed + 1)
    z = np.zeros(batch.n_records, np.int32)
    b = BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq.astype(np.int32),
                batch.nm, batch.nm_kind, batch.cigar_off, batch.cigar, z, z, z, [], "")
    path = os.path.join(str(tmp_path), name + ".bam")
    cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=with_seq, threads=8)
    return ref, batch, b, path


@pytest.mark.parametrize("fmt", ["dense", "sparse"])
def test_config2_contig_four_methods(tmp_path, fmt):
    ref, batch, b, path = make(tmp_path, 120, 8_000_000, 150_000, seed=41)
    args = dict(methods=["mean", "trimmed_mean", "covered_fraction", "variance"], output_format=fmt)
    assert binary.run("contig", [path], **args) == O.run_cli("contig", [path], bams=[b], **args)


def test_config3_genome_definition_relative_abundance_rpkm_tpm(tmp_path):
    ref, batch, b, path = make(tmp_path, 200, 10_000_000, 200_000, seed=43)
    gd = tmp_path / "genomes.tsv"
    gd.write_text("".join("%s\t%s\n" % (n.split("~")[0], n) for n in ref.names[:170]))   # 30 contigs in no genome
    for fmt in ("dense", "sparse"):
        args = dict(methods=["relative_abundance", "rpkm", "tpm"], genome_definition=str(gd), output_format=fmt)
        assert binary.run("genome", [path], **args) == O.run_cli("genome", [path], bams=[b], **args)
    args = dict(methods=["relative_abundance", "mean", "covered_bases"], separator="~", output_format="sparse")
    assert binary.run("genome", [path], **args) == O.run_cli("genome", [path], bams=[b], **args)


def test_config4_multi_sample_dense_table(tmp_path):
    ref = synth.make_reference(60, 4_000_000, seed=45, min_len=1500, max_len=300_000)
    paths, bs = [], []
    for k in range(3):
        batch = synth.make_reads(ref, 60_000, seed=50 + k)
        z = np.zeros(batch.n_records, np.int32)
        bs.append(BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq,
                          batch.l_seq.astype(np.int32), batch.nm, batch.nm_kind, batch.cigar_off, batch.cigar, z, z,
                          z, [], ""))
        paths.append(str(tmp_path / ("s%d.bam" % k)))
        cbam.write_bam(paths[-1], ref.names, ref.lengths, batch, with_seq=2, threads=4)
    args = dict(methods=["mean", "variance", "rpkm"])
    assert binary.run("contig", paths, **args) == O.run_cli("contig", paths, bams=bs, **args)


def test_config5_full_filter_path_all_methods(tmp_path):
    ref, batch, b, path = make(tmp_path, 150, 9_000_000, 250_000, seed=47, with_seq=2)
    args = dict(methods=ALL_CONTIG_METHODS, min_read_percent_identity=95, min_read_aligned_length=50,
                proper_pairs_only=True, output_format="sparse")
    got = binary.run("contig", [path], **args)
    assert got == O.run_cli("contig", [path], bams=[b], **args)
    assert got.count("\n") == 151
    # coverage_histogram has its own printer and cannot be combined (coverm.rs:1438-1446)
    args = dict(methods=["coverage_histogram"], min_read_percent_identity=95, min_read_aligned_length=50,
                proper_pairs_only=True)
    assert binary.run("contig", [path], **args) == O.run_cli("contig", [path], bams=[b], **args)


def test_large_size_properties():
    """20 M reads (~0.4x of config 2): laws that must hold whatever the size."""
    ref = synth.make_reference(2000, 400_000_000, seed=1)
    batch = synth.make_reads(ref, 20_000_000, seed=2)
    flag = batch.flag
    considered = ((flag & 0x4) == 0) & ((flag & 0x100) == 0)          # default FlagFilter + mapped
    op = batch.cigar & 15
    ln = (batch.cigar >> 4).astype(np.int64)
    m_len = np.where((op == 0) | (op == 7) | (op == 8), ln, 0)
    per_rec = np.add.reduceat(np.concatenate([m_len, [0]]), batch.cigar_off[:-1].astype(np.int64))
    per_rec[batch.cigar_off[1:] == batch.cigar_off[:-1]] = 0
    aligned_per_contig = np.bincount(batch.tid[considered], weights=per_rec[considered].astype(np.float64),
                                     minlength=len(ref.lengths)).astype(np.int64)
    reads_per_contig = np.bincount(batch.tid[considered], minlength=len(ref.lengths))
    with Session(0, FilterConfig(), 0, want_hist=True, want_identity=True) as s:
        s.set_targets(ref.lengths)
        s.push(batch)
        st, summ = s.finish()
        hist = s.hist()
        st2, _ = s.finish()
        assert st.tobytes() == st2.tobytes()                                  # deterministic / idempotent
        assert (hist == s.hist()).all()
    assert int(summ.n_considered) == int(considered.sum())
    np.testing.assert_array_equal(st["n_pass"], reads_per_contig)
    # conservation: with no end exclusion the summed depth equals the aligned M/=/X bases (reads never leave contigs)
    np.testing.assert_array_equal(st["win_sum_d"].astype(np.int64), aligned_per_contig)
    for t in range(0, len(ref.lengths), 37):
        h = hist[int(st["hist_off"][t]):int(st["hist_off"][t]) + int(st["hist_len"][t])].astype(np.int64)
        if st["n_pass"][t] == 0:
            continue
        d = np.arange(len(h))
        assert h.sum() == ref.lengths[t]
        assert (h * d).sum() == st["win_sum_d"][t] and (h * d * d).sum() == st["win_sum_d2"][t]
        assert h[1:].sum() == st["win_covered"][t] == st["full_covered"][t]
        assert st["win_max_d"][t] == len(h) - 1 and h[-1] > 0
    # additivity under tid-range sharding (what the multi-GPU path relies on)
    cut_tid = len(ref.lengths) // 2
    cut = int(np.searchsorted(batch.tid, cut_tid, side="left"))
    parts = []
    for lo, hi in ((0, cut), (cut, batch.n_records)):
        with Session(0, FilterConfig(), 0, want_hist=False) as s:
            s.set_targets(ref.lengths)
            s.push(batch.slice(lo, hi))
            parts.append(s.finish()[0])
    for f in ("n_pass", "win_sum_d", "win_sum_d2", "win_covered", "full_covered", "sum_nm"):
        np.testing.assert_array_equal(parts[0][f] + parts[1][f], st[f], err_msg=f)


# ---------------------------------------------------------------------------------- parity at BASELINE sizes, oracle as checker
def _oracle_arrays(ref, batch):
    z = np.zeros(1, np.int32)
    return BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq.astype(np.int32), batch.nm,
                   batch.nm_kind, batch.cigar_off, batch.cigar, z, z, z, [], "")


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    """20 M reads over 2 000 contigs / 200 genomes (0.4x of config 2's sample, same depth profile), also as a BAM file (realistic
    entropy, ~2 GB) for the binary."""
    ref = synth.make_reference(2000, 400_000_000, seed=1)
    batch = synth.make_reads(ref, 20_000_000, seed=2)
    d = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path_factory.mktemp("big"))
    path = os.path.join(d, "coverm_amd_test_big_%d.bam" % os.getpid())
    cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=2, threads=16)
    yield ref, batch, _oracle_arrays(ref, batch), path
    os.remove(path)


def test_config3_full_size_genome_definition_vs_oracle(big, tmp_path):
    """`coverm genome --genome-definition … -m relative_abundance rpkm tpm` at 20 M reads: the whole product path (file -> device
    ingest -> kernels -> C++ genome scan -> printer) against the oracle's C scan (genome.rs:17-322) — text equality, dense and sparse."""
    ref, batch, b, path = big
    gd = tmp_path / "genomes.tsv"
    keep = [n for i, n in enumerate(ref.names) if i % 11 != 3]          # some contigs in no genome (genome.rs:170-171)
    gd.write_text("".join("%s\t%s\n" % (n.split("~")[0], n) for n in keep))
    for fmt in ("dense", "sparse"):
        args = dict(methods=["relative_abundance", "rpkm", "tpm"], genome_definition=str(gd), output_format=fmt)
        got = binary.run("genome", [path], threads=16, **args)
        assert got == O.run_cli("genome", [path], bams=[b], **args)
        assert got.count("\n") >= 200


def test_config5_full_size_filters_all_methods_vs_oracle(big):
    """Config 5's flags (--min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only) and every method at
    20 M reads: per-contig integer statistics, histograms and f64 identity sums bit-exact against the oracle, then the text."""
    ref, batch, b, path = big
    ff = O.FlagFilter(False, True, False)
    fp = O.FilterParameters(ff, 50, float(np.float32(0.95)), 0.0, 255, 0, 0.0, 0.0)
    exp, exp_hist, prim = O.integer_stats(b, ff, fp, 75)
    filt = FilterConfig(False, True, False, filter_single=True, min_mapq=255, min_aligned_length=50,
                        min_percent_identity=float(np.float32(0.95)), min_aligned_percent=0.0)
    with Session(0, filt, 75, want_hist=True, want_identity=True) as s:
        s.set_targets(ref.lengths)
        for lo in range(0, batch.n_records, 6_000_000):              # several pushes, as a streamed ingest delivers them
            s.push(batch.slice(lo, min(batch.n_records, lo + 6_000_000)))
        st, summ = s.finish()
        hist = s.hist()
    assert summ.num_detected_primary_alignments == prim
    for f in ("n_primary", "n_pass", "n_nonsupp", "sum_nm", "sum_indel", "win_sum_d", "win_sum_d2", "win_covered", "full_covered",
              "win_min_d", "win_max_d", "hist_len"):
        np.testing.assert_array_equal(st[f], exp[f], err_msg=f)
    np.testing.assert_array_equal(st["sum_identity_primary"].view(np.uint64), exp["id_primary"].view(np.uint64))
    np.testing.assert_array_equal(hist, exp_hist)
    args = dict(methods=ALL_CONTIG_METHODS, min_read_percent_identity=95, min_read_aligned_length=50, proper_pairs_only=True)
    assert binary.run("contig", [path], threads=16, **args) == O.run_cli("contig", [path], bams=[b], **args)


def test_unsorted_input_with_histograms_is_a_clean_error():
    """Records of several contigs interleaved (a name-sorted BAM) with a histogram-based method: the histogram arena is sized by
    R + n_targets bins, and every contig's bound must stay within its own considered-record count — the reference's panic
    ('BAM file appears to be unsorted') must come back as COV_ERR_UNSORTED, not as a memory fault."""
    from coverm_amd.native import CovError, ERR_UNSORTED
    ref = synth.make_reference(300, 20_000_000, seed=5, min_len=1500, max_len=300_000)
    batch = synth.make_reads(ref, 400_000, seed=6)
    rng = np.random.default_rng(3)
    perm = rng.permutation(batch.n_records)                      # destroy the order completely
    nops = (batch.cigar_off[1:] - batch.cigar_off[:-1])[perm]
    off = np.zeros(batch.n_records + 1, np.uint32)
    np.cumsum(nops, out=off[1:])
    src = np.repeat(batch.cigar_off[:-1][perm].astype(np.int64) - off[:-1], nops) + np.arange(int(off[-1]))
    from coverm_amd.engine import RecordBatch
    shuf = RecordBatch(batch.tid[perm], batch.pos[perm], batch.flag[perm], batch.mapq[perm], batch.nm[perm], batch.nm_kind[perm],
                       batch.l_seq[perm], off, batch.cigar[src])
    for want_hist in (True, False):
        with Session(0, FilterConfig(), 75, want_hist=want_hist) as s:
            s.set_targets(ref.lengths)
            s.push(shuf)
            with pytest.raises(CovError) as ei:
                s.finish()
            assert ei.value.status == ERR_UNSORTED
            # the session is still usable
            s.reset()
            s.push(batch)
            st, _ = s.finish()
            assert int(st["n_pass"].sum()) > 0


def _paired_view(ref, batch):
    """The oracle's view of what the writer's with_seq = 3 puts into the file: records 2k / 2k + 1 on one reference are mates (one
    name, next_refID = their reference); every other record has a name of its own."""
    n = batch.n_records
    i = np.arange(n, dtype=np.int64)
    m = i ^ 1
    same = (m < n) & (batch.tid[np.minimum(m, n - 1)] == batch.tid)
    name_id = np.where(same, i & ~1, i)
    z = np.zeros(n, np.int32)
    return BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq.astype(np.int32), batch.nm, batch.nm_kind,
                   batch.cigar_off, batch.cigar, batch.tid.copy(), z, z, [b"n%d" % k for k in name_id], "")


@pytest.mark.parametrize("args", [dict(min_read_percent_identity_pair=95, min_read_aligned_length_pair=200, proper_pairs_only=True),
                                  dict(min_mapq=20, proper_pairs_only=True),
                                  dict(min_read_percent_identity=97, min_read_aligned_percent_pair=90)])
def test_pair_mode_filters_through_the_binary_use_the_device_join(tmp_path, args):
    """Pair-mode reader filters (filter.rs:117-228; --min-mapq with --proper-pairs-only selects the pair branch, filter.rs:48-61) on
    a paired file: the binary keeps mates on the device (device ingest + cov_pair_filter_apply, no whole-file host decode) and its
    text equals the oracle's; the host path (COVERM_PAIR_ON_HOST=1) gives the same text."""
    ref = synth.make_reference(150, 9_000_000, seed=47, min_len=1500, max_len=400_000)
    batch = synth.make_reads(ref, 300_000, seed=48)
    path = str(tmp_path / "pairs.bam")
    cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=3, threads=8)
    b = _paired_view(ref, batch)
    kw = dict(methods=["mean", "trimmed_mean", "covered_fraction", "variance", "count", "anir"], **args)
    want = O.run_cli("contig", [path], bams=[b], **kw)
    r = __import__("subprocess").run(binary.argv("contig", [path], **kw), capture_output=True, text=True, env=dict(os.environ, COVERM_CLI_TIMING="1"))
    assert r.returncode == 0, r.stderr
    assert "pair filter on the device" in r.stderr
    assert r.stdout == want
    assert binary.run("contig", [path], env={"COVERM_PAIR_ON_HOST": "1"}, **kw) == want
