"""covh_contig_coverage_estimated (coverm_host.h): the scan loop of contig.rs:40-104 with calculate_coverage already done elsewhere.
On the CPU the floats fed in are the host's own evaluation (on a GPU they come from k_estimate, tests/test_gpu_estimates.py): rows,
zero rows, streamed text and ReadsMapped must be those of covh_contig_coverage."""
import numpy as np
import pytest

from coverm_amd import host
from coverm_amd.host import CoverageEstimator as E, CoverageTaker
from oracle import oracle as O
from tests.test_host_golden import alignment_file, oracle_sample

FP = O.FilterParameters(O.FlagFilter(True, True, False))
EST = [E.new_estimator_mean(0.0, 75, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, 75), E.new_estimator_covered_fraction(0.1),
       E.new_estimator_variance(0.0, 75), E.new_estimator_length(), E.new_estimator_read_count()]


@pytest.mark.parametrize("name", ["7seqs.reads_for_seq1_and_seq2.bam", "2seqs.reads_for_seq1_and_seq2.bam", "tpm_test.bam"])
@pytest.mark.parametrize("print_zero", [True, False])
def test_estimated_equals_evaluated(name, print_zero):
    af = alignment_file(name)
    sample = oracle_sample(af, FP, 75, True, False)
    t1 = CoverageTaker.new_cached_single_float_coverage_taker(len(EST))
    rm1 = host.contig_coverage(af.ref_names, af.ref_lens, [sample], t1, EST, True)       # every contig's row, zeros included
    full = t1.cached_coverages(0).reshape(len(af.ref_lens), len(EST))
    est_rows = np.where((sample.stats["n_pass"] > 0)[:, None], full, 0.0).astype(np.float32)
    for kind in ("cached", "stream"):
        mk = (lambda: CoverageTaker.new_cached_single_float_coverage_taker(len(EST))) if kind == "cached" else CoverageTaker.new_single_float_coverage_streaming_coverage_printer
        a, b = mk(), mk()
        rma = host.contig_coverage(af.ref_names, af.ref_lens, [sample], a, EST, print_zero)
        rmb = host.contig_coverage(af.ref_names, af.ref_lens, [sample], b, EST, print_zero, estimates=[est_rows])
        assert (rma[0].num_mapped_reads, rma[0].num_reads) == (rmb[0].num_mapped_reads, rmb[0].num_reads)
        if kind == "cached":
            np.testing.assert_array_equal(a.cached_coverages(0).view(np.uint32), b.cached_coverages(0).view(np.uint32))
        else:
            assert a.text() == b.text()
    assert rm1[0].num_reads == sample.num_detected_primary_alignments


def test_histogram_and_tpm_estimators_are_refused_with_floats():
    af = alignment_file("7seqs.reads_for_seq1_and_seq2.bam")
    sample = oracle_sample(af, FP, 75, True, False)
    for bad in (E.new_estimator_tpm(0.0), E.new_estimator_pileup_counts(0.0, 75)):
        t = CoverageTaker.new_cached_single_float_coverage_taker(1)
        with pytest.raises(Exception):
            host.contig_coverage(af.ref_names, af.ref_lens, [sample], t, [bad], True, estimates=[np.zeros((len(af.ref_lens), 1), np.float32)])


def _stats(n, zero_every=0):
    from coverm_amd import native
    st = np.zeros(n, dtype=native.CONTIG_STATS_DTYPE)
    st["n_pass"] = 3
    st["n_primary"] = 2
    if zero_every:
        st["n_pass"][::zero_every] = 0
    return st


def test_bulk_fill_of_a_cached_taker_over_several_samples():
    """Every target an entry of a cached taker: names and (entry, coverage) pairs are written in one pass (covh_taker::names_bulk).  A
    second sample over the same header takes the same path; one over other names must still be noticed
    (coverage_takers.rs:140-148: CoverM stops when the reference sets differ)."""
    n = 1500
    names = ["contig_%d" % i for i in range(n)]
    lens = np.full(n, 5000, np.int64)
    est = [E.new_estimator_mean(0.0, 0, False), E.new_estimator_length()]
    rng = np.random.default_rng(3)
    taker = CoverageTaker.new_cached_single_float_coverage_taker(len(est))
    want = []
    for k in range(3):
        st = _stats(n, zero_every=5 + k)
        ef = rng.random((n, 2)).astype(np.float32)
        rm = host.contig_coverage(names, lens, [host.SampleResult("s%d" % k, st, None, 7)], taker, est, True, estimates=[ef])
        exp = np.where((st["n_pass"] > 0)[:, None], ef, np.asarray([0.0, 5000.0], np.float32)[None, :]).astype(np.float32)   # zero rows: 0 and the length
        want.append(exp)
        assert rm[0].num_mapped_reads == int((((exp[:, :1] > 0) | (exp[:, 1:] > 0)).ravel() * (st["n_pass"] > 0) * st["n_primary"]).sum())
    for k in range(3):
        np.testing.assert_array_equal(taker.cached_coverages(k).reshape(n, 2).view(np.uint32), want[k].view(np.uint32))
    assert not taker.names_mismatch()
    other = ["other_%d" % i for i in range(n)]
    host.contig_coverage(other, lens, [host.SampleResult("s3", _stats(n), None, 7)], taker, est, True, estimates=[np.zeros((n, 2), np.float32)])
    assert taker.names_mismatch()
