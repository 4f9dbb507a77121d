"""CoverageEstimator::calculate_coverage on the device (cov_set_estimators / cov_fetch_estimates, kernel k_estimate) against the host's
evaluation (csrc/host_coverage.cpp `calculate`, itself pinned on the reference's goldens through the oracle): the same f32, bit for
bit, for every estimator the device offers — /root/reference src/mosdepth_genome_coverage_estimators.rs:530-839 — on the reference's
fixture BAMs and on synthetic samples (deep contigs: histograms of several 64-bin batches; long reads; contigs shorter than the end
exclusion; every min_covered_fraction / trim setting of the goldens)."""
import numpy as np
import pytest

from coverm_amd import host, synth
from coverm_amd.engine import FilterConfig, RecordBatch, Session
from coverm_amd.host import CoverageEstimator as E
from coverm_amd.native import CovError
from tests.fixtures import load_fixture
from tests.golden import cases
from tests.test_gpu_abi_parity import _long_read_batch, to_batch
from tests.knobs import set_knobs

pytestmark = pytest.mark.gpu


def estimator_sets(excl):
    return [
        [E.new_estimator_mean(0.0, excl, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, excl), E.new_estimator_covered_fraction(0.0),
         E.new_estimator_variance(0.0, excl)],
        [E.new_estimator_mean(0.1, excl, True), E.new_estimator_trimmed_mean(0.1, 0.9, 0.1, excl), E.new_estimator_covered_bases(0.1),
         E.new_estimator_variance(0.3, excl), E.new_estimator_length(), E.new_estimator_read_count(), E.new_estimator_reads_per_base(),
         E.new_estimator_rpkm(0.0), E.new_estimator_rpkm(0.5), E.new_estimator_anir(), E.new_estimator_covered_fraction(0.76)],
        [E.new_estimator_trimmed_mean(0.0, 1.0, 0.0, excl), E.new_estimator_trimmed_mean(0.5, 0.5, 0.0, excl), E.new_estimator_trimmed_mean(0.9, 0.1, 0.0, excl),
         E.new_estimator_trimmed_mean(0.25, 0.75, 0.0, excl), E.new_estimator_trimmed_mean(0.0, 0.01, 0.0, excl), E.new_estimator_trimmed_mean(0.99, 1.0, 0.0, excl)],
    ]


def both_ways(names, lens, batch, est, excl, ff=(True, True, False), chunks=1):
    """(device floats as the taker recorded them, the host evaluation's, both ReadsMapped) of one sample."""
    out = []
    for dev in (True, False):
        with Session(0, FilterConfig(*ff), excl, want_hist=True, want_identity="primary") as s:
            s.set_targets(lens)
            if dev:
                s.set_estimators(est)
            edges = np.linspace(0, batch.n_records, chunks + 1).astype(int)
            for lo, hi in zip(edges[:-1], edges[1:]):
                s.push(batch.slice(lo, hi))
            st, summ = s.finish()
            taker = host.CoverageTaker.new_cached_single_float_coverage_taker(len(est))
            if dev:
                ef = s.estimates()
                sample = host.SampleResult("s", st, None, int(summ.num_detected_primary_alignments))
                rm = host.contig_coverage(names, lens, [sample], taker, est, True, estimates=[ef])
                # the rows of contigs without a considered record are zeros
                assert (ef[st["n_pass"] == 0] == 0).all()
            else:
                sample = host.SampleResult("s", st, s.hist(), int(summ.num_detected_primary_alignments))
                rm = host.contig_coverage(names, lens, [sample], taker, est, True)
            out.append((taker.cached_coverages(0), (rm[0].num_mapped_reads, rm[0].num_reads)))
    return out


def assert_same(out):
    (dev, rm_d), (hst, rm_h) = out
    assert dev.shape == hst.shape
    np.testing.assert_array_equal(dev.view(np.uint32), hst.view(np.uint32))      # bit for bit, NaN included
    assert rm_d == rm_h


@pytest.mark.parametrize("name", [f for f in cases.FIXTURE_FILES if "unsorted" not in f])
@pytest.mark.parametrize("excl", [0, 75])
def test_fixtures_every_estimator(name, excl):
    b = load_fixture(name)
    for est in estimator_sets(excl):
        assert_same(both_ways(list(b.ref_names), np.asarray(b.ref_lens, np.int64), to_batch(b), est, excl))


def test_deep_contigs_histograms_of_many_batches():
    """Depths in the hundreds: the trimmed mean's walk crosses several 64-bin batches, its start and its end in different ones."""
    ref = synth.make_reference(12, 60_000, seed=3, min_len=900, max_len=9_000)
    batch = synth.make_reads(ref, 180_000, seed=4)
    for excl in (0, 75, 600):
        for est in estimator_sets(excl):
            assert_same(both_ways(ref.names, ref.lengths, batch, est, excl, chunks=3))


def test_short_read_sample_and_long_reads():
    ref = synth.make_reference(400, 30_000_000, seed=7, min_len=120, max_len=500_000)     # some contigs shorter than 2 x 75
    batch = synth.make_reads(ref, 300_000, seed=8)
    for est in estimator_sets(75):
        assert_same(both_ways(ref.names, ref.lengths, batch, est, 75, ff=(True, False, True)))
    lens = np.asarray([400_000, 90_000, 1_200_000, 300], np.int64)
    lb = _long_read_batch(lens, 1_500, 20_000, seed=9)
    for est in estimator_sets(0)[:2]:
        assert_same(both_ways(["c%d" % i for i in range(len(lens))], lens, lb, est, 0))


def test_a_lane_per_contig(monkeypatch):
    """k_estimate_lanes (the session's choice from 65 536 contigs on, forced here): a lane walks its contig's bins one by one, the whole wave
    the contigs with more than 96 bins — fixtures, deep contigs (histograms of several hundred bins beside shallow ones in one wave), contigs
    shorter than the end exclusion."""
    monkeypatch.setenv("COVERM_EST_LANES", "1")
    for name in ["7seqs.reads_for_seq1_and_seq2.bam", "2seqs.reads_for_seq1.bam", "k141_2005182.bam", "eg2.bam"]:
        b = load_fixture(name)
        for excl in (0, 75):
            for est in estimator_sets(excl):
                assert_same(both_ways(list(b.ref_names), np.asarray(b.ref_lens, np.int64), to_batch(b), est, excl))
    ref = synth.make_reference(12, 60_000, seed=3, min_len=900, max_len=9_000)
    batch = synth.make_reads(ref, 180_000, seed=4)
    for excl in (0, 75, 600):
        for est in estimator_sets(excl):
            assert_same(both_ways(ref.names, ref.lengths, batch, est, excl, chunks=3))
    ref = synth.make_reference(700, 30_000_000, seed=7, min_len=120, max_len=500_000)
    batch = synth.make_reads(ref, 300_000, seed=8)
    for est in estimator_sets(75):
        assert_same(both_ways(ref.names, ref.lengths, batch, est, 75, ff=(True, False, True)))


def test_an_assembly_of_many_short_contigs():
    """More contigs than k_prep_lean's loop is launched for (fewer than 128 records per contig: k_prep_generic walks every step), the
    session's own choice of k_estimate_lanes (>= 65 536 contigs), the histogram layout over many blocks, convert_results on threads."""
    ref = synth.make_reference(70_000, 90_000_000, seed=21, min_len=1000, max_len=40_000)
    batch = synth.make_reads(ref, 900_000, seed=22)
    for est in estimator_sets(75)[:2]:
        assert_same(both_ways(ref.names, ref.lengths, batch, est, 75))


def test_with_spills_of_the_bounded_store(monkeypatch):
    set_knobs(monkeypatch, store_cap_records=40000)
    ref = synth.make_reference(150, 12_000_000, seed=11, min_len=1500, max_len=300_000)
    batch = synth.make_reads(ref, 200_000, seed=12)
    for est in estimator_sets(75)[:2]:
        assert_same(both_ways(ref.names, ref.lengths, batch, est, 75, chunks=19))


def test_estimators_the_device_does_not_offer_are_refused():
    with Session(0, FilterConfig(), 75, want_hist=True) as s:
        s.set_targets([1000])
        for bad in (E.new_estimator_tpm(0.0), E.new_estimator_pileup_counts(0.0, 75)):
            with pytest.raises(CovError):
                s.set_estimators([bad])
        with pytest.raises(CovError):
            s.set_estimators([E.new_estimator_anir()])          # no identity sums asked of this session
    with Session(0, FilterConfig(), 75, want_hist=False) as s:
        s.set_targets([1000])
        with pytest.raises(CovError):
            s.set_estimators([E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, 75)])
        s.set_estimators([E.new_estimator_mean(0.0, 75, False)])
        s.set_estimators([])
        s.finish()
        with pytest.raises(CovError):
            s.estimates()
