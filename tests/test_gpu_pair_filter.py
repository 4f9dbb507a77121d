"""Reader-stage pair filter on the device (cov_pair_filter_apply, csrc/pair_kernels.hip.h): over what the device ingest extracted
(mate reference + read-name hash) it must select exactly the records, in exactly the order, that the oracle's restatement of
ReferenceSortedBamFilter::read (filter.rs:117-228) returns — on the reference's own filter goldens, on synthetic pairs at scale,
with read names that repeat (the park / take-out sequence), and across table chunks."""
import numpy as np
import pytest

from coverm_amd import bam as cbam
from coverm_amd.engine import FilterConfig, Session
from oracle import bamio
from oracle import oracle as O
from tests.fixtures import load_fixture
from tests.golden import cases
from tests.test_host_golden import _paired_sample
from tests.knobs import set_knobs

pytestmark = pytest.mark.gpu
FIELDS = ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq")


def _device_selection(path, ofp, threads=3):
    fs, fpairs = O.filter_mode(ofp)
    assert fpairs
    with Session(0, FilterConfig(), 75) as s:
        cbam.gpu_ingest(s, path, threads=threads, want_mates=True)
        nsel, nprim = cbam.pair_filter_apply(s, fs, ofp.min_mapq, (ofp.min_aligned_length_single, ofp.min_percent_identity_single, ofp.min_aligned_percent_single),
                                             (ofp.min_aligned_length_pair, ofp.min_percent_identity_pair, ofp.min_aligned_percent_pair))
        got = cbam.session_records(s)
    assert got.n_records == nsel
    return got, nprim


def _expect(b, ofp):
    order, prim = O.reader_stage(b, ofp)
    return np.asarray(order, dtype=np.int64), prim


def _compare(got, b, order):
    assert got.n_records == len(order)
    for f in FIELDS:
        np.testing.assert_array_equal(getattr(got, f), np.asarray(getattr(b, f))[order].astype(getattr(got, f).dtype), err_msg=f)
    ncig = (b.cigar_off[1:].astype(np.int64) - b.cigar_off[:-1].astype(np.int64))[order]
    np.testing.assert_array_equal(np.diff(got.cigar_off.astype(np.int64)), ncig)
    exp_cig = np.concatenate([b.cigar[b.cigar_off[i]:b.cigar_off[i + 1]] for i in order]) if len(order) else np.zeros(0, np.uint32)
    np.testing.assert_array_equal(got.cigar, exp_cig)


PAIR_CASES = [c for c in cases.FILTER_CASES if c.get("mode") in ((False, True), (True, True), None)]


@pytest.mark.parametrize("case", PAIR_CASES, ids=lambda c: c["id"])
def test_reference_filter_goldens_on_the_device(tmp_path, case):
    """filter.rs:342-844: the expected read-name sequences of the reference's own tests, device ingest forced."""
    d = load_fixture(case["bam"])
    p = str(tmp_path / "f.bam")
    bamio.write_bam(p, d, level=6)
    ofp = O.FilterParameters(O.FlagFilter(*case["ff"]), case["single"][0], case["single"][1], case["single"][2], case["mapq"],
                             case["pair"][0], case["pair"][1], case["pair"][2])
    if not O.filter_mode(ofp)[1]:
        pytest.skip("single-read branch (k_prep)")
    got, nprim = _device_selection(p, ofp)
    order, prim = _expect(d, ofp)
    _compare(got, d, order)
    assert nprim == prim
    if "count" in case:
        assert got.n_records == case["count"]
    elif case.get("exhaustive"):
        assert [d.qname[i].decode() for i in order] == case["qnames"]


@pytest.mark.parametrize("chunk", [0, 2048])
@pytest.mark.parametrize("params", [dict(min_percent_identity_pair=0.95), dict(min_aligned_length_pair=200, min_mapq=20),
                                    dict(min_percent_identity_single=0.9, min_aligned_percent_pair=0.8), dict(min_mapq=30, proper=True)])
def test_synthetic_pairs_with_repeated_names(tmp_path, monkeypatch, params, chunk):
    """40 k synthetic pairs: mates on other contigs, missing mates, improper pairs, secondary records and ~2 % of the names used by two
    or more pairs (entries with more than two records: collected, replayed in file order, judged).  chunk = 2048: dozens of table
    chunks cut at reference boundaries."""
    if chunk:
        set_knobs(monkeypatch, pair_chunk=chunk)
    params = dict(params)
    proper = params.pop("proper", False)
    b = _paired_sample(40_000, seed=3)
    p = str(tmp_path / "pairs.bam")
    bamio.write_bam(p, b, level=1)
    ofp = O.FilterParameters(O.FlagFilter(not proper, True, False), **params)
    got, nprim = _device_selection(p, ofp)
    order, prim = _expect(b, ofp)
    assert len(order) > 1000
    _compare(got, b, order)
    assert nprim == prim


def test_missing_nm_in_a_judged_pair_is_the_references_panic(tmp_path):
    b = _paired_sample(3_000, seed=5)
    ofp = O.FilterParameters(O.FlagFilter(True, True, False), min_percent_identity_pair=0.9)
    order, _ = _expect(b, ofp)
    b.nm_kind[order[len(order) // 2]] = bamio.NM_ABSENT
    p = str(tmp_path / "nonm.bam")
    bamio.write_bam(p, b, level=1)
    with pytest.raises(RuntimeError) as ei:
        _device_selection(p, ofp)
    assert "does not have an 'NM' auxiliary tag" in str(ei.value)


def test_pair_filter_needs_mate_columns(tmp_path):
    b = _paired_sample(500, seed=6)
    p = str(tmp_path / "x.bam")
    bamio.write_bam(p, b, level=1)
    with Session(0, FilterConfig(), 75) as s:
        cbam.gpu_ingest(s, p, threads=2)
        with pytest.raises(RuntimeError):
            cbam.pair_filter_apply(s, False, 255, pair=(1, 0.0, 0.0))
