"""The dense printer's direct path (host_coverage.cpp print_dense: tables of 4096 rows and more whose samples list the same entries in
increasing order are read where the taker keeps them, their rows formatted on several threads) against CoverageTakerTypeIterator's merge
(coverage_takers.rs:265-377, what the reference's print_dense walks, coverage_printer.rs:380-560), and the sparse printer's threaded rows:
the text must be the same byte for byte, normalised columns (relative abundance), RPKM and TPM included."""
import numpy as np
import pytest

from coverm_amd import host
from tests.knobs import set_knobs


def _table(n_entries, n_samples, nc, seed, holes=False):
    rng = np.random.default_rng(seed)
    tk = host.CoverageTaker.new_cached_single_float_coverage_taker(nc)
    rms = []
    for si in range(n_samples):
        tk.start_stoit("sample%d" % si)
        vals = (rng.random((n_entries, nc)) * rng.choice([1.0, 40.0, 1e-3, 1e6], (n_entries, 1))).astype(np.float32)
        vals[rng.random(n_entries) < 0.2] = 0.0
        for e in range(n_entries):
            if holes and si == 1 and e % 97 == 5:
                continue              # an entry one sample does not have: not a plain table
            tk.start_entry(e, "contig_%d%s" % (e, "\r" if e % 1000 == 7 else ""))
            for k in range(nc):
                tk.add_single_coverage(float(vals[e, k]))
            tk.finish_entry()
        rms.append(host.ReadsMapped(int(rng.integers(1000, 5000)), int(rng.integers(5000, 9000))))
    return tk, rms


CASES = [dict(norm=[], rpkm=None, tpm=None), dict(norm=[0], rpkm=None, tpm=None), dict(norm=[0, 2], rpkm=1, tpm=None), dict(norm=[], rpkm=None, tpm=2)]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("n_samples", [1, 3])
def test_dense_direct_path_equals_the_merge(monkeypatch, case, n_samples):
    c = CASES[case]
    heads = ["Mean", "RPKM", "TPM"]
    texts = []
    for plain in (1, 0):
        set_knobs(monkeypatch, printer_plain=plain)
        tk, rms = _table(6000, n_samples, 3, seed=case)
        host.finalise_printing(tk, 2, "Contig", heads, rms, c["norm"], c["rpkm"], c["tpm"])
        texts.append(tk.text())
    assert texts[0] == texts[1] and texts[0].count("\n") == 6001 + (1 if c["norm"] else 0)


def test_a_table_with_a_missing_entry_takes_the_merge(monkeypatch):
    """Sample 1 lacks some entries: the direct path must stand aside (the merge prints zeros for them), whatever the knob says."""
    texts = []
    for plain in (1, 0):
        set_knobs(monkeypatch, printer_plain=plain)
        tk, rms = _table(5000, 2, 2, seed=9, holes=True)
        host.finalise_printing(tk, 2, "Contig", ["Mean", "Variance"], rms, [], None, None)
        texts.append(tk.text())
    assert texts[0] == texts[1] and texts[0].count("\n") == 5001


@pytest.mark.parametrize("printer", [1, 2])
def test_threaded_rows_equal_serial_rows(printer):
    """40 000 rows: more than one formatting thread; the same table in two halves of 20 000 rows stays on one thread — row k's text
    must not depend on which thread wrote it (the halves' rows are the whole table's rows)."""
    heads = ["Mean", "Covered Fraction"]
    whole, rm = _table(40_000, 1, 2, seed=3)
    host.finalise_printing(whole, printer, "Contig", heads, rm, [], None, None)
    rows = whole.text().splitlines()
    rng = np.random.default_rng(3)      # the same values again, entry by entry, through the streaming formatter of single values
    vals = (rng.random((40_000, 2)) * rng.choice([1.0, 40.0, 1e-3, 1e6], (40_000, 1))).astype(np.float32)
    vals[rng.random(40_000) < 0.2] = 0.0
    first = 1 if printer == 2 else 0
    assert len(rows) == 40_000 + first
    for e in (0, 1, 7, 19_999, 20_000, 32_767, 32_768, 39_999):
        want = ("" if printer == 2 else "sample0\t") + "contig_%d" % e + "".join("\t" + host.format_f32(float(v)) for v in vals[e])
        assert rows[first + e] == want, e
