"""Runs in its own process (the window size of the device ingest is read once per process): tests/test_gpu_ingest.py sets
COVERM_KNOBS ingest_round_blocks / ingest_carry_kb so that small files are parsed in many windows.
argv: mode (short | long | carry_overflow) and a scratch directory."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coverm_amd import bam as cbam, synth  # noqa: E402
from coverm_amd.engine import FilterConfig, Session  # noqa: E402

FIELDS = ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")
mode, tmp = sys.argv[1], sys.argv[2]
ref = synth.make_reference(40, 6_000_000, seed=18, min_len=5000, max_len=800_000)
if mode == "short":
    b = synth.make_reads(ref, 150_000, seed=23)
else:
    b = synth.make_long_reads(ref, 1500, seed=29, mean_len=20_000)      # records of ~30-100 KB: most windows end inside one
p = os.path.join(tmp, mode + ".bam")
cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=4)
n_blocks, q, raw = 0, 0, open(p, "rb").read()
while q < len(raw):
    q += int.from_bytes(raw[q + 16:q + 18], "little") + 1
    n_blocks += 1
whole = cbam.read_alignment_file(p, threads=2, want_names=False)
with Session(0, FilterConfig(), 75, want_hist=True, want_identity=True) as s:
    if mode == "carry_overflow":
        try:
            cbam.gpu_ingest(s, p, threads=4)
        except cbam.IngestFallback as e:
            assert "carry buffer" in str(e), str(e)
            assert cbam.session_records(s).n_records == 0
            print("WINDOWS_OK %s blocks=%d fallback" % (mode, n_blocks))
            sys.exit(0)
        raise SystemExit("expected a fallback: a record larger than the carry buffer")
    for rep in range(2):        # the second file goes behind the first in the same store
        names, lens, n, _ = cbam.gpu_ingest(s, p, threads=4)
        assert n == whole.records.n_records, (n, whole.records.n_records)
    got = cbam.session_records(s)
    R = whole.records.n_records
    assert got.n_records == 2 * R
    for f in FIELDS[:7]:
        a = getattr(got, f)
        np.testing.assert_array_equal(a[:R], getattr(whole.records, f), err_msg=f)
        np.testing.assert_array_equal(a[R:], getattr(whole.records, f), err_msg=f + " (second file)")
    C = int(whole.records.cigar_off[-1])
    np.testing.assert_array_equal(got.cigar_off[:R + 1], whole.records.cigar_off)
    np.testing.assert_array_equal(got.cigar_off[R:] - C, whole.records.cigar_off)
    np.testing.assert_array_equal(got.cigar[:C], whole.records.cigar)
    np.testing.assert_array_equal(got.cigar[C:], whole.records.cigar)
print("WINDOWS_OK %s blocks=%d records=%d" % (mode, n_blocks, R))
