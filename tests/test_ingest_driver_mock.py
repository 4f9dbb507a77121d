"""Host driver of the device ingest (covh_bam_gpu_ingest: reader thread, per-chunk header hop, block table, staging slots)
against a CPU mock of cov_ingest_* that inflates every block from exactly the bytes fed and checks ISIZE / CRC-32 / order.
The mock only exists in the sanitizer build of tools/asan_host.sh (which runs this file under ASan + UBSan); with the real
library these tests skip — the same driver is then covered on the GPU by tests/test_gpu_ingest.py."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from coverm_amd import native
from tests.knobs import with_knobs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_mock():
    try:
        return hasattr(native.lib(), "cov_mock_ingest_present")
    except Exception:
        return False


WORKER = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r)
import numpy as np
from coverm_amd import bam as cbam, native, synth
from oracle import bamio
from tests.fixtures import load_fixture
L = native.lib()
mode, tmp = sys.argv[1], sys.argv[2]
if mode == "synth":
    ref = synth.make_reference(40, 6_000_000, seed=18, min_len=5000, max_len=800_000)
    b = synth.make_reads(ref, 60_000, seed=19)
    p = os.path.join(tmp, "s.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=3)
    n_expect = b.n_records
else:
    d = load_fixture("eg2.bam")
    p = os.path.join(tmp, "f.bam")
    bamio.write_bam(p, d, level=6, block=700)       # thousands of tiny blocks: many per chunk, many straddling pieces
    n_expect = len(d.tid)
cfg = native.CovConfig()
sess = C.c_void_p()
assert L.cov_create(C.byref(cfg), C.byref(sess)) == 0
err = C.create_string_buffer(512)
L.covh_bam_read_header.restype = C.c_void_p
hd = L.covh_bam_read_header(p.encode(), err, 512)
assert hd, err.value
L.covh_bam_gpu_ingest.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
n = C.c_uint64(0)
t = (C.c_double * 8)()
rc = L.covh_bam_gpu_ingest(p.encode(), 3, sess, hd, 1, C.byref(n), t, err, 512)
assert rc == 0, (rc, err.value)
assert n.value == n_expect, (n.value, n_expect)
print("MOCK_OK", mode, n.value, "io=" + str(int(t[7])))
'''


# what timing[7] must say afterwards: 0 staging slots, 1 mapped file registered piece by piece, 2 mapped and registered up front
IO_MODES = {"default": ({}, 3), "pread": ({"COVERM_INGEST_IO": "pread"}, 0), "mmap": ({"COVERM_INGEST_IO": "mmap"}, 1),
            "mmap-upfront": ({"COVERM_INGEST_IO": "mmap-upfront"}, 2),
            "mmap_refused": ({"COVERM_INGEST_IO": "mmap", "COVERM_MOCK_NO_REGISTER": "1"}, 3),
            "mmap-upfront_refused": ({"COVERM_INGEST_IO": "mmap-upfront", "COVERM_MOCK_NO_REGISTER": "1"}, 3)}


@pytest.mark.parametrize("io", sorted(IO_MODES))
@pytest.mark.parametrize("mode,piece_kb", [("synth", 0), ("synth", 64), ("synth", 200), ("tiny_blocks", 64), ("tiny_blocks", 0)])
def test_ingest_driver_feeds_consistent_blocks(tmp_path, mode, piece_kb, io):
    """io: the bytes come from staging slots filled from a mapping of the file with non-temporal stores (the default since round 6:
    timing[7] = 3), from staging slots filled by pread (0), from the registered mapping of the file (registered piece by piece, 1, or the
    whole span up front, 2), or the registration is refused by the runtime and the driver copies the mapping's bytes into staging slots
    by itself (3).  The driver reports which it used."""
    if not _has_mock():
        pytest.skip("needs the sanitizer build's CPU mock of cov_ingest_* (tools/asan_host.sh); the GPU suite covers the driver otherwise")
    env = dict(os.environ)
    env.pop("COVERM_INGEST_IO", None)
    if piece_kb:
        env = with_knobs(env, ingest_piece_kb=piece_kb)
    extra, code = IO_MODES[io]
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", WORKER % ROOT, mode, str(tmp_path)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "MOCK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "io=%d" % code in r.stdout, r.stdout[-500:]
