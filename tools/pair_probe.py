"""Pair-mode reader stage (csrc/host_filter.cpp: covh_pair_mode_order + covh_batch_select) timed at scale, against the decoder.

Synthetic proper pairs: mate 1 at a random position, mate 2 150-450 bases further on the same contig, shared read name; the
records are coordinate sorted, as a real BAM is.  Prints records/s for the hash-join and for the gather of the selected records."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from coverm_amd import native, synth  # noqa: E402
from tests.harness_cli import _PairFilter  # noqa: E402
from coverm_amd.native import CovBatch  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ref = synth.make_reference(5000, 1_000_000_000, seed=1)
t0 = time.time()
rng = np.random.default_rng(5)
L = ref.lengths
w = L * rng.lognormal(0.0, 1.0, len(L))
cdf = np.cumsum(w) / w.sum()
tid1 = np.minimum(np.searchsorted(cdf, rng.random(pairs), side="right"), len(L) - 1).astype(np.int32)
pos1 = (rng.random(pairs) * np.maximum(L[tid1] - 800, 1)).astype(np.int32)
pos2 = pos1 + rng.integers(150, 450, pairs).astype(np.int32)
tid = np.concatenate([tid1, tid1]); pos = np.concatenate([pos1, pos2])
pid = np.concatenate([np.arange(pairs), np.arange(pairs)]).astype(np.int64)
flag = np.concatenate([np.full(pairs, 99, np.uint16), np.full(pairs, 147, np.uint16)])
order = np.argsort((tid.astype(np.int64) << 32) | pos.astype(np.int64), kind="stable")
tid, pos, pid, flag = tid[order], pos[order], pid[order], flag[order]
n = 2 * pairs
# fixed-width names "p%010d"
digits = np.zeros((n, 11), np.uint8)
digits[:, 0] = ord("p")
x = pid.copy()
for k in range(10, 0, -1):
    digits[:, k] = 48 + (x % 10)
    x //= 10
blob = digits.tobytes()
qoff = (np.arange(n + 1, dtype=np.uint64) * 11).astype(np.uint32)
cigar = np.full(n, (150 << 4), np.uint32)
coff = np.arange(n + 1, dtype=np.uint32)
rb = dict(tid=tid, pos=pos, flag=flag, mapq=np.full(n, 30, np.uint8), nm=rng.poisson(1.5, n).astype(np.uint32), nm_kind=np.ones(n, np.uint8),
          l_seq=np.full(n, 150, np.uint32), cigar_off=coff, cigar=cigar)
print("generated %d pairs in %.1fs" % (pairs, time.time() - t0), flush=True)
Lb = native.lib()
cb = CovBatch()
for k, a in rb.items():
    setattr(cb, k, a.ctypes.data)
cb.n_records = n
pf = _PairFilter(0, 255, 0, 0.0, 0.0, 100, np.float32(0.95), np.float32(0.0))      # --min-read-aligned-length-pair 100 --min-read-percent-identity-pair 95
mtid = np.ascontiguousarray(tid, np.int32)
Lb.covh_pair_mode_order.argtypes = [C.POINTER(CovBatch), C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(_PairFilter), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
Lb.covh_batch_select.argtypes = [C.POINTER(CovBatch), C.c_void_p, C.c_uint64, C.c_int, C.POINTER(CovBatch)]
Lb.covh_batch_free.argtypes = [C.POINTER(CovBatch)]
Lb.covh_free.argtypes = [C.c_void_p]
for thr in (threads, 1):
    out = C.c_void_p(); n_out = C.c_uint64(0)
    t0 = time.time()
    rc = Lb.covh_pair_mode_order(C.byref(cb), mtid.ctypes.data, qoff.ctypes.data, blob, C.byref(pf), thr, C.byref(out), C.byref(n_out))
    t1 = time.time() - t0
    assert rc == 0
    sel = CovBatch()
    t0 = time.time()
    assert Lb.covh_batch_select(C.byref(cb), out, n_out, thr, C.byref(sel)) == 0
    t2 = time.time() - t0
    print("threads %2d: pair_mode_order %.3fs = %.1f M records/s (%d of %d records kept); batch_select %.3fs = %.1f M records/s" % (
        thr, t1, n / t1 / 1e6, n_out.value, n, t2, n_out.value / max(t2, 1e-9) / 1e6), flush=True)
    Lb.covh_batch_free(C.byref(sel))
    Lb.covh_free(out)
