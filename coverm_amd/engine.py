"""Thin Python face of a covermhip session (one sample on one GPU).

Mirrors the C ABI 1:1 — create / set_targets / push / finish / fetch_hist / copy_depth — and adds
nothing of its own: the statistics come from the HIP kernels.  Host arrays are numpy; device arrays
may be anything exposing `data_ptr()` (torch tensors), which is how bench.py hands over HBM-resident
batches without a copy.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import native
from .native import CovBatch, CovConfig, CovError, CovSummary


@dataclass
class RecordBatch:
    """SoA batch of BAM records in file order (cov_batch)."""
    tid: np.ndarray        # int32
    pos: np.ndarray        # int32
    flag: np.ndarray       # uint16
    mapq: np.ndarray       # uint8
    nm: np.ndarray         # uint32
    nm_kind: np.ndarray    # uint8
    l_seq: np.ndarray      # uint32
    cigar_off: np.ndarray  # uint32, n+1
    cigar: np.ndarray      # uint32

    @property
    def n_records(self):
        return int(self.tid.shape[0])

    @staticmethod
    def from_arrays(tid, pos, flag, mapq, nm, nm_kind, l_seq, cigar_off, cigar):
        a = np.ascontiguousarray
        return RecordBatch(a(tid, np.int32), a(pos, np.int32), a(flag, np.uint16), a(mapq, np.uint8),
                           a(nm, np.uint32), a(nm_kind, np.uint8), a(np.asarray(l_seq).astype(np.uint32)),
                           a(cigar_off, np.uint32), a(cigar, np.uint32))

    def slice(self, lo, hi):
        """Records [lo, hi) sharing the CIGAR array (cigar_off keeps absolute offsets)."""
        return RecordBatch(self.tid[lo:hi], self.pos[lo:hi], self.flag[lo:hi], self.mapq[lo:hi], self.nm[lo:hi],
                           self.nm_kind[lo:hi], self.l_seq[lo:hi], self.cigar_off[lo:hi + 1], self.cigar)


def _ptr(x):
    if isinstance(x, np.ndarray):
        return x.ctypes.data if x.size else None
    return x.data_ptr()  # torch tensor


@dataclass
class FilterConfig:
    """FlagFilter (lib.rs:60-64) + the single-read thresholds of FilterParameters (coverm.rs:1648-1657)."""
    include_improper_pairs: bool = True
    include_supplementary: bool = True
    include_secondary: bool = False
    filter_single: bool = False
    min_mapq: int = 255
    min_aligned_length: int = 0
    min_percent_identity: float = 0.0
    min_aligned_percent: float = 0.0


def make_config(device: int = 0, filt: Optional[FilterConfig] = None, contig_end_exclusion: int = 75,
                want_hist: bool = False, want_identity=False) -> CovConfig:
    """cov_config from a FilterConfig.  want_identity: False, True (both sums) or "primary" / "nonsupp"."""
    filt = filt or FilterConfig()
    cfg = CovConfig()
    cfg.device = device
    cfg.include_improper_pairs = int(filt.include_improper_pairs)
    cfg.include_supplementary = int(filt.include_supplementary)
    cfg.include_secondary = int(filt.include_secondary)
    cfg.filter_single = int(filt.filter_single)
    cfg.min_mapq = filt.min_mapq
    cfg.min_aligned_length = filt.min_aligned_length
    cfg.min_percent_identity = filt.min_percent_identity
    cfg.min_aligned_percent = filt.min_aligned_percent
    cfg.contig_end_exclusion = contig_end_exclusion
    cfg.want = (native.WANT_HIST if want_hist else 0) | (native.WANT_IDENTITY if want_identity else 0)
    if want_identity == "primary":
        cfg.want |= native.WANT_IDENTITY_PRIMARY_ONLY
    elif want_identity == "nonsupp":
        cfg.want |= native.WANT_IDENTITY_NONSUPP_ONLY
    return cfg


class Session:
    def __init__(self, device: int = 0, filt: Optional[FilterConfig] = None, contig_end_exclusion: int = 75,
                 want_hist: bool = False, want_identity: bool = False):
        self._lib = native.lib()
        cfg = make_config(device, filt, contig_end_exclusion, want_hist, want_identity)
        self.cfg = cfg
        self._h = C.c_void_p()
        st = self._lib.cov_create(C.byref(cfg), C.byref(self._h))
        if st != native.COV_OK:
            raise CovError(st, self._lib.cov_last_error(None).decode())
        self.n_targets = 0
        self.target_len = None
        self._keep = []   # host/device arrays that must outlive the pushes
        self.summary = None
        self.stats = None

    # -- lifecycle
    def close(self):
        if self._h:
            self._lib.cov_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st):
        if st != native.COV_OK:
            raise CovError(st, self._lib.cov_last_error(self._h).decode())

    # -- ABI
    def set_targets(self, target_len, mask=None):
        tl = np.ascontiguousarray(target_len, dtype=np.uint64)
        self._check(self._lib.cov_set_targets(self._h, len(tl), tl.ctypes.data if len(tl) else None))
        self.n_targets = len(tl)
        self.target_len = tl
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert len(m) == len(tl)
            self._check(self._lib.cov_set_target_mask(self._h, m.ctypes.data if len(m) else None))

    def _batch(self, b, n=None):
        cb = CovBatch()
        for k in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar"):
            setattr(cb, k, _ptr(getattr(b, k) if not isinstance(b, dict) else b[k]))
        cb.n_records = n if n is not None else b.n_records
        return cb

    def push(self, batch: RecordBatch):
        cb = self._batch(batch)
        self._check(self._lib.cov_push_batch(self._h, C.byref(cb)))

    def push_device(self, tensors: dict, n_records: int):
        """`tensors`: dict of device arrays (torch) with the cov_batch field names; adopted without a copy."""
        self._keep.append(tensors)
        cb = self._batch(tensors, n_records)
        self._check(self._lib.cov_push_batch_device(self._h, C.byref(cb)))

    def reset(self):
        self._check(self._lib.cov_reset(self._h))
        self._keep.clear()

    def finish(self):
        stats = np.empty(self.n_targets, dtype=native.CONTIG_STATS_DTYPE)      # (cov_finish writes every entry)
        summ = CovSummary()
        self._check(self._lib.cov_finish(self._h, stats.ctypes.data if self.n_targets else None, C.byref(summ)))
        self.stats = stats
        self.summary = summ
        return stats, summ

    def hist(self):
        n = int(self.summary.hist_total)
        h = np.zeros(n, dtype=np.uint64)
        self._check(self._lib.cov_fetch_hist(self._h, h.ctypes.data if n else None))
        return h

    def depth(self, tid: int):
        d = np.zeros(int(self.target_len[tid]), dtype=np.int32)
        self._check(self._lib.cov_copy_depth(self._h, tid, d.ctypes.data if d.size else None))
        return d

    def set_estimators(self, estimators):
        """cov_set_estimators: CoverageEstimator::calculate_coverage of every contig on the device at each finish (contig mode).
        `estimators`: coverm_amd.host.CoverageEstimator structures (same layout as cov_estimator); [] turns it off."""
        n = len(estimators)
        arr = (type(estimators[0]) * n)(*estimators) if n else None
        self._n_est = n
        self._check(self._lib.cov_set_estimators(self._h, arr, C.c_uint32(n)))

    def estimates(self):
        """cov_fetch_estimates: n_targets x n_estimators f32 of the last finish."""
        out = np.empty((self.n_targets, getattr(self, "_n_est", 0)), dtype=np.float32)
        self._check(self._lib.cov_fetch_estimates(self._h, out.ctypes.data if out.size else None))
        return out

    def kernel_ms(self):
        out = {}
        for k, name in native.KERNEL_NAMES.items():
            ms = C.c_double(0)
            n = C.c_uint32(0)
            self._check(self._lib.cov_kernel_ms(self._h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def last_paths(self):
        """cov_last_paths: which branches of the device pipeline the last finish took (diagnostics for the parity tests)."""
        v = (C.c_uint32 * 4)()
        self._check(self._lib.cov_last_paths(self._h, C.byref(v)))
        return dict(listed_steps=int(v[0]), generic_only=bool(v[1]), slow_tiles=int(v[2]), bucket_records=int(v[3]))

    def algorithmic_bytes(self):
        b = C.c_uint64(0)
        self._check(self._lib.cov_algorithmic_bytes(self._h, C.byref(b)))
        return int(b.value)
