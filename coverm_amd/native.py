"""ctypes binding of libcovermhip.so (include/covermhip.h).

This is the same binding shape a Rust `extern "C"` block would have.  Loading fails
loudly when the shared library is missing: there is no Python or CPU fallback for the hot path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcovermhip.so")

COV_OK = 0
ERR_UNSORTED, ERR_NM_MISSING, ERR_NM_BADTYPE, ERR_POS_OOB, ERR_BAD_CIGAR, ERR_BAD_TID = 1, 2, 3, 4, 6, 7
ERR_INVALID_ARG, ERR_HIP, ERR_STATE = 16, 17, 18
WANT_HIST, WANT_IDENTITY = 1, 2
WANT_IDENTITY_PRIMARY_ONLY, WANT_IDENTITY_NONSUPP_ONLY = 4, 8
K_PREP, K_RANGES, K_PILEUP, K_IDENTITY, K_HIST, K_HIST_COMPACT, K_ESTIMATE, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7
KERNEL_NAMES = {K_PREP: "k_prep", K_RANGES: "k_ranges", K_PILEUP: "k_pileup", K_IDENTITY: "k_identity",
                K_HIST: "k_hist", K_HIST_COMPACT: "k_hist_compact", K_ESTIMATE: "k_estimate"}


class CovConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("include_improper_pairs", C.c_uint8), ("include_supplementary", C.c_uint8),
                ("include_secondary", C.c_uint8), ("filter_single", C.c_uint8), ("min_mapq", C.c_uint8),
                ("reserved0", C.c_uint8 * 3), ("min_aligned_length", C.c_uint32),
                ("min_percent_identity", C.c_float), ("min_aligned_percent", C.c_float),
                ("contig_end_exclusion", C.c_uint64), ("want", C.c_uint32), ("reserved1", C.c_uint32)]


class CovBatch(C.Structure):
    _fields_ = [("tid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
                ("nm", C.c_void_p), ("nm_kind", C.c_void_p), ("l_seq", C.c_void_p), ("cigar_off", C.c_void_p),
                ("cigar", C.c_void_p), ("n_records", C.c_uint64)]


class CovSummary(C.Structure):
    _fields_ = [("num_detected_primary_alignments", C.c_uint64), ("n_records", C.c_uint64),
                ("n_considered", C.c_uint64), ("hist_total", C.c_uint64)]


# numpy mirror of cov_contig_stats (128 bytes)
CONTIG_STATS_DTYPE = np.dtype([
    ("n_primary", "<u8"), ("n_pass", "<u8"), ("n_nonsupp", "<u8"), ("sum_nm", "<u8"), ("sum_indel", "<u8"),
    ("sum_identity_primary", "<f8"), ("sum_identity_nonsupp", "<f8"), ("win_sum_d", "<u8"), ("win_sum_d2", "<u8"),
    ("win_covered", "<u8"), ("full_covered", "<u8"), ("first_record", "<u8"), ("last_record", "<u8"),
    ("win_min_d", "<u4"), ("win_max_d", "<u4"), ("hist_len", "<u4"), ("reserved", "<u4"), ("hist_off", "<u8")])
assert CONTIG_STATS_DTYPE.itemsize == 128

EXPORTS = ["cov_abi_version", "cov_create", "cov_destroy", "cov_last_error", "cov_set_targets",
           "cov_set_target_mask", "cov_push_batch", "cov_push_batch_device", "cov_finish", "cov_fetch_hist",
           "cov_copy_depth", "cov_reset", "cov_kernel_ms", "cov_algorithmic_bytes", "cov_last_paths"]

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def _torch_hip_first():
    """PyTorch-ROCm ships its own copy of the HIP runtime.  If torch is already imported in this process its runtime
    must come up BEFORE libcovermhip.so is loaded: the loader then binds our NEEDED libamdhip64 to the copy torch
    loaded (same SONAME) and both share one runtime; in the other order torch later reports "No HIP GPUs are available"."""
    import sys
    t = sys.modules.get("torch")
    if t is None:
        return
    try:
        if t.cuda.is_available():
            t.cuda.init()
    except Exception:
        pass


def lib():
    """Loads libcovermhip.so.  Raises if it has not been built: the HIP path is the only path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            "%s not found — build it with `python -m coverm_amd.build` (hipcc, gfx950). "
            "coverm_amd has no CPU fallback for the coverage hot path." % LIB_PATH)
    _torch_hip_first()
    L = C.CDLL(LIB_PATH)
    L.cov_abi_version.restype = C.c_int
    L.cov_last_error.restype = C.c_char_p
    L.cov_last_error.argtypes = [C.c_void_p]
    L.cov_create.argtypes = [C.POINTER(CovConfig), C.POINTER(C.c_void_p)]
    L.cov_destroy.argtypes = [C.c_void_p]
    L.cov_destroy.restype = None
    L.cov_set_targets.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.cov_set_target_mask.argtypes = [C.c_void_p, C.c_void_p]
    L.cov_push_batch.argtypes = [C.c_void_p, C.POINTER(CovBatch)]
    L.cov_push_batch_device.argtypes = [C.c_void_p, C.POINTER(CovBatch)]
    L.cov_finish.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CovSummary)]
    L.cov_fetch_hist.argtypes = [C.c_void_p, C.c_void_p]
    L.cov_copy_depth.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.cov_reset.argtypes = [C.c_void_p]
    L.cov_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.cov_algorithmic_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.cov_last_paths.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 4)]
    L.cov_set_estimators.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.cov_fetch_estimates.argtypes = [C.c_void_p, C.c_void_p]
    _lib = L
    return L


class CovError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("covermhip status %d: %s" % (status, message))
        self.status = status
        self.message = message
