"""BAM / SAM input through the C++ reader of the host layer (csrc/host_bam.cpp, covh_bam_*).

`read_alignment_file` is what `generate_named_bam_readers_from_bam_files` (bam_generator.rs:356-371) is to the
reference: path in, header + records out (here already as the SoA batch the C ABI takes).
"""
import ctypes as C
import os

import numpy as np

from . import native
from .cli import AlignmentFile
from .engine import RecordBatch
from .native import CovBatch

_bound = False


def _lib():
    global _bound
    L = native.lib()
    if not _bound:
        L.covh_bam_open.restype = C.c_void_p
        L.covh_bam_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.covh_bam_close.argtypes = [C.c_void_p]
        L.covh_bam_close.restype = None
        L.covh_bam_n_targets.restype = C.c_uint32
        L.covh_bam_n_targets.argtypes = [C.c_void_p]
        L.covh_bam_target_name.restype = C.c_char_p
        L.covh_bam_target_name.argtypes = [C.c_void_p, C.c_uint32]
        L.covh_bam_target_len.restype = C.c_uint64
        L.covh_bam_target_len.argtypes = [C.c_void_p, C.c_uint32]
        L.covh_bam_n_records.restype = C.c_uint64
        L.covh_bam_n_records.argtypes = [C.c_void_p]
        L.covh_bam_n_cigar.restype = C.c_uint64
        L.covh_bam_n_cigar.argtypes = [C.c_void_p]
        L.covh_bam_batch.argtypes = [C.c_void_p, C.POINTER(CovBatch)]
        L.covh_bam_batch.restype = None
        L.covh_bam_mtid.restype = C.c_void_p
        L.covh_bam_mtid.argtypes = [C.c_void_p]
        L.covh_bam_qname_off.restype = C.c_void_p
        L.covh_bam_qname_off.argtypes = [C.c_void_p]
        L.covh_bam_qnames.restype = C.c_void_p
        L.covh_bam_qnames.argtypes = [C.c_void_p]
        _bound = True
    return L


def _copy(ptr, dtype, n):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).copy()


def read_alignment_file(path: str, threads: int = None, want_names: bool = True) -> AlignmentFile:
    L = _lib()
    if threads is None:
        threads = min(16, os.cpu_count() or 1)
    err = C.create_string_buffer(512)
    h = L.covh_bam_open(path.encode(), threads, int(want_names), err, 512)
    if not h:
        raise IOError(err.value.decode() or "cannot read %s" % path)
    try:
        nt = L.covh_bam_n_targets(h)
        names = [L.covh_bam_target_name(h, i).decode() for i in range(nt)]
        lens = np.asarray([L.covh_bam_target_len(h, i) for i in range(nt)], dtype=np.int64)
        n = int(L.covh_bam_n_records(h))
        nc = int(L.covh_bam_n_cigar(h))
        cb = CovBatch()
        L.covh_bam_batch(h, C.byref(cb))
        rec = RecordBatch(_copy(cb.tid, np.int32, n), _copy(cb.pos, np.int32, n), _copy(cb.flag, np.uint16, n),
                          _copy(cb.mapq, np.uint8, n), _copy(cb.nm, np.uint32, n), _copy(cb.nm_kind, np.uint8, n),
                          _copy(cb.l_seq, np.uint32, n), _copy(cb.cigar_off, np.uint32, n + 1),
                          _copy(cb.cigar, np.uint32, nc))
        mtid = _copy(L.covh_bam_mtid(h), np.int32, n)
        qn = None
        if want_names:
            off = _copy(L.covh_bam_qname_off(h), np.uint32, n + 1)
            blob = C.string_at(L.covh_bam_qnames(h), int(off[-1])) if n else b""
            qn = [blob[off[i]:off[i + 1]] for i in range(n)]
        return AlignmentFile(path, names, lens, rec, qn, mtid)
    finally:
        L.covh_bam_close(h)


def write_bam(path: str, names, lens, batch: RecordBatch, with_seq: bool = True, level: int = 1, threads: int = None):
    """Threaded BGZF/BAM writer (covh_bam_write) for synthetic inputs."""
    L = _lib()
    if threads is None:
        threads = min(32, os.cpu_count() or 1)
    cb = CovBatch()
    for k in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar"):
        a = getattr(batch, k)
        setattr(cb, k, a.ctypes.data if a.size else None)
    cb.n_records = batch.n_records
    nm = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
    ln = np.ascontiguousarray(lens, dtype=np.uint64)
    L.covh_bam_write.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(CovBatch), C.c_int, C.c_int, C.c_int]
    rc = L.covh_bam_write(path.encode(), len(names), nm, ln.ctypes.data, C.byref(cb), int(with_seq), level, threads)
    if rc:
        raise IOError("covh_bam_write failed (%d)" % rc)
