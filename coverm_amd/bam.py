"""BAM / SAM input through the C++ reader of the host layer (csrc/host_bam.cpp, covh_bam_*).

`read_alignment_file` is what `generate_named_bam_readers_from_bam_files` (bam_generator.rs:356-371) is to the
reference: path in, header + records out (here already as the SoA batch the C ABI takes).
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import native
from .engine import RecordBatch
from .native import CovBatch

_bound = False



@dataclass
class AlignmentFile:
    """A decoded BAM/SAM: header + records in file order (+ mate fields for pair-mode filtering)."""
    path: str
    ref_names: List[str]
    ref_lens: np.ndarray
    records: "RecordBatch"
    qname: Optional[List[bytes]] = None
    mtid: Optional[np.ndarray] = None

    @property
    def stoit_name(self):  # bam_generator.rs:358-365: file stem
        return os.path.splitext(os.path.basename(self.path))[0]


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


def stream_batches(path: str, threads: int = None, span_index: int = 0, span_count: int = 1, stats: dict = None):
    """Generator over the streamed reader (covh_bam_stream_*): yields (ref_names, ref_lens) first, then one RecordBatch
    (copied out of the page-locked ring) per window.  `stats`, if given, receives peak_bytes / n_records / timing."""
    L = _lib()
    if not getattr(L, "_stream_bound", False):
        L.covh_bam_stream_open.restype = C.c_void_p
        L.covh_bam_stream_open.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
        L.covh_bam_stream_close.argtypes = [C.c_void_p]
        L.covh_bam_stream_close.restype = None
        L.covh_bam_stream_n_targets.restype = C.c_uint32
        L.covh_bam_stream_n_targets.argtypes = [C.c_void_p]
        L.covh_bam_stream_target_name.restype = C.c_char_p
        L.covh_bam_stream_target_name.argtypes = [C.c_void_p, C.c_uint32]
        L.covh_bam_stream_target_len.restype = C.c_uint64
        L.covh_bam_stream_target_len.argtypes = [C.c_void_p, C.c_uint32]
        L.covh_bam_stream_next.argtypes = [C.c_void_p, C.POINTER(CovBatch)]
        L.covh_bam_stream_error.restype = C.c_char_p
        L.covh_bam_stream_error.argtypes = [C.c_void_p]
        L.covh_bam_stream_peak_bytes.restype = C.c_uint64
        L.covh_bam_stream_peak_bytes.argtypes = [C.c_void_p]
        L.covh_bam_stream_n_records.restype = C.c_uint64
        L.covh_bam_stream_n_records.argtypes = [C.c_void_p]
        L.covh_bam_stream_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.covh_bam_stream_timing.restype = None
        L._stream_bound = True
    if threads is None:
        threads = min(16, os.cpu_count() or 1)
    err = C.create_string_buffer(512)
    h = L.covh_bam_stream_open(path.encode(), threads, span_index, span_count, err, 512)
    if not h:
        raise IOError(err.value.decode() or "cannot read %s" % path)
    try:
        nt = L.covh_bam_stream_n_targets(h)
        yield ([L.covh_bam_stream_target_name(h, i).decode() for i in range(nt)],
               np.asarray([L.covh_bam_stream_target_len(h, i) for i in range(nt)], dtype=np.int64))
        cb = CovBatch()
        while True:
            rc = L.covh_bam_stream_next(h, C.byref(cb))
            if rc < 0:
                raise IOError(L.covh_bam_stream_error(h).decode())
            if rc == 0:
                break
            n = int(cb.n_records)
            off = _copy(cb.cigar_off, np.uint32, n + 1)
            nc = int(off[-1])
            yield RecordBatch(_copy(cb.tid, np.int32, n), _copy(cb.pos, np.int32, n), _copy(cb.flag, np.uint16, n),
                              _copy(cb.mapq, np.uint8, n), _copy(cb.nm, np.uint32, n), _copy(cb.nm_kind, np.uint8, n),
                              _copy(cb.l_seq, np.uint32, n), off, _copy(cb.cigar, np.uint32, nc))
        if stats is not None:
            t = (C.c_double * 8)()
            L.covh_bam_stream_timing(h, t)
            stats.update(peak_bytes=int(L.covh_bam_stream_peak_bytes(h)), n_records=int(L.covh_bam_stream_n_records(h)),
                         timing=dict(read=t[0], inflate=t[1], parse=t[2], wait_inflate=t[3], wait_parse=t[4]))
    finally:
        L.covh_bam_stream_close(h)


def read_streamed(path: str, threads: int = None, span_index: int = 0, span_count: int = 1, stats: dict = None):
    """All batches of the streamed reader concatenated: (ref_names, ref_lens, RecordBatch)."""
    it = stream_batches(path, threads, span_index, span_count, stats)
    names, lens = next(it)
    parts = list(it)
    if not parts:
        z = np.zeros(0, np.uint32)
        return names, lens, RecordBatch(z.astype(np.int32), z.astype(np.int32), z.astype(np.uint16), z.astype(np.uint8), z,
                                        z.astype(np.uint8), z, np.zeros(1, np.uint32), z)
    off = [0]
    for p_ in parts:
        off.append(off[-1] + int(p_.cigar_off[-1]))
    coff = np.concatenate([p_.cigar_off[:-1].astype(np.int64) + o for p_, o in zip(parts, off)] + [[off[-1]]]).astype(np.uint32)
    cat = lambda k: np.concatenate([getattr(p_, k) for p_ in parts])
    return names, lens, RecordBatch(cat("tid"), cat("pos"), cat("flag"), cat("mapq"), cat("nm"), cat("nm_kind"), cat("l_seq"),
                                    coff, cat("cigar"))


def gpu_ingest(session, path: str, threads: int = None, check_crc: bool = True, mask=None, span=(0, 1), want_mates: bool = False):
    """Device ingest (covh_bam_read_header + covh_bam_gpu_ingest): the GPU inflates the BGZF blocks, finds the records and fills
    the session's record store.  Sets the session's targets from the file's header.  Returns (ref_names, ref_lens, n_records,
    timing dict); raises IngestFallback when the file needs the CPU reader.  span = (index, count): one tid span of the file
    (covh_bam_gpu_ingest_span; the spans of a file partition its records in order)."""
    L = _lib()
    if not getattr(L, "_ingest_bound", False):
        L.covh_bam_read_header.restype = C.c_void_p
        L.covh_bam_read_header.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.covh_bam_header_free.argtypes = [C.c_void_p]
        L.covh_bam_header_free.restype = None
        L.covh_bam_header_n_targets.restype = C.c_uint32
        L.covh_bam_header_n_targets.argtypes = [C.c_void_p]
        L.covh_bam_header_target_name.restype = C.c_char_p
        L.covh_bam_header_target_name.argtypes = [C.c_void_p, C.c_uint32]
        L.covh_bam_header_target_len.restype = C.c_uint64
        L.covh_bam_header_target_len.argtypes = [C.c_void_p, C.c_uint32]
        L.covh_bam_header_first_record.restype = C.c_uint64
        L.covh_bam_header_first_record.argtypes = [C.c_void_p]
        L.covh_bam_gpu_ingest_span.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
        L._ingest_bound = True
    if threads is None:
        threads = min(16, os.cpu_count() or 1)
    err = C.create_string_buffer(512)
    hd = L.covh_bam_read_header(path.encode(), err, 512)
    if not hd:
        raise IOError(err.value.decode() or "cannot read %s" % path)
    try:
        nt = L.covh_bam_header_n_targets(hd)
        names = [L.covh_bam_header_target_name(hd, i).decode() for i in range(nt)]
        lens = np.asarray([L.covh_bam_header_target_len(hd, i) for i in range(nt)], dtype=np.int64)
        session.set_targets(lens, mask)
        L.cov_ingest_want_mates.argtypes = [C.c_void_p, C.c_int]
        if L.cov_ingest_want_mates(session._h, int(want_mates)) != 0:
            raise RuntimeError("cov_ingest_want_mates failed")
        n = C.c_uint64(0)
        t = (C.c_double * 8)()
        rc = L.covh_bam_gpu_ingest_span(path.encode(), threads, session._h, hd, int(check_crc), int(span[0]), int(span[1]), C.byref(n), t, err, 512)
        if rc == 1:
            raise IngestFallback(err.value.decode())
        if rc != 0:
            raise IOError(err.value.decode())
        return names, lens, int(n.value), dict(read=t[0], slot_wait=t[1], end=t[2], total=t[3], begin=t[4], walk=t[5], feed=t[6])
    finally:
        L.covh_bam_header_free(hd)


class _DevPairFilter(C.Structure):   # cov_pair_filter
    _fields_ = [("filter_single", C.c_int32), ("min_mapq", C.c_uint8), ("pad", C.c_uint8 * 3), ("min_aligned_length_single", C.c_uint32),
                ("min_percent_identity_single", C.c_float), ("min_aligned_percent_single", C.c_float), ("min_aligned_length_pair", C.c_uint32),
                ("min_percent_identity_pair", C.c_float), ("min_aligned_percent_pair", C.c_float)]


def pair_filter_apply(session, filter_single: bool, min_mapq: int, single=(0, 0.0, 0.0), pair=(0, 0.0, 0.0)):
    """cov_pair_filter_apply: the reader-stage pair filter (filter.rs:117-228) over the store a device ingest with want_mates filled.
    single / pair = (min_aligned_length, min_percent_identity, min_aligned_percent).  Returns (n_selected, n_primary); raises
    IngestFallback when the device declines, NativeError-like RuntimeError with the reference's message on NM errors."""
    L = _lib()
    L.cov_pair_filter_apply.argtypes = [C.c_void_p, C.POINTER(_DevPairFilter), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.cov_last_error.restype = C.c_char_p
    L.cov_last_error.argtypes = [C.c_void_p]
    f = _DevPairFilter(int(filter_single), int(min_mapq), (C.c_uint8 * 3)(), int(single[0]), float(single[1]), float(single[2]),
                       int(pair[0]), float(pair[1]), float(pair[2]))
    nsel, nprim = C.c_uint64(0), C.c_uint64(0)
    rc = L.cov_pair_filter_apply(session._h, C.byref(f), C.byref(nsel), C.byref(nprim))
    if rc == 19:
        raise IngestFallback(L.cov_last_error(session._h).decode())
    if rc != 0:
        raise RuntimeError("cov_pair_filter_apply: %d: %s" % (rc, L.cov_last_error(session._h).decode()))
    return int(nsel.value), int(nprim.value)


class IngestFallback(RuntimeError):
    """The device ingest declined the file (reason in the message): decode it with the CPU reader."""


def session_records(session) -> RecordBatch:
    """Test hook (cov_copy_records): the session's own record store copied back to the host."""
    L = _lib()
    L.cov_copy_records.argtypes = [C.c_void_p, C.POINTER(CovBatch), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    n, nc = C.c_uint64(0), C.c_uint64(0)
    assert L.cov_copy_records(session._h, None, C.byref(n), C.byref(nc)) == 0
    n, nc = int(n.value), int(nc.value)
    rb = RecordBatch(np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint16), np.zeros(n, np.uint8), np.zeros(n, np.uint32),
                     np.zeros(n, np.uint8), np.zeros(n, np.uint32), np.zeros(n + 1, np.uint32), np.zeros(nc, np.uint32))
    cb = CovBatch()
    for k in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar"):
        a = getattr(rb, k)
        setattr(cb, k, a.ctypes.data if a.size else None)
    cb.n_records = n
    assert L.cov_copy_records(session._h, C.byref(cb), None, None) == 0
    return rb


def write_bam(path: str, names, lens, batch: RecordBatch, with_seq=True, level: int = 1, threads: int = None):
    """Threaded BGZF/BAM writer (covh_bam_write) for synthetic inputs.  with_seq: False / 0 = SEQ '*', True / 1 = constant
    SEQ and QUAL, 2 = realistic entropy (random bases, Phred-like qualities, Illumina-style names)."""
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
