"""Python face of the C++ host layer (include/coverm_host.h): estimators, scan drivers, takers, printers.

Names and argument order follow the reference (estimators.rs constructors, contig_coverage,
mosdepth_genome_coverage*, CoverageTakerType, CoveragePrinter) so tests read like the reference's own.
All arithmetic happens in C++ (csrc/host_coverage.cpp); this module only marshals.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import native

MEAN, TRIMMED_MEAN, PILEUP_COUNTS, COVERED_FRACTION, COVERED_BASES, RPKM, TPM, VARIANCE, LENGTH, READ_COUNT, \
    READS_PER_BASE, ANIR = range(12)
TAKER_STREAM, TAKER_PILEUP, TAKER_CACHED = 0, 1, 2
PRINTER_STREAMED, PRINTER_SPARSE, PRINTER_DENSE, PRINTER_METABAT = 0, 1, 2, 3

COLUMN_HEADERS = {  # estimators.rs:84-105
    MEAN: ["Mean"], TRIMMED_MEAN: ["Trimmed Mean"], PILEUP_COUNTS: ["Coverage", "Bases"],
    COVERED_FRACTION: ["Covered Fraction"], COVERED_BASES: ["Covered Bases"], RPKM: ["RPKM"], TPM: ["TPM"],
    VARIANCE: ["Variance"], LENGTH: ["Length"], READ_COUNT: ["Read Count"], READS_PER_BASE: ["Reads per base"],
    ANIR: ["ANIr"]}


class CoverageEstimator(C.Structure):
    """covh_estimator; use the new_estimator_* constructors (estimators.rs:107-224)."""
    _fields_ = [("kind", C.c_int32), ("min_fraction_covered_bases", C.c_float),
                ("contig_end_exclusion", C.c_uint64), ("exclude_mismatches", C.c_int32),
                ("trim_min", C.c_float), ("trim_max", C.c_float)]

    @staticmethod
    def new_estimator_mean(min_fraction_covered_bases, contig_end_exclusion, exclude_mismatches):
        return CoverageEstimator(MEAN, min_fraction_covered_bases, contig_end_exclusion, int(exclude_mismatches), 0, 0)

    @staticmethod
    def new_estimator_trimmed_mean(min, max, min_fraction_covered_bases, contig_end_exclusion):
        return CoverageEstimator(TRIMMED_MEAN, min_fraction_covered_bases, contig_end_exclusion, 0, min, max)

    @staticmethod
    def new_estimator_pileup_counts(min_fraction_covered_bases, contig_end_exclusion):
        return CoverageEstimator(PILEUP_COUNTS, min_fraction_covered_bases, contig_end_exclusion, 0, 0, 0)

    @staticmethod
    def new_estimator_covered_fraction(min_fraction_covered_bases):
        return CoverageEstimator(COVERED_FRACTION, min_fraction_covered_bases, 0, 0, 0, 0)

    @staticmethod
    def new_estimator_covered_bases(min_fraction_covered_bases):
        return CoverageEstimator(COVERED_BASES, min_fraction_covered_bases, 0, 0, 0, 0)

    @staticmethod
    def new_estimator_rpkm(min_fraction_covered_bases):
        return CoverageEstimator(RPKM, min_fraction_covered_bases, 0, 0, 0, 0)

    @staticmethod
    def new_estimator_tpm(min_fraction_covered_bases):
        return CoverageEstimator(TPM, min_fraction_covered_bases, 0, 0, 0, 0)

    @staticmethod
    def new_estimator_variance(min_fraction_covered_bases, contig_end_exclusion):
        return CoverageEstimator(VARIANCE, min_fraction_covered_bases, contig_end_exclusion, 0, 0, 0)

    @staticmethod
    def new_estimator_length():
        return CoverageEstimator(LENGTH, 0, 0, 0, 0, 0)

    @staticmethod
    def new_estimator_read_count():
        return CoverageEstimator(READ_COUNT, 0, 0, 0, 0, 0)

    @staticmethod
    def new_estimator_reads_per_base():
        return CoverageEstimator(READS_PER_BASE, 0, 0, 0, 0, 0)

    @staticmethod
    def new_estimator_anir():
        return CoverageEstimator(ANIR, 0, 0, 0, 0, 0)

    def column_headers(self):
        return COLUMN_HEADERS[self.kind]


class _ReadsMapped(C.Structure):
    _fields_ = [("num_mapped_reads", C.c_uint64), ("num_reads", C.c_uint64)]


class _Sample(C.Structure):
    _fields_ = [("stoit_name", C.c_char_p), ("stats", C.c_void_p), ("hist", C.c_void_p),
                ("num_detected_primary_alignments", C.c_uint64)]


class _Header(C.Structure):
    _fields_ = [("n_targets", C.c_uint32), ("names", C.c_char_p), ("name_off", C.c_void_p),
                ("target_len", C.c_void_p)]


class _Entry(C.Structure):
    _fields_ = [("win_len", C.c_uint64), ("win_sum_d", C.c_uint64), ("win_sum_d2", C.c_uint64),
                ("win_covered", C.c_uint64), ("full_len", C.c_uint64), ("full_covered", C.c_uint64),
                ("n_reads", C.c_uint64), ("mismatches", C.c_uint64), ("win_min_d", C.c_uint32),
                ("hist_len", C.c_uint32), ("hist", C.c_void_p), ("sum_identity", C.c_double)]


@dataclass
class ReadsMapped:  # lib.rs:54-57
    num_mapped_reads: int
    num_reads: int


@dataclass
class SampleResult:
    """One BAM's device output: what cov_finish / cov_fetch_hist returned."""
    stoit_name: str
    stats: np.ndarray                 # native.CONTIG_STATS_DTYPE[n_targets]
    hist: Optional[np.ndarray]        # uint64 or None
    num_detected_primary_alignments: int


_bound = False


def _lib():
    global _bound
    L = native.lib()
    if not _bound:
        L.covh_taker_new.restype = C.c_void_p
        L.covh_taker_new.argtypes = [C.c_int, C.c_size_t]
        L.covh_taker_free.argtypes = [C.c_void_p]
        L.covh_taker_start_stoit.argtypes = [C.c_void_p, C.c_char_p]
        L.covh_taker_start_entry.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p]
        L.covh_taker_add_single_coverage.argtypes = [C.c_void_p, C.c_float]
        L.covh_taker_finish_entry.argtypes = [C.c_void_p]
        L.covh_taker_iterate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.covh_taker_free.restype = None
        L.covh_taker_text.restype = C.c_void_p
        L.covh_taker_text.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.covh_taker_clear_text.argtypes = [C.c_void_p]
        L.covh_taker_cached_coverages.restype = C.c_size_t
        L.covh_taker_cached_coverages.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.covh_last_error.restype = C.c_char_p
        L.covh_wants.restype = C.c_uint32
        L.covh_wants.argtypes = [C.c_void_p, C.c_size_t]
        L.covh_calculate_coverage.restype = C.c_float
        L.covh_format_f32.restype = C.c_size_t
        L.covh_format_f32.argtypes = [C.c_float, C.c_char_p, C.c_size_t]
        L.covh_format_f64.restype = C.c_size_t
        L.covh_format_f64.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
        _bound = True
    return L


class HostError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("coverm host status %d: %s" % (status, message))
        self.status = status


def format_f32(v) -> str:
    b = C.create_string_buffer(512)
    _lib().covh_format_f32(C.c_float(float(v)), b, 512)
    return b.value.decode()


def format_f64(v) -> str:
    b = C.create_string_buffer(512)
    _lib().covh_format_f64(C.c_double(float(v)), b, 512)
    return b.value.decode()


def _est_array(estimators: Sequence[CoverageEstimator]):
    arr = (CoverageEstimator * max(1, len(estimators)))()
    for i, e in enumerate(estimators):
        arr[i] = e
    return arr


def wants(estimators):
    """(want_hist, want_identity) for the covermhip session."""
    w = _lib().covh_wants(_est_array(estimators), len(estimators))
    return bool(w & native.WANT_HIST), bool(w & native.WANT_IDENTITY)


def calculate_coverage(est: CoverageEstimator, *, win_len=0, win_sum_d=0, win_sum_d2=0, win_covered=0, full_len=0,
                       full_covered=0, n_reads=0, mismatches=0, win_min_d=0, hist=None, sum_identity=0.0,
                       unobserved=(0,)) -> float:
    h = np.ascontiguousarray(hist, np.uint64) if hist is not None else None
    e = _Entry(win_len, win_sum_d, win_sum_d2, win_covered, full_len, full_covered, n_reads, mismatches, win_min_d,
               0 if h is None else len(h), None if h is None else h.ctypes.data, sum_identity)
    u = np.asarray(unobserved, dtype=np.uint64)
    return float(_lib().covh_calculate_coverage(C.byref(est), C.byref(e), u.ctypes.data_as(C.c_void_p), len(u)))


class CoverageTaker:
    """CoverageTakerType (coverage_takers.rs:8-72); text accumulates in C++ and is read with .text()."""

    def __init__(self, kind: int, num_coverages: int = 0):
        self._L = _lib()
        self._h = C.c_void_p(self._L.covh_taker_new(kind, num_coverages))
        self.kind = kind
        self.num_coverages = num_coverages

    @staticmethod
    def new_single_float_coverage_streaming_coverage_printer():
        return CoverageTaker(TAKER_STREAM)

    @staticmethod
    def new_pileup_coverage_coverage_printer():
        return CoverageTaker(TAKER_PILEUP)

    @staticmethod
    def new_cached_single_float_coverage_taker(num_coverages):
        return CoverageTaker(TAKER_CACHED, num_coverages)

    def text(self) -> str:
        n = C.c_size_t(0)
        p = self._L.covh_taker_text(self._h, C.byref(n))
        return C.string_at(p, n.value).decode()

    # trait CoverageTaker (coverage_takers.rs:29-38)
    def start_stoit(self, name: str):
        _lib().covh_taker_start_stoit(self._h, name.encode())

    def start_entry(self, entry_order_id: int, entry_name: str):
        _lib().covh_taker_start_entry(self._h, C.c_size_t(entry_order_id), entry_name.encode())

    def add_single_coverage(self, cov: float):
        _lib().covh_taker_add_single_coverage(self._h, C.c_float(cov))

    def finish_entry(self):
        _lib().covh_taker_finish_entry(self._h)

    def names_mismatch(self) -> bool:
        """An entry id arrived with two different names (coverage_takers.rs:140-148: CoverM stops there)."""
        L = _lib()
        L.covh_taker_names_mismatch.argtypes = [C.c_void_p]
        return bool(L.covh_taker_names_mismatch(self._h))

    def iterate(self, num_coverages: int):
        """CoverageTakerTypeIterator: list of (entry_index, stoit_index, [coverages])."""
        L = _lib()
        L.covh_taker_iterate.restype = C.c_size_t
        n = L.covh_taker_iterate(self._h, None, None, None, C.c_size_t(0))
        ei = np.zeros(max(1, n), np.uint64); si = np.zeros(max(1, n), np.uint64)
        cv = np.zeros(max(1, n * num_coverages), np.float32)
        L.covh_taker_iterate(self._h, ei.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p), cv.ctypes.data_as(C.c_void_p),
                             C.c_size_t(n))
        return [(int(ei[i]), int(si[i]), [float(x) for x in cv[i * num_coverages:(i + 1) * num_coverages]]) for i in range(n)]

    def cached_coverages(self, stoit: int = 0) -> np.ndarray:
        """f32 coverages recorded for one sample by the cached taker (entries x estimators, recording order)."""
        n = self._L.covh_taker_cached_coverages(self._h, C.c_size_t(stoit), None, C.c_size_t(0))
        out = np.zeros(n, dtype=np.float32)
        if n:
            self._L.covh_taker_cached_coverages(self._h, C.c_size_t(stoit), out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        return out

    def __del__(self):
        try:
            if self._h:
                self._L.covh_taker_free(self._h)
                self._h = None
        except Exception:
            pass


_header_cache = {}


def _header(names: List[str], target_len):
    """covh_header for (names, lengths); marshalled once per distinct header object (5000 names cost ~0.5 ms)."""
    key = (id(names), len(names), id(target_len))
    hit = _header_cache.get(key)
    if hit is not None and hit[2] is names and hit[3] is target_len:
        return hit[0], hit[1]
    enc = [n.encode() for n in names]
    off = np.zeros(len(enc) + 1, dtype=np.uint32)
    if enc:
        np.cumsum([len(e) for e in enc], out=off[1:])
    tl = np.ascontiguousarray(target_len, dtype=np.uint64)
    blob = b"".join(enc)
    h = _Header(len(enc), blob, off.ctypes.data, tl.ctypes.data)
    if len(_header_cache) > 8:
        _header_cache.clear()
    _header_cache[key] = (h, (blob, off, tl), names, target_len)
    return h, (blob, off, tl)


def _samples(samples: Sequence[SampleResult]):
    arr = (_Sample * max(1, len(samples)))()
    keep = []
    for i, s in enumerate(samples):
        st = np.ascontiguousarray(s.stats)
        assert st.dtype == native.CONTIG_STATS_DTYPE
        hi = np.ascontiguousarray(s.hist, np.uint64) if s.hist is not None else None
        nm = s.stoit_name.encode()
        keep.append((st, hi, nm))
        arr[i] = _Sample(nm, st.ctypes.data if len(st) else None, hi.ctypes.data if hi is not None and len(hi) else None,
                         s.num_detected_primary_alignments)
    return arr, keep


def _finish(rc, rm, n):
    if rc != 0:
        raise HostError(rc, _lib().covh_last_error().decode())
    return [ReadsMapped(int(rm[i].num_mapped_reads), int(rm[i].num_reads)) for i in range(n)]


def contig_coverage(names, target_len, samples, coverage_taker: CoverageTaker, coverage_estimators,
                    print_zero_coverage_contigs: bool, estimates=None) -> List[ReadsMapped]:
    """contig.rs:13-253 over device results instead of BAM readers.  `estimates`: per sample the n_targets x n_estimators f32
    array Session.estimates() returned (calculate_coverage done on the device) or None."""
    h, k1 = _header(names, target_len)
    sa, k2 = _samples(samples)
    rm = (_ReadsMapped * max(1, len(samples)))()
    L = _lib()
    if estimates is None:
        rc = L.covh_contig_coverage(C.byref(h), sa, C.c_size_t(len(samples)), coverage_taker._h,
                                    _est_array(coverage_estimators), C.c_size_t(len(coverage_estimators)),
                                    C.c_int(int(print_zero_coverage_contigs)), rm)
    else:
        keep = [None if e is None else np.ascontiguousarray(e, np.float32) for e in estimates]
        ptrs = (C.c_void_p * max(1, len(samples)))(*[None if e is None else e.ctypes.data for e in keep])
        rc = L.covh_contig_coverage_estimated(C.byref(h), sa, C.c_size_t(len(samples)), coverage_taker._h,
                                              _est_array(coverage_estimators), C.c_size_t(len(coverage_estimators)),
                                              C.c_int(int(print_zero_coverage_contigs)), rm, ptrs)
    return _finish(rc, rm, len(samples))


def mosdepth_genome_coverage_with_contig_names(names, target_len, samples, genomes: List[str], genome_of_tid,
                                               coverage_taker: CoverageTaker, print_zero_coverage_genomes: bool,
                                               coverage_estimators) -> List[ReadsMapped]:
    """genome.rs:17-322."""
    h, k1 = _header(names, target_len)
    sa, k2 = _samples(samples)
    rm = (_ReadsMapped * max(1, len(samples)))()
    g = np.ascontiguousarray(genome_of_tid, np.int32)
    gn = (C.c_char_p * max(1, len(genomes)))(*[x.encode() for x in genomes])
    rc = _lib().covh_genome_coverage_with_contig_names(
        C.byref(h), sa, C.c_size_t(len(samples)), g.ctypes.data_as(C.c_void_p), gn, C.c_size_t(len(genomes)),
        coverage_taker._h, C.c_int(int(print_zero_coverage_genomes)), _est_array(coverage_estimators),
        C.c_size_t(len(coverage_estimators)), rm)
    return _finish(rc, rm, len(samples))


def mosdepth_genome_coverage(names, target_len, samples, split_char: str, coverage_taker: CoverageTaker,
                             print_zero_coverage_genomes: bool, coverage_estimators, single_genome: bool
                             ) -> List[ReadsMapped]:
    """genome.rs:419-797."""
    h, k1 = _header(names, target_len)
    sa, k2 = _samples(samples)
    rm = (_ReadsMapped * max(1, len(samples)))()
    rc = _lib().covh_genome_coverage_separator(
        C.byref(h), sa, C.c_size_t(len(samples)), C.c_uint8(ord(split_char)), coverage_taker._h,
        C.c_int(int(print_zero_coverage_genomes)), _est_array(coverage_estimators),
        C.c_size_t(len(coverage_estimators)), C.c_int(int(single_genome)), rm)
    return _finish(rc, rm, len(samples))


def print_headers(taker: CoverageTaker, printer: int, entry_type: str, headers: List[str]):
    hs = (C.c_char_p * max(1, len(headers)))(*[x.encode() for x in headers])
    _lib().covh_print_headers(taker._h, C.c_int(printer), entry_type.encode(), hs, C.c_size_t(len(headers)))


def finalise_printing(taker: CoverageTaker, printer: int, entry_type: str, headers: List[str],
                      reads_mapped: Optional[List[ReadsMapped]], columns_to_normalise: List[int],
                      rpkm_column: Optional[int], tpm_column: Optional[int]):
    hs = (C.c_char_p * max(1, len(headers)))(*[x.encode() for x in headers])
    n = len(reads_mapped) if reads_mapped is not None else 0
    rm = (_ReadsMapped * max(1, n))()
    for i in range(n):
        rm[i] = _ReadsMapped(reads_mapped[i].num_mapped_reads, reads_mapped[i].num_reads)
    norm = (C.c_int64 * max(1, len(columns_to_normalise)))(*columns_to_normalise)
    _lib().covh_finalise_printing(taker._h, C.c_int(printer), entry_type.encode(), hs, C.c_size_t(len(headers)),
                                  rm if reads_mapped is not None else None, C.c_size_t(n), norm,
                                  C.c_size_t(len(columns_to_normalise)),
                                  C.c_int64(-1 if rpkm_column is None else rpkm_column),
                                  C.c_int64(-1 if tpm_column is None else tpm_column))


# ---------------------------------------------------------------------------------------------------
# Per-gene coverage (--gff; src/genes.rs) through covh_gene_coverage.
_DEPTH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32))


class _GenomeNamer(C.Structure):   # covh_genome_namer
    _fields_ = [("mode", C.c_int32), ("separator", C.c_uint8), ("genome_of_tid", C.c_void_p),
                ("genome_names", C.POINTER(C.c_char_p))]


class Genes:
    """GeneDefinitions (genes.rs:28-32): parsed from a GFF/GTF file or given explicitly (0-based half-open)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def read_gff(path: str, feature_type: Optional[str] = None) -> "Genes":
        L = _lib()
        L.covh_genes_read_gff.restype = C.c_void_p
        L.covh_genes_read_gff.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        err = C.create_string_buffer(512)
        h = L.covh_genes_read_gff(path.encode(), feature_type.encode() if feature_type is not None else None, err, 512)
        if not h:
            raise IOError(err.value.decode())
        return Genes(h)

    @staticmethod
    def from_list(genes) -> "Genes":
        L = _lib()
        L.covh_genes_from_arrays.restype = C.c_void_p
        n = len(genes)
        ids = (C.c_char_p * max(1, n))(*[g[0].encode() for g in genes])
        ctg = (C.c_char_p * max(1, n))(*[g[1].encode() for g in genes])
        st = np.asarray([g[2] for g in genes], dtype=np.uint64)
        en = np.asarray([g[3] for g in genes], dtype=np.uint64)
        L.covh_genes_from_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        return Genes(L.covh_genes_from_arrays(ids, ctg, st.ctypes.data, en.ctypes.data, n))

    def as_list(self):
        L = _lib()
        L.covh_genes_count.restype = C.c_size_t
        L.covh_genes_count.argtypes = [C.c_void_p]
        L.covh_genes_get.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        out = []
        for i in range(L.covh_genes_count(self._h)):
            a, b, s, e = C.c_char_p(), C.c_char_p(), C.c_uint64(), C.c_uint64()
            L.covh_genes_get(self._h, i, C.byref(a), C.byref(b), C.byref(s), C.byref(e))
            out.append((a.value.decode(), b.value.decode(), int(s.value), int(e.value)))
        return out

    def __del__(self):
        try:
            if self._h:
                L = _lib()
                L.covh_genes_free.argtypes = [C.c_void_p]
                L.covh_genes_free(self._h)
                self._h = None
        except Exception:
            pass


def gene_coverage(names, target_len, genes: Genes, stoit_name: str, records, cfg, depth_of, num_detected_primary: int,
                  coverage_taker: CoverageTaker, coverage_estimators, print_zero_coverage_genes: bool,
                  namer_mode: int = 0, separator: str = "~", genome_of_tid=None, genome_names=None, session=None) -> ReadsMapped:
    """genes.rs:182-344 for one BAM.  `records`: engine.RecordBatch the scan sees; `cfg`: native.CovConfig (flag filter,
    single-read thresholds); `session` (a finished engine.Session holding `records`) selects the device reductions; without
    it `depth_of(tid)` supplies the contig's int32 depth and the reductions run on the host."""
    from .native import CovBatch
    L = _lib()
    h, k1 = _header(names, target_len)
    cb = CovBatch()
    for k in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar"):
        a = getattr(records, k)
        setattr(cb, k, a.ctypes.data if a.size else None)
    cb.n_records = records.n_records
    errs = []

    def _depth(ctx, tid, out):
        try:
            d = np.ascontiguousarray(depth_of(int(tid)), dtype=np.int32)
            C.memmove(out, d.ctypes.data, d.nbytes)
            return 0
        except Exception as e:   # reported by the caller
            errs.append(e)
            return 18
    cbk = _DEPTH_FN(_depth)
    nm = _GenomeNamer()
    nm.mode = namer_mode
    nm.separator = ord(separator) if separator else 0
    keep = None
    if namer_mode == 3:
        g = np.ascontiguousarray(genome_of_tid, dtype=np.int32)
        gn = (C.c_char_p * max(1, len(genome_names)))(*[x.encode() for x in genome_names])
        nm.genome_of_tid = g.ctypes.data
        nm.genome_names = gn
        keep = (g, gn)
    rm = _ReadsMapped()
    L.covh_gene_coverage.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, _DEPTH_FN,
                                     C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    rc = L.covh_gene_coverage(C.byref(h), genes._h, C.byref(nm), stoit_name.encode(), C.byref(cb), C.byref(cfg),
                              session._h if session is not None else None, cbk, None,
                              int(num_detected_primary), coverage_taker._h, _est_array(coverage_estimators),
                              len(coverage_estimators), int(print_zero_coverage_genes), C.byref(rm))
    del keep
    if errs:
        raise errs[0]
    if rc:
        raise HostError(rc, _lib().covh_last_error().decode())
    return ReadsMapped(int(rm.num_mapped_reads), int(rm.num_reads))
