"""TEST INFRASTRUCTURE ONLY — Python face of the CPU oracle (see coverm_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It binds oracle/_build/libcoverm_oracle.so (the literal C restatement of the reference's scan
loops and estimators) and restates the small CPU-only layers around it in Python:

  * pair-mode read filtering               filter.rs:117-228, 281-336
  * CoverageTaker implementations          coverage_takers.rs:74-219, 265-377
  * CoveragePrinter (sparse/dense/MetaBAT) coverage_printer.rs:20-553
  * method-name -> estimator construction  bin/coverm.rs:1315-1504
  * FilterParameters / parse_percentage    bin/coverm.rs:1296-1312, 1648-1704
  * genome definition TSV                  genome_parsing.rs:77-141
  * per-gene coverage (--gff)              genes.rs:42-567

Floats are formatted like Rust's `Display` for f32/f64 (shortest round-trip, positional).
"""
import ctypes as C
import io
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import bamio
from .bamio import BamData

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libcoverm_oracle.so")

MEAN, TRIMMED_MEAN, PILEUP_COUNTS, COVERED_FRACTION, COVERED_BASES, RPKM, TPM, VARIANCE, LENGTH, \
    READ_COUNT, READS_PER_BASE, ANIR = range(12)

ERR_NAMES = {1: "unsorted", 2: "nm_missing", 3: "nm_badtype", 4: "pos_oob", 5: "no_separator", 6: "bad_cigar"}
UNSORTED_MESSAGE = ("BAM file appears to be unsorted. Input BAM files must be sorted by reference "
                    "(i.e. by samtools sort)")  # contig.rs:130-131


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__("oracle error %d (%s)" % (code, ERR_NAMES.get(code, "?")))
        self.code = code
        self.kind = ERR_NAMES.get(code, "?")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "coverm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


class EstParam(C.Structure):
    _fields_ = [("kind", C.c_int32), ("min_fraction_covered_bases", C.c_float),
                ("contig_end_exclusion", C.c_uint64), ("exclude_mismatches", C.c_int32),
                ("trim_min", C.c_float), ("trim_max", C.c_float)]


class _Emit(C.Structure):
    _fields_ = [("type", C.c_int32), ("pad", C.c_int32), ("a", C.c_int64), ("b", C.c_uint64),
                ("cov", C.c_float), ("name_tid", C.c_int32)]


class _Out(C.Structure):
    _fields_ = [("e", C.POINTER(_Emit)), ("n", C.c_size_t), ("cap", C.c_size_t)]


class _Records(C.Structure):
    _fields_ = [("tid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
                ("nm", C.c_void_p), ("nm_kind", C.c_void_p), ("l_seq", C.c_void_p),
                ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("n_records", C.c_uint64),
                ("order", C.c_void_p), ("n_order", C.c_uint64)]


class _FlagFilter(C.Structure):
    _fields_ = [("include_improper_pairs", C.c_int32), ("include_supplementary", C.c_int32),
                ("include_secondary", C.c_int32)]


class _ReadsMapped(C.Structure):
    _fields_ = [("num_mapped_reads", C.c_uint64), ("num_reads", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_estimate_one.restype = C.c_float
        _lib.orc_count_primary.restype = C.c_uint64
        _lib.orc_free.argtypes = [C.c_void_p]
        _lib.orc_free.restype = None
    return _lib


_libm = C.CDLL("libm.so.6")
_libm.logf.restype = C.c_float; _libm.logf.argtypes = [C.c_float]
_libm.expf.restype = C.c_float; _libm.expf.argtypes = [C.c_float]
f32 = np.float32


# ------------------------------------------------------------------ plain data
@dataclass
class FlagFilter:  # lib.rs:60-64
    include_improper_pairs: bool = True
    include_supplementary: bool = True
    include_secondary: bool = False

    def c(self):
        return _FlagFilter(int(self.include_improper_pairs), int(self.include_supplementary),
                           int(self.include_secondary))


@dataclass
class FilterParameters:  # bin/coverm.rs:1648-1657
    flag_filters: FlagFilter
    min_aligned_length_single: int = 0
    min_percent_identity_single: float = 0.0
    min_aligned_percent_single: float = 0.0
    min_mapq: int = 255
    min_aligned_length_pair: int = 0
    min_percent_identity_pair: float = 0.0
    min_aligned_percent_pair: float = 0.0

    def doing_filtering(self) -> bool:  # :1695-1703
        return (self.min_percent_identity_single > 0.0 or self.min_percent_identity_pair > 0.0
                or self.min_aligned_percent_single > 0.0 or self.min_mapq < 255
                or self.min_aligned_percent_pair > 0.0 or self.min_aligned_length_single > 0
                or self.min_aligned_length_pair > 0)


@dataclass
class ReadsMapped:  # lib.rs:54-57
    num_mapped_reads: int
    num_reads: int


def parse_percentage(v: Optional[float]) -> float:
    """bin/coverm.rs:1296-1312: values in [1,100] are percentages."""
    if v is None:
        return 0.0
    p = f32(v)
    if 1.0 <= p <= 100.0:
        p = f32(p / f32(100.0))
    elif not (0.0 <= p <= 100.0):
        raise ValueError("Invalid alignment percentage: '%s'" % v)
    return float(p)


def est_mean(min_frac=0.0, excl=0, exclude_mismatches=False):
    return EstParam(MEAN, min_frac, excl, int(exclude_mismatches), 0, 0)


def est_trimmed_mean(tmin, tmax, min_frac=0.0, excl=0):
    return EstParam(TRIMMED_MEAN, min_frac, excl, 0, tmin, tmax)


def est_pileup_counts(min_frac=0.0, excl=0):
    return EstParam(PILEUP_COUNTS, min_frac, excl, 0, 0, 0)


def est_covered_fraction(min_frac=0.0):
    return EstParam(COVERED_FRACTION, min_frac, 0, 0, 0, 0)


def est_covered_bases(min_frac=0.0):
    return EstParam(COVERED_BASES, min_frac, 0, 0, 0, 0)


def est_rpkm(min_frac=0.0):
    return EstParam(RPKM, min_frac, 0, 0, 0, 0)


def est_tpm(min_frac=0.0):
    return EstParam(TPM, min_frac, 0, 0, 0, 0)


def est_variance(min_frac=0.0, excl=0):
    return EstParam(VARIANCE, min_frac, excl, 0, 0, 0)


def est_length():
    return EstParam(LENGTH, 0, 0, 0, 0, 0)


def est_read_count():
    return EstParam(READ_COUNT, 0, 0, 0, 0, 0)


def est_reads_per_base():
    return EstParam(READS_PER_BASE, 0, 0, 0, 0, 0)


def est_anir():
    return EstParam(ANIR, 0, 0, 0, 0, 0)


COLUMN_HEADERS = {  # estimators.rs:84-105
    MEAN: ["Mean"], TRIMMED_MEAN: ["Trimmed Mean"], PILEUP_COUNTS: ["Coverage", "Bases"],
    COVERED_FRACTION: ["Covered Fraction"], COVERED_BASES: ["Covered Bases"], RPKM: ["RPKM"], TPM: ["TPM"],
    VARIANCE: ["Variance"], LENGTH: ["Length"], READ_COUNT: ["Read Count"],
    READS_PER_BASE: ["Reads per base"], ANIR: ["ANIr"],
}


# ------------------------------------------------------------------ Rust float Display
def fmt_f32(x) -> str:
    x = f32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    if x == 0:
        return "-0" if np.signbit(x) else "0"
    return np.format_float_positional(x, unique=True, trim="-")


def fmt_f64(x) -> str:
    x = np.float64(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    if x == 0:
        return "-0" if np.signbit(x) else "0"
    return np.format_float_positional(x, unique=True, trim="-")


# ------------------------------------------------------------------ takers (coverage_takers.rs)
class StreamingTaker:
    """SingleFloatCoverageStreamingCoveragePrinter, coverage_takers.rs:103-109, 159-167, 212-214."""

    def __init__(self, stream):
        self.stream = stream
        self.stoit = None

    def start_stoit(self, name): self.stoit = name
    def start_entry(self, entry_order_id, entry_name): self.stream.write("%s\t%s" % (self.stoit, entry_name))
    def add_single_coverage(self, cov): self.stream.write("\t0" if cov == 0.0 else "\t" + fmt_f32(cov))
    def add_coverage_entry(self, num_reads, num_bases): raise AssertionError("unreachable")
    def finish_entry(self): self.stream.write("\n")


class PileupTaker:
    """PileupCoverageCoveragePrinter, coverage_takers.rs:110-116, 192-206."""

    def __init__(self, stream):
        self.stream = stream
        self.stoit = None
        self.entry = None

    def start_stoit(self, name): self.stoit = name
    def start_entry(self, entry_order_id, entry_name): self.entry = entry_name
    def add_single_coverage(self, cov): raise AssertionError("unreachable")

    def add_coverage_entry(self, num_reads, num_bases):
        self.stream.write("%s\t%s\t%d\t%d\n" % (self.stoit, self.entry, num_reads, num_bases))

    def finish_entry(self): pass


class CachedTaker:
    """CachedSingleFloatCoverageTaker, coverage_takers.rs:117-153, 171-181, and its iterator :265-377."""

    def __init__(self, num_coverages):
        self.stoit_names = []
        self.entry_names = []
        self.coverages = []  # per stoit: list of (entry_index, coverage)
        self.cur_stoit = None
        self.cur_entry = None
        self.num_coverages = num_coverages

    def start_stoit(self, name):
        self.stoit_names.append(name)
        self.coverages.append([])
        self.cur_stoit = len(self.stoit_names) - 1

    def start_entry(self, entry_order_id, entry_name):
        if entry_order_id >= len(self.entry_names):
            self.entry_names.extend([None] * (entry_order_id + 1 - len(self.entry_names)))
        if self.entry_names[entry_order_id] is None:
            self.entry_names[entry_order_id] = entry_name
        if self.entry_names[entry_order_id] != entry_name:
            raise RuntimeError("Found a difference amongst the reference sets used for mapping.")
        self.cur_entry = entry_order_id

    def add_single_coverage(self, cov): self.coverages[self.cur_stoit].append((self.cur_entry, f32(cov)))
    def add_coverage_entry(self, num_reads, num_bases): raise AssertionError("unreachable")
    def finish_entry(self): pass

    def iterate(self):
        """Yields (entry_index, stoit_index, [coverages]) — coverage_takers.rs:265-377."""
        ns = len(self.stoit_names)
        nc = self.num_coverages
        nxt = [0] * ns
        cur = 0
        last = None
        while cur <= ns:
            lowest = None
            for si, ci in enumerate(nxt):
                if ci < len(self.coverages[si]):
                    ei = self.coverages[si][ci][0]
                    if last is None or ei > last:
                        if lowest is None or ei < lowest:
                            lowest = ei
            if lowest is not None:
                chosen = nxt[cur]
                lst = self.coverages[cur]
                if chosen >= len(lst) or lst[chosen][0] != lowest:
                    ret = (lowest, cur, [f32(0.0)] * nc)
                else:
                    ret = (lowest, cur, [lst[chosen + k][1] for k in range(nc)])
                for si in range(ns):
                    if len(self.coverages[si]) > nxt[si] and self.coverages[si][nxt[si]][0] == lowest:
                        nxt[si] += nc
                last = lowest
                yield ret
            else:
                cur += 1
                if cur >= ns:
                    return
                nxt = [0] * ns
                last = None


# ------------------------------------------------------------------ printers (coverage_printer.rs)
def print_headers(printer: str, entry_type: str, headers: List[str], stream):
    """coverage_printer.rs:123-152.  Returns state the dense printer needs later."""
    if printer in ("streamed", "sparse"):
        stream.write("Sample\t%s" % entry_type + "".join("\t" + h for h in headers) + "\n")


def _lnf(x): return f32(_libm.logf(C.c_float(float(x))))
def _expf(x): return f32(_libm.expf(C.c_float(float(x))))


def print_sparse_cached(taker: CachedTaker, stream, reads_mapped: Optional[List[ReadsMapped]],
                        columns_to_normalise: List[int], rpkm_column: Optional[int], tpm_column: Optional[int]):
    """coverage_printer.rs:155-356 (f32 arithmetic; sparse TPM widened to f64 at :320-323)."""
    nc = taker.num_coverages
    first = next((n for n in taker.entry_names if n is not None), None)
    extra_cols = first.count("\t") if first is not None else 0

    def print_previous_stoit(covs, entries, si):
        mult = [None] * nc
        totals = [None] * nc
        for i in columns_to_normalise:
            t = f32(0.0)
            for cs in covs:
                t = f32(t + cs[i])
            totals[i] = t
            if reads_mapped is not None:
                rm = reads_mapped[si]
                mult[i] = f32(f32(rm.num_mapped_reads) / f32(rm.num_reads))
        if tpm_column is not None:
            t = f32(0.0)
            for cs in covs:
                t = f32(t + cs[tpm_column])
            totals[tpm_column] = t
        stoit = taker.stoit_names[si]
        if columns_to_normalise:
            stream.write("%s\tunmapped" % stoit + "\t" * extra_cols)
            for k, col in enumerate(columns_to_normalise):
                lo = 0 if k == 0 else columns_to_normalise[k - 1] + 1
                stream.write("\tNA" * max(0, col - lo))
                stream.write("\t" + fmt_f32(f32(100.0) * f32(f32(1.0) - mult[col])))
            stream.write("\tNA" * max(0, nc - (columns_to_normalise[-1] + 1)))
            stream.write("\n")
        for ei, cs in zip(entries, covs):
            stream.write("%s\t%s" % (stoit, taker.entry_names[ei].rstrip("\r")))
            for i in range(nc):
                if i in columns_to_normalise:
                    v = f32(f32(f32(cs[i] * f32(100.0)) * mult[i]) / totals[i])
                    stream.write("\t" + fmt_f32(v))
                elif rpkm_column == i:
                    nmr = reads_mapped[si].num_mapped_reads
                    stream.write("\t" + fmt_f32(f32(0.0) if nmr == 0 else f32(cs[i] / f32(nmr))))
                elif tpm_column == i:
                    nmr = reads_mapped[si].num_mapped_reads
                    if nmr == 0:
                        stream.write("\t" + fmt_f64(0.0))
                    else:
                        e = _expf(f32(_lnf(cs[i]) - _lnf(totals[i])))
                        stream.write("\t" + fmt_f64(np.float64(e) * np.float64(10 ** 6)))
                else:
                    stream.write("\t" + fmt_f32(cs[i]))
            stream.write("\n")

    covs, entries, cur = [], [], 0
    for ei, si, cs in taker.iterate():
        if cur != si:
            print_previous_stoit(covs, entries, cur)
            covs, entries, cur = [], [], si
        covs.append(cs)
        entries.append(ei)
    print_previous_stoit(covs, entries, cur)


def print_dense_cached(entry_type: str, headers: List[str], taker: CachedTaker, stream,
                       reads_mapped: Optional[List[ReadsMapped]], columns_to_normalise: List[int],
                       rpkm_column: Optional[int], tpm_column: Optional[int]):
    """coverage_printer.rs:359-553 (dense TPM stays f32, :536-539)."""
    nc = taker.num_coverages
    stream.write(entry_type)
    for s in taker.stoit_names:
        for h in headers:
            stream.write("\t%s %s" % (s, h))
    stream.write("\n")
    mult = [f32(f32(r.num_mapped_reads) / f32(r.num_reads)) for r in reads_mapped] if reads_mapped is not None else []
    if columns_to_normalise:
        stream.write("unmapped" + "\t" * entry_type.count("\t"))
        for si in range(len(taker.stoit_names)):
            for k, col in enumerate(columns_to_normalise):
                lo = 0 if k == 0 else columns_to_normalise[k - 1] + 1
                stream.write("\tNA" * max(0, col - lo))
                stream.write("\t" + fmt_f32(f32(100.0) * f32(f32(1.0) - mult[si])))
            stream.write("\tNA" * max(0, nc - (columns_to_normalise[-1] + 1)))
        stream.write("\n")
    totals = [[None] * nc for _ in taker.stoit_names]
    by_stoit = []
    for ei, si, cs in taker.iterate():
        for i in columns_to_normalise:
            totals[si][i] = cs[i] if totals[si][i] is None else f32(totals[si][i] + cs[i])
        if tpm_column is not None:
            i = tpm_column
            totals[si][i] = cs[i] if totals[si][i] is None else f32(totals[si][i] + cs[i])
        if len(by_stoit) <= si:
            by_stoit.append([])
        by_stoit[si].append((ei, si, cs))
    if not by_stoit:
        return
    for k in range(len(by_stoit[0])):
        stream.write(taker.entry_names[by_stoit[0][k][0]].rstrip("\r"))
        for si, ents in enumerate(by_stoit):
            ei, esi, cs = ents[k]
            for i, cov in enumerate(cs):
                if i in columns_to_normalise:
                    v = f32(f32(f32(cs[i] / totals[esi][i]) * f32(100.0)) * mult[si])
                    stream.write("\t" + fmt_f32(v))
                elif rpkm_column == i:
                    nmr = reads_mapped[si].num_mapped_reads
                    stream.write("\t" + fmt_f32(f32(0.0) if nmr == 0 else f32(cs[i] / f32(nmr))))
                elif tpm_column == i:
                    nmr = reads_mapped[si].num_mapped_reads
                    if nmr == 0:
                        stream.write("\t" + fmt_f32(0.0))
                    else:
                        stream.write("\t" + fmt_f32(f32(_expf(f32(_lnf(cs[i]) - _lnf(totals[esi][i]))) * f32(10 ** 6))))
                else:
                    stream.write("\t" + fmt_f32(cov))
        stream.write("\n")


def print_metabat(taker: CachedTaker, stream):
    """coverage_printer.rs:57-119."""
    stream.write("contigName\tcontigLen\ttotalAvgDepth")
    for s in taker.stoit_names:
        stream.write("\t%s.bam\t%s.bam-var" % (s, s))
    stream.write("\n")
    by_stoit = []
    for ei, si, cs in taker.iterate():
        if len(by_stoit) <= si:
            by_stoit.append([])
        by_stoit[si].append((ei, si, cs))

    def r4(x):  # (x as f64 * 10000.0).round() / 10000.0 ; Rust round = half away from zero
        v = np.float64(x) * 10000.0
        return np.float64(np.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)) / 10000.0

    for k in range(len(by_stoit[0])):
        total = f32(0.0)
        for ents in by_stoit:
            total = f32(total + ents[k][2][1])
        v = np.float64(total) * 10000.0 / np.float64(len(taker.coverages))
        v = np.float64(np.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)) / 10000.0
        stream.write("%s\t%s\t%s" % (taker.entry_names[k], fmt_f32(by_stoit[0][k][2][0]), fmt_f64(v)))
        for ents in by_stoit:
            c = ents[k][2]
            stream.write("\t%s\t%s" % (fmt_f64(r4(c[1])), fmt_f64(r4(c[2]))))
        stream.write("\n")


# ------------------------------------------------------------------ C bridge
def _records(b: BamData, order: Optional[np.ndarray]):
    keep = dict(tid=np.ascontiguousarray(b.tid, np.int32), pos=np.ascontiguousarray(b.pos, np.int32),
                flag=np.ascontiguousarray(b.flag, np.uint16), mapq=np.ascontiguousarray(b.mapq, np.uint8),
                nm=np.ascontiguousarray(b.nm, np.uint32), nm_kind=np.ascontiguousarray(b.nm_kind, np.uint8),
                l_seq=np.ascontiguousarray(b.l_seq.astype(np.uint32)),
                cigar_off=np.ascontiguousarray(b.cigar_off, np.uint32),
                cigar=np.ascontiguousarray(b.cigar, np.uint32))
    if order is not None:
        keep["order"] = np.ascontiguousarray(order, np.uint64)
    r = _Records()
    for k, v in keep.items():
        setattr(r, k, v.ctypes.data if v.size else None)
    r.n_records = b.n_records
    r.n_order = len(order) if order is not None else b.n_records
    return r, keep


def _params(estimators: Sequence[EstParam]):
    arr = (EstParam * max(1, len(estimators)))()
    for i, e in enumerate(estimators):
        arr[i] = e
    return arr


def _collect(out: _Out):
    ems = [(out.e[i].type, out.e[i].a, out.e[i].b, out.e[i].cov, out.e[i].name_tid) for i in range(out.n)]
    lib().orc_out_free(C.byref(out))
    return ems


def _genome_prefix(name: str, sep: str) -> str:
    return name[:name.index(sep)]


def replay(ems, taker, entry_name_of):
    for typ, a, bb, cov, name_tid in ems:
        if typ == 0:
            taker.start_entry(int(a), entry_name_of(name_tid))
        elif typ == 1:
            taker.add_single_coverage(f32(cov))
        elif typ == 2:
            taker.add_coverage_entry(int(a), int(bb))
        else:
            taker.finish_entry()


# ------------------------------------------------------------------ reader stage (bam_generator.rs / filter.rs)
def filter_mode(fp: FilterParameters):
    """filter.rs:48-61 -> (filtering_single, filtering_pairs)."""
    fs0 = fp.min_aligned_length_single > 0 or fp.min_percent_identity_single > 0.0 or fp.min_aligned_percent_single > 0.0
    fp0 = fp.min_aligned_length_pair > 0 or fp.min_percent_identity_pair > 0.0 or fp.min_aligned_percent_pair > 0.0
    fs = fs0 or (not fp0 and fp.min_mapq != 255)
    fpairs = fp0 or ((not fs or not fp.flag_filters.include_improper_pairs) and fp.min_mapq != 255)
    return fs, fpairs


def _aligned_single(b, i):
    c = b.cigar[b.cigar_off[i]:b.cigar_off[i + 1]]
    op = c & 15
    return int((c >> 4)[(op == 0) | (op == 1) | (op == 2) | (op == 7) | (op == 8)].sum()) & 0xFFFFFFFF


def _aligned_pair(b, i):  # filter.rs:301-318: no Del
    c = b.cigar[b.cigar_off[i]:b.cigar_off[i + 1]]
    op = c & 15
    return int((c >> 4)[(op == 0) | (op == 1) | (op == 7) | (op == 8)].sum()) & 0xFFFFFFFF


def _nm(b, i):
    if b.nm_kind[i] == bamio.NM_UNSIGNED:
        return int(b.nm[i])
    raise OracleError(2 if b.nm_kind[i] == bamio.NM_ABSENT else 3)


def _single_passes(b, i, fp: FilterParameters):  # filter.rs:243-279
    if fp.min_mapq != 255 and (b.mapq[i] < fp.min_mapq or b.mapq[i] == 255):
        return False
    edit = _nm(b, i)
    aligned = _aligned_single(b, i)
    with np.errstate(divide="ignore", invalid="ignore"):
        return bool(aligned >= fp.min_aligned_length_single
                    and f32(aligned) / f32(int(b.l_seq[i])) >= f32(fp.min_aligned_percent_single)
                    and f32(1.0) - f32(edit) / f32(aligned) >= f32(fp.min_percent_identity_single))


def _pair_passes(b, i1, i2, fp: FilterParameters):  # filter.rs:281-336
    if fp.min_mapq != 255 and (b.mapq[i1] < fp.min_mapq or b.mapq[i2] < fp.min_mapq
                               or b.mapq[i1] == 255 or b.mapq[i2] == 255):
        return False
    e1, e2 = _nm(b, i1), _nm(b, i2)
    aligned = (_aligned_pair(b, i1) + _aligned_pair(b, i2)) & 0xFFFFFFFF
    with np.errstate(divide="ignore", invalid="ignore"):
        return bool(aligned >= fp.min_aligned_length_pair
                    and f32(aligned) / f32(int(b.l_seq[i1]) + int(b.l_seq[i2])) >= f32(fp.min_aligned_percent_pair)
                    and f32(1.0) - f32(e1 + e2) / f32(aligned) >= f32(fp.min_percent_identity_pair))


def reader_stage(b: BamData, fp: Optional[FilterParameters]):
    """Returns (order or None, num_detected_primary_alignments).

    None filter / not doing_filtering -> BamFileNamedReader (bam_generator.rs:113-119).
    Otherwise FilteredBamReader -> ReferenceSortedBamFilter::read with filter_out=true."""
    prim = int(((b.flag & 0x900) == 0).sum())
    if fp is None or not fp.doing_filtering():
        return None, prim
    fs, fpairs = filter_mode(fp)
    ff = fp.flag_filters
    if fs and not fpairs:  # filter.rs:88-116, in C
        r, keep = _records(b, None)
        order = np.zeros(max(1, b.n_records), dtype=np.uint64)
        n = C.c_uint64(0)
        p = C.c_uint64(0)
        cff = ff.c()
        rc = lib().orc_filter_single(C.byref(r), C.byref(cff), C.c_uint32(fp.min_aligned_length_single),
                                     C.c_float(fp.min_percent_identity_single),
                                     C.c_float(fp.min_aligned_percent_single), C.c_uint8(fp.min_mapq),
                                     order.ctypes.data_as(C.c_void_p), C.byref(n), C.byref(p))
        if rc:
            raise OracleError(rc)
        return order[:n.value].copy(), int(p.value)
    # pair mode, filter.rs:117-228 (filter_out = true)
    order = []
    first_set = {}
    current_reference = -1
    for i in range(b.n_records):
        flag = int(b.flag[i])
        # an unmapped record is not returned at filter.rs:133-135 (filter_out=true) and simply falls
        # through to the tests below, exactly as in the reference
        if flag & 0x100 or flag & 0x800:
            continue
        if not flag & 0x2:
            continue
        if b.tid[i] != current_reference:
            current_reference = int(b.tid[i])
            first_set = {}
        q = b.qname[i]
        if q not in first_set:
            if b.mtid[i] == current_reference:
                first_set[q] = i
        else:
            i1 = first_set.pop(q)
            ok = ((not fs) or (_single_passes(b, i1, fp) and _single_passes(b, i, fp))) and _pair_passes(b, i, i1, fp)
            if ok:
                order.append(i1)
                order.append(i)
    return np.asarray(order, dtype=np.uint64), prim


def reader_filter(b: BamData, fp: FilterParameters, filter_out: bool = True):
    """ReferenceSortedBamFilter::read as a whole (filter.rs:84-228), record by record, for both values of filter_out: the
    selection of `coverm filter` (bin/coverm.rs:408-472; filter_out = not --inverse).  Always constructed there, whether or
    not any threshold is set (with none, filter.rs:48-61 selects the pair branch).  Returns the indices in return order."""
    fs, fpairs = filter_mode(fp)
    ff = fp.flag_filters
    order = []
    if fs and not fpairs:                                   # :88-116
        for i in range(b.n_records):
            flag = int(b.flag[i])
            unmapped, supp, sec = bool(flag & 0x4), bool(flag & 0x800), bool(flag & 0x100)
            if unmapped and not filter_out:                  # :97-99
                order.append(i)
                continue
            passes1 = (not unmapped) and (ff.include_supplementary or not supp) and (ff.include_secondary or not sec)
            if passes1 and _single_passes(b, i, fp) == filter_out:      # :103-113
                order.append(i)
        return np.asarray(order, dtype=np.uint64)
    first_set = {}
    current_reference = -1                                  # :76
    for i in range(b.n_records):
        flag = int(b.flag[i])
        if flag & 0x4 and not filter_out:                    # :133-135
            order.append(i)
            continue
        if flag & 0x100 or flag & 0x800:                     # :138-140
            continue
        if not flag & 0x2:                                   # :141-147
            if not filter_out:
                order.append(i)
            continue
        if b.tid[i] != current_reference:                    # :150-162
            current_reference = int(b.tid[i])
            first_set = {}
        q = b.qname[i]
        if q not in first_set:                               # :168-184
            if b.mtid[i] == current_reference:
                first_set[q] = i
        else:
            i1 = first_set.pop(q)
            ok = ((not fs) or (_single_passes(b, i1, fp) and _single_passes(b, i, fp))) and _pair_passes(b, i, i1, fp)
            if ok == filter_out:                             # :212-220
                order.append(i1)
                order.append(i)
    return np.asarray(order, dtype=np.uint64)


# ------------------------------------------------------------------ scan entry points
def contig_coverage(bams: Sequence[BamData], stoit_names: Sequence[str], taker, estimators: Sequence[EstParam],
                    print_zero_coverage_contigs: bool, flag_filters: FlagFilter,
                    filter_params: Optional[FilterParameters] = None) -> List[ReadsMapped]:
    """contig.rs:13-253."""
    out_rm = []
    for b, name in zip(bams, stoit_names):
        order, prim = reader_stage(b, filter_params)
        taker.start_stoit(name)
        r, keep = _records(b, order)
        tl = np.ascontiguousarray(b.ref_lens, np.int64)
        out = _Out()
        rm = _ReadsMapped()
        ff = flag_filters.c()
        rc = lib().orc_contig_coverage(C.byref(r), tl.ctypes.data_as(C.c_void_p), C.c_int32(len(tl)),
                                       _params(estimators), C.c_int32(len(estimators)),
                                       C.c_int32(int(print_zero_coverage_contigs)), C.byref(ff),
                                       C.c_uint64(prim), C.byref(out), C.byref(rm))
        ems = _collect(out)
        if rc:
            raise OracleError(rc)
        replay(ems, taker, lambda t: b.ref_names[t])
        out_rm.append(ReadsMapped(int(rm.num_mapped_reads), int(rm.num_reads)))
    return out_rm


def genome_coverage_with_contig_names(bams, stoit_names, genomes: List[str], contig_to_genome: dict, taker,
                                      print_zero_coverage_genomes: bool, flag_filters: FlagFilter,
                                      estimators: Sequence[EstParam],
                                      filter_params: Optional[FilterParameters] = None) -> List[ReadsMapped]:
    """genome.rs:17-322."""
    out_rm = []
    for b, name in zip(bams, stoit_names):
        order, prim = reader_stage(b, filter_params)
        taker.start_stoit(name)
        r, keep = _records(b, order)
        tl = np.ascontiguousarray(b.ref_lens, np.int64)
        g_of = np.asarray([contig_to_genome.get(n, -1) for n in b.ref_names], dtype=np.int32)
        if (g_of >= 0).sum() == 0:
            raise RuntimeError("Error: There are no found reference sequences that are a part of a genome")
        out = _Out()
        rm = _ReadsMapped()
        ff = flag_filters.c()
        rc = lib().orc_genome_coverage_with_contig_names(
            C.byref(r), tl.ctypes.data_as(C.c_void_p), C.c_int32(len(tl)), g_of.ctypes.data_as(C.c_void_p),
            C.c_int32(len(genomes)), _params(estimators), C.c_int32(len(estimators)),
            C.c_int32(int(print_zero_coverage_genomes)), C.byref(ff), C.c_uint64(prim), C.byref(out), C.byref(rm))
        ems = _collect(out)
        if rc:
            raise OracleError(rc)
        replay(ems, taker, lambda t: genomes[-1 - t])
        out_rm.append(ReadsMapped(int(rm.num_mapped_reads), int(rm.num_reads)))
    return out_rm


def genome_coverage_separator(bams, stoit_names, split_char: str, taker, print_zero_coverage_genomes: bool,
                              estimators: Sequence[EstParam], flag_filters: FlagFilter, single_genome: bool,
                              filter_params: Optional[FilterParameters] = None) -> List[ReadsMapped]:
    """genome.rs:419-797."""
    out_rm = []
    for b, name in zip(bams, stoit_names):
        order, prim = reader_stage(b, filter_params)
        taker.start_stoit(name)
        r, keep = _records(b, order)
        tl = np.ascontiguousarray(b.ref_lens, np.int64)
        names = "".join(b.ref_names).encode()
        off = np.zeros(len(b.ref_names) + 1, dtype=np.uint32)
        np.cumsum([len(n.encode()) for n in b.ref_names], out=off[1:])
        out = _Out()
        rm = _ReadsMapped()
        ff = flag_filters.c()
        rc = lib().orc_genome_coverage_separator(
            C.byref(r), C.c_char_p(names), off.ctypes.data_as(C.c_void_p), tl.ctypes.data_as(C.c_void_p),
            C.c_int32(len(tl)), C.c_uint8(ord(split_char)), C.c_int32(int(single_genome)), _params(estimators),
            C.c_int32(len(estimators)), C.c_int32(int(print_zero_coverage_genomes)), C.byref(ff),
            C.c_uint64(prim), C.byref(out), C.byref(rm))
        ems = _collect(out)
        if rc:
            raise OracleError(rc)

        def nm(t, b=b):
            return "genome1" if t == -1000000 else _genome_prefix(b.ref_names[t], split_char)
        replay(ems, taker, nm)
        out_rm.append(ReadsMapped(int(rm.num_mapped_reads), int(rm.num_reads)))
    return out_rm


def contig_deltas(b: BamData, flag_filters: FlagFilter, tid: int, order=None) -> np.ndarray:
    """ups_and_downs of one contig exactly as contig.rs:144-202 leaves it."""
    r, keep = _records(b, order)
    tl = np.ascontiguousarray(b.ref_lens, np.int64)
    ud = np.zeros(int(tl[tid]), dtype=np.int32)
    ff = flag_filters.c()
    rc = lib().orc_contig_deltas(C.byref(r), tl.ctypes.data_as(C.c_void_p), C.byref(ff), C.c_int32(tid),
                                 ud.ctypes.data_as(C.c_void_p))
    if rc:
        raise OracleError(rc)
    return ud


ORC_STATS_DTYPE = np.dtype([
    ("n_primary", "<u8"), ("n_pass", "<u8"), ("n_nonsupp", "<u8"), ("sum_nm", "<u8"), ("sum_indel", "<u8"),
    ("id_primary", "<f8"), ("id_nonsupp", "<f8"), ("win_sum_d", "<u8"), ("win_sum_d2", "<u8"),
    ("win_covered", "<u8"), ("full_covered", "<u8"), ("first_record", "<u8"), ("last_record", "<u8"),
    ("win_min_d", "<u4"), ("win_max_d", "<u4"), ("hist_len", "<u4"), ("seen", "<u4"), ("hist_off", "<u8")])


def integer_stats(b: BamData, flag_filters: FlagFilter, filter_params: Optional[FilterParameters], excl: int,
                  mask: Optional[np.ndarray] = None):
    """Per-contig integer sufficient statistics computed the reference's way (orc_integer_stats).
    Returns (stats[n_targets] as ORC_STATS_DTYPE, hist uint64[], num_detected_primary_alignments)."""
    order, prim = reader_stage(b, filter_params)
    r, keep = _records(b, order)
    tl = np.ascontiguousarray(b.ref_lens, np.int64)
    out = np.zeros(max(1, len(tl)), dtype=ORC_STATS_DTYPE)
    hist_p = C.c_void_p()
    hn = C.c_uint64(0)
    ff = flag_filters.c()
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    rc = lib().orc_integer_stats(C.byref(r), tl.ctypes.data_as(C.c_void_p), C.c_int32(len(tl)),
                                 m.ctypes.data_as(C.c_void_p) if m is not None else None, C.byref(ff),
                                 C.c_uint64(excl), out.ctypes.data_as(C.c_void_p), C.byref(hist_p), C.byref(hn))
    hist = np.zeros(hn.value, dtype=np.uint64)
    if hn.value:
        C.memmove(hist.ctypes.data, hist_p, hn.value * 8)
    lib().orc_free(hist_p)
    if rc:
        raise OracleError(rc)
    return out[:len(tl)], hist, prim


def estimate_one(p: EstParam, ud: np.ndarray, n_reads=0, mismatches=0, sum_identity=0.0, unobserved=(0,)):
    ud = np.ascontiguousarray(ud, np.int32)
    un = np.asarray(unobserved, dtype=np.uint64)
    return float(lib().orc_estimate_one(C.byref(p), ud.ctypes.data_as(C.c_void_p), C.c_uint64(len(ud)),
                                        C.c_uint64(n_reads), C.c_uint64(mismatches), C.c_double(sum_identity),
                                        un.ctypes.data_as(C.c_void_p), C.c_uint64(len(un))))


def estimate_emit(p: EstParam, ud: np.ndarray, n_reads=0, mismatches=0, sum_identity=0.0, unobserved=(0,)):
    """add_contig + calculate_coverage + print_coverage on one delta array: (coverage, emitted taker calls)."""
    ud = np.ascontiguousarray(ud, np.int32)
    un = np.asarray(unobserved, dtype=np.uint64)
    out = _Out()
    L = lib()
    L.orc_estimate_emit.restype = C.c_float
    c = float(L.orc_estimate_emit(C.byref(p), ud.ctypes.data_as(C.c_void_p), C.c_uint64(len(ud)), C.c_uint64(n_reads),
                                  C.c_uint64(mismatches), C.c_double(sum_identity), un.ctypes.data_as(C.c_void_p),
                                  C.c_uint64(len(un)), C.byref(out)))
    return c, _collect(out)


# ------------------------------------------------------------------ per-gene coverage (src/genes.rs)
import re as _re

_GFF_KEYS = ("ID", "locus_tag", "gene_id", "Name", "gene", "Parent")


def _gff_attribute(attributes: str, key: str):  # genes.rs:144-161
    for entry in attributes.split(";"):
        entry = entry.strip()
        if not entry:
            continue
        if entry.startswith(key + "="):
            return entry[len(key) + 1:].strip()
        if entry.startswith(key + " "):
            return entry[len(key) + 1:].strip().strip('"')
    return None


def read_gff(path: str, feature_type: Optional[str] = None):
    """GeneDefinitions::read_gff, genes.rs:42-126 -> list of (id, contig, start0, end0_exclusive)."""
    genes, auto_id = [], 0
    with open(path) as fh:
        for line in fh.read().split("\n"):
            trimmed = line.rstrip()
            if not trimmed or trimmed.startswith("#"):
                continue
            f = trimmed.split("\t")
            if len(f) < 8:
                continue
            if feature_type is not None and f[2] != feature_type:
                continue
            if not _re.fullmatch(r"\+?[0-9]+", f[3]) or not _re.fullmatch(r"\+?[0-9]+", f[4]):
                continue
            s1, e1 = int(f[3]), int(f[4])
            if s1 >= 1 << 64 or e1 >= 1 << 64 or s1 == 0 or e1 < s1:
                continue
            attrs = f[8] if len(f) > 8 else ""
            gid = None
            for k in _GFF_KEYS:
                v = _gff_attribute(attrs, k)
                if v:
                    gid = v
                    break
            if gid is None:
                auto_id += 1
                gid = "%s_gene_%d" % (f[0], auto_id)
            genes.append((gid, f[0], s1 - 1, e1))
    return genes


def resolve_genes(genes, ref_names, ref_lens, genome_namer=None):
    """resolve_genes_against_header, genes.rs:346-419 -> per-tid lists of [entry_id, display name, start, end]."""
    name_to_tid = {}
    for tid, n in enumerate(ref_names):
        name_to_tid[n] = tid                      # HashMap::insert: a repeated name keeps the last tid
    by_tid = [[] for _ in ref_names]
    for gid, contig, start, end in genes:
        tid = name_to_tid.get(contig)
        if tid is None:
            continue
        L = int(ref_lens[tid])
        s, e = min(start, L), min(end, L)
        if s >= e:
            continue
        if genome_namer is not None:
            g = genome_namer(contig)
            if g is None:
                continue
            name = "%s\t%s\t%s" % (gid, contig, g)
        else:
            name = "%s\t%s" % (gid, contig)
        by_tid[tid].append([0, name, s, e])
    nxt = 0
    for lst in by_tid:
        lst.sort(key=lambda g: g[2])               # stable, like sort_by_key
        for g in lst:
            g[0] = nxt
            nxt += 1
    return by_tid


def gene_coverage(bams: Sequence[BamData], stoit_names: Sequence[str], taker, estimators: Sequence[EstParam], genes,
                  genome_namer, print_zero_coverage_genes: bool, flag_filters: FlagFilter,
                  filter_params: Optional[FilterParameters] = None) -> List[ReadsMapped]:
    """gene_coverage, genes.rs:182-344 (+ emit_genes_for_contig :462-552).  Depth deltas of a contig come from the C
    restatement of the contig scan (same CIGAR walk, :258-290); per-gene estimator values from orc_estimate_one fed with
    the gene's delta array exactly as :508-535 builds it."""
    out_rm = []
    for b, name in zip(bams, stoit_names):
        order, prim = reader_stage(b, filter_params)
        taker.start_stoit(name)
        by_tid = resolve_genes(genes, b.ref_names, b.ref_lens, genome_namer)
        idx = np.arange(b.n_records) if order is None else np.asarray(order, dtype=np.int64)

        def zero_genes(tid):
            for eid, gname, s, e in by_tid[tid]:
                taker.start_entry(eid, gname)
                for p in estimators:
                    taker.add_single_coverage(float(e - s) if p.kind == 8 else 0.0)   # print_zero_coverage, estimators.rs:971-991
                taker.finish_entry()

        def emit(tid, starts, is_prim, mism, ident):
            glist = by_tid[tid]
            if not glist:
                return
            ud = contig_deltas(b, flag_filters, tid, order)
            L = len(ud)
            cov = np.cumsum(ud, dtype=np.int32)
            st = np.asarray(starts, dtype=np.uint64)
            pp = np.concatenate([[0], np.cumsum(np.asarray(is_prim, dtype=np.uint64), dtype=np.uint64)])
            pm = np.concatenate([[0], np.cumsum(np.asarray(mism, dtype=np.uint64), dtype=np.uint64)])
            pi = np.concatenate([[0.0], np.add.accumulate(np.asarray(ident, dtype=np.float64))]) if len(ident) else np.zeros(1)
            for eid, gname, s, e in glist:
                e = min(e, L)
                if s >= e:
                    continue
                gud = np.empty(e - s, dtype=np.int32)
                gud[0] = cov[s]
                gud[1:] = ud[s + 1:e]
                lo = int(np.searchsorted(st, s, side="left")); hi = int(np.searchsorted(st, e, side="left"))
                n_reads = int(pp[hi] - pp[lo]); mm = int(pm[hi] - pm[lo]); sid = float(pi[hi] - pi[lo])
                res = [estimate_emit(p, gud, n_reads, mm, sid) for p in estimators]     # (coverage, emits): genes.rs:536-549
                if print_zero_coverage_genes or any(c > 0.0 for c, _ in res):
                    taker.start_entry(eid, gname)
                    for c, ems in res:
                        for typ, a, bb, cv, _t in ems:
                            if typ == 1:
                                taker.add_single_coverage(f32(cv))
                            elif typ == 2:
                                taker.add_coverage_entry(int(a), int(bb))
                    taker.finish_entry()

        last_tid = -2
        starts, is_prim, mism, ident = [], [], [], []
        mapped_total = 0
        ff = flag_filters

        def previous(last, cur):
            if last != -2:
                emit(last, starts, is_prim, mism, ident)
            if print_zero_coverage_genes:
                t = 0 if last == -2 else last + 1
                while t < cur:
                    zero_genes(t)
                    t += 1

        for i in idx:
            flag = int(b.flag[i])
            if (not ff.include_secondary and flag & 0x100) or (not ff.include_supplementary and flag & 0x800) \
                    or (not ff.include_improper_pairs and not flag & 0x2):
                continue
            if flag & 0x4:
                continue
            tid = int(b.tid[i])
            if tid != last_tid:
                if tid < last_tid:
                    raise OracleError(1)
                previous(last_tid, tid)
                last_tid = tid
                starts, is_prim, mism, ident = [], [], [], []
            primary = not (flag & 0x900)
            mapped_total += primary
            c = b.cigar[b.cigar_off[i]:b.cigar_off[i + 1]]
            op, ln = c & 15, (c >> 4).astype(np.int64)
            aligned = int(ln[(op == 0) | (op == 7) | (op == 8) | (op == 2) | (op == 1)].sum())
            indels = int(ln[(op == 2) | (op == 1)].sum())
            if b.nm_kind[i] != 1:
                raise OracleError(2 if b.nm_kind[i] == 0 else 3)
            edit = int(b.nm[i])
            starts.append(int(b.pos[i])); is_prim.append(1 if primary else 0)
            mism.append(max(0, edit - indels))                                    # saturating_sub, :296
            ident.append((float(aligned) - float(edit)) / float(aligned) if primary and aligned > 0 else 0.0)
        previous(last_tid, len(b.ref_names))
        out_rm.append(ReadsMapped(int(mapped_total), int(prim)))
    return out_rm


# ------------------------------------------------------------------ CLI-level driver (bin/coverm.rs)
def read_genome_definition(path: str):
    """read_genome_definition_file, genome_parsing.rs:71-141: `genome<TAB>contig [comment]` lines; the contig is the first
    whitespace-separated token of the second column, the genome name is trimmed, genomes keep file order, a contig given
    to two genomes or a line without exactly one tab (blank lines included) is fatal.  Returns (genomes, contig->index)."""
    genomes, idx, c2g = [], {}, {}
    with open(path, newline="") as fh:
        text = fh.read()
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()                                   # BufRead::lines: no empty item after the final newline
    for line in lines:
        if line.endswith("\r"):
            line = line[:-1]
        f = line.split("\t")
        if len(f) != 2:
            raise ValueError('The line "%s" in the genome definition file is not a genome name and contig name separated by a tab' % line)
        g = f[0].strip()
        toks = f[1].split()
        if not toks:
            raise ValueError("Failed to split contig name by whitespace in genome definition file")
        c = toks[0]
        if c in c2g and genomes[c2g[c]] != g:
            raise ValueError("The contig name '%s' was assigned to multiple genomes" % c)
        if g not in idx:
            idx[g] = len(genomes)
            genomes.append(g)
        c2g.setdefault(c, idx[g])
    return genomes, c2g


def estimators_and_taker(methods: Sequence[str], min_covered_fraction: float, contig_end_exclusion: int,
                         trim_min: float, trim_max: float, output_format: str, stream):
    """bin/coverm.rs:1315-1504.  Returns dict(estimators, taker, printer, columns_to_normalise, rpkm, tpm)."""
    mcf = parse_percentage(min_covered_fraction)
    est, norm, rpkm, tpm = [], [], None, None
    if list(methods) == ["metabat"]:
        est = [est_length(), est_mean(mcf, contig_end_exclusion, False), est_variance(mcf, contig_end_exclusion)]
        return dict(estimators=est, taker=CachedTaker(3), printer="metabat", columns_to_normalise=[], rpkm=None,
                    tpm=None)
    for i, m in enumerate(methods):
        if m == "mean": est.append(est_mean(mcf, contig_end_exclusion, False))
        elif m == "coverage_histogram": est.append(est_pileup_counts(mcf, contig_end_exclusion))
        elif m == "trimmed_mean":
            est.append(est_trimmed_mean(parse_percentage(trim_min), parse_percentage(trim_max), mcf,
                                        contig_end_exclusion))
        elif m == "covered_fraction": est.append(est_covered_fraction(mcf))
        elif m == "covered_bases": est.append(est_covered_bases(mcf))
        elif m == "rpkm": rpkm = i; est.append(est_rpkm(mcf))
        elif m == "tpm": tpm = i; est.append(est_tpm(mcf))
        elif m == "variance": est.append(est_variance(mcf, contig_end_exclusion))
        elif m == "length": est.append(est_length())
        elif m == "relative_abundance": norm.append(i); est.append(est_mean(mcf, contig_end_exclusion, False))
        elif m == "count": est.append(est_read_count())
        elif m == "reads_per_base": est.append(est_reads_per_base())
        elif m == "anir": est.append(est_anir())
        else: raise ValueError(m)
    if "coverage_histogram" in methods:
        taker, printer = PileupTaker(stream), "streamed"
    elif not norm and rpkm is None and tpm is None and output_format == "sparse":
        taker, printer = StreamingTaker(stream), "streamed"
    else:
        taker, printer = CachedTaker(len(est)), output_format
    return dict(estimators=est, taker=taker, printer=printer, columns_to_normalise=norm, rpkm=rpkm, tpm=tpm)


def run_cli(mode: str, bam_paths: Sequence[str], methods: Sequence[str] = None, min_covered_fraction=None,
            contig_end_exclusion: int = 75, trim_min=5, trim_max=95, output_format: str = None,
            no_zeros: bool = False, proper_pairs_only=False, exclude_supplementary=False,
            include_secondary=False, min_read_aligned_length=0, min_read_percent_identity=None,
            min_read_aligned_percent=None, min_mapq=255, min_read_aligned_length_pair=0,
            min_read_percent_identity_pair=None, min_read_aligned_percent_pair=None,
            separator: Optional[str] = None, single_genome=False, genome_definition: Optional[str] = None,
            bams: Optional[Sequence[BamData]] = None, gff: Optional[str] = None, gff_feature_type: Optional[str] = None) -> str:
    """`coverm contig|genome --bam-files ...` restated end to end (bin/coverm.rs:56-407, 473-663)."""
    if methods is None:
        methods = ["mean"] if mode == "contig" else ["relative_abundance"]
    if min_covered_fraction is None:
        min_covered_fraction = 0 if mode == "contig" else 10
    if output_format is None:
        output_format = "dense"
    stream = io.StringIO()
    et = estimators_and_taker(methods, min_covered_fraction, contig_end_exclusion, trim_min, trim_max,
                              output_format, stream)
    fp = FilterParameters(FlagFilter(not proper_pairs_only, not exclude_supplementary, include_secondary),
                          min_read_aligned_length, parse_percentage(min_read_percent_identity),
                          parse_percentage(min_read_aligned_percent), min_mapq, min_read_aligned_length_pair,
                          parse_percentage(min_read_percent_identity_pair),
                          parse_percentage(min_read_aligned_percent_pair))
    if list(methods) == ["metabat"]:  # :1680-1693
        fp.min_percent_identity_single = float(f32(0.97001))
        fp.flag_filters = FlagFilter(True, True, True)
    headers = [h for e in et["estimators"] for h in COLUMN_HEADERS[e.kind]]
    for i in et["columns_to_normalise"]:
        headers[i] = "Relative Abundance (%)"
    entry_type = "Contig" if mode == "contig" else "Genome"
    if gff is not None:   # coverm.rs:511-518, 1557-1590
        entry_type = "Gene\tContig" if mode == "contig" else "Gene\tContig\tGenome"
    print_headers(et["printer"], entry_type, headers, stream)
    if bams is None:
        bams = [bamio.read_alignment_file(p) for p in bam_paths]
    stoits = [os.path.splitext(os.path.basename(p))[0] for p in bam_paths]  # bam_generator.rs:358-365
    if gff is not None:
        genes = read_gff(gff, gff_feature_type)
        namer = None
        if mode == "genome":
            if single_genome:
                namer = lambda c: "genome1"
            elif separator is not None:
                namer = lambda c: c.split(separator, 1)[0] if separator in c else None
            else:
                _g, c2g = read_genome_definition(genome_definition)
                namer = lambda c: _g[c2g[c]] if c in c2g else None
        rms = gene_coverage(bams, stoits, et["taker"], et["estimators"], genes, namer, not no_zeros, fp.flag_filters, fp)
    elif mode == "contig":
        rms = contig_coverage(bams, stoits, et["taker"], et["estimators"], not no_zeros, fp.flag_filters, fp)
    elif separator is not None or single_genome:
        rms = genome_coverage_separator(bams, stoits, "0" if single_genome else separator, et["taker"],
                                        not no_zeros, et["estimators"], fp.flag_filters, single_genome, fp)
    else:
        genomes, c2g = read_genome_definition(genome_definition)
        rms = genome_coverage_with_contig_names(bams, stoits, genomes, c2g, et["taker"], not no_zeros,
                                                fp.flag_filters, et["estimators"], fp)
    if et["printer"] == "sparse":
        print_sparse_cached(et["taker"], stream, rms, et["columns_to_normalise"], et["rpkm"], et["tpm"])
    elif et["printer"] == "dense":
        print_dense_cached(entry_type, headers, et["taker"], stream, rms, et["columns_to_normalise"], et["rpkm"],
                           et["tpm"])
    elif et["printer"] == "metabat":
        print_metabat(et["taker"], stream)
    return stream.getvalue()
