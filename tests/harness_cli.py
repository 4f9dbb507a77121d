"""TEST HARNESS (not part of the product: the orchestrator is csrc/host_cli.cpp, the `coverm-amd` binary).
`coverm contig` / `coverm genome` over --bam-files in Python, driving the product's C++ host layer (and, with a GPU, its sessions) from
in-memory records, so that the CPU-only suite can run the host layer on the oracle's statistics and tests/dist_worker.py can shard it.

Mirrors the reference orchestrator for this path only (src/bin/coverm.rs): FilterParameters (:1648-1704),
EstimatorsAndTaker::generate_from_clap (:1315-1504), run_contig (:2088-2131), run_genome (:1539-1628),
parse_percentage (:1296-1312), parse_separator (:1522-1537).  Read mapping, index building, dereplication
and the other subcommands are out of scope.

Records reach the GPU through coverm_amd.engine.Session (C ABI); the per-sample statistics come back and
the C++ host layer (coverm_amd.host) turns them into the reference's exact output text.
"""
import argparse
import os
import sys
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import ctypes as C

import numpy as np

from coverm_amd import host
from coverm_amd.bam import AlignmentFile  # noqa: F401  (re-exported: the tests take it from here)
from coverm_amd.engine import FilterConfig, RecordBatch, Session
from coverm_amd.host import CoverageEstimator, CoverageTaker, SampleResult

CONCATENATED_FASTA_FILE_SEPARATOR = "~"  # lib.rs:46
UNSORTED_MESSAGE = ("BAM file appears to be unsorted. Input BAM files must be sorted by reference "
                    "(i.e. by samtools sort)")


@dataclass
class FlagFilter:  # lib.rs:60-64
    include_improper_pairs: bool = True
    include_supplementary: bool = True
    include_secondary: bool = False


@dataclass
class FilterParameters:  # bin/coverm.rs:1648-1657
    flag_filters: FlagFilter = field(default_factory=FlagFilter)
    min_aligned_length_single: int = 0
    min_percent_identity_single: float = 0.0
    min_aligned_percent_single: float = 0.0
    min_mapq: int = 255
    min_aligned_length_pair: int = 0
    min_percent_identity_pair: float = 0.0
    min_aligned_percent_pair: float = 0.0

    def doing_filtering(self):  # :1695-1703
        return (self.min_percent_identity_single > 0.0 or self.min_percent_identity_pair > 0.0
                or self.min_aligned_percent_single > 0.0 or self.min_mapq < 255
                or self.min_aligned_percent_pair > 0.0 or self.min_aligned_length_single > 0
                or self.min_aligned_length_pair > 0)

    def filter_mode(self):
        """ReferenceSortedBamFilter::new mode selection, filter.rs:48-61 -> (filtering_single, filtering_pairs)."""
        fs0 = (self.min_aligned_length_single > 0 or self.min_percent_identity_single > 0.0
               or self.min_aligned_percent_single > 0.0)
        fp0 = (self.min_aligned_length_pair > 0 or self.min_percent_identity_pair > 0.0
               or self.min_aligned_percent_pair > 0.0)
        fs = fs0 or (not fp0 and self.min_mapq != 255)
        fp = fp0 or ((not fs or not self.flag_filters.include_improper_pairs) and self.min_mapq != 255)
        return fs, fp


def parse_percentage(v) -> float:
    """bin/coverm.rs:1296-1312: f32; values in [1,100] are divided by 100."""
    if v is None:
        return 0.0
    p = np.float32(v)
    if 1.0 <= p <= 100.0:
        p = np.float32(p / np.float32(100.0))
    elif not (0.0 <= p <= 100.0):
        raise SystemExit("Invalid alignment percentage: '%s'" % v)
    return float(p)


@dataclass
class EstimatorsAndTaker:  # bin/coverm.rs:1315-1504
    estimators: List[CoverageEstimator]
    taker: CoverageTaker
    printer: int
    columns_to_normalise: List[int]
    rpkm_column: Optional[int]
    tpm_column: Optional[int]

    @staticmethod
    def generate(methods: Sequence[str], min_covered_fraction, contig_end_exclusion: int, trim_min, trim_max,
                 output_format: str) -> "EstimatorsAndTaker":
        E = CoverageEstimator
        mcf = parse_percentage(min_covered_fraction)
        est, norm, rpkm, tpm = [], [], None, None
        if "metabat" in methods:
            if len(methods) > 1:
                raise SystemExit("Cannot specify the metabat method with any other coverage methods")
            est = [E.new_estimator_length(), E.new_estimator_mean(mcf, contig_end_exclusion, False),
                   E.new_estimator_variance(mcf, contig_end_exclusion)]
            return EstimatorsAndTaker(est, CoverageTaker.new_cached_single_float_coverage_taker(3),
                                      host.PRINTER_METABAT, [], None, None)
        for i, m in enumerate(methods):
            if m == "mean": est.append(E.new_estimator_mean(mcf, contig_end_exclusion, False))
            elif m == "coverage_histogram": est.append(E.new_estimator_pileup_counts(mcf, contig_end_exclusion))
            elif m == "trimmed_mean":
                est.append(E.new_estimator_trimmed_mean(parse_percentage(trim_min), parse_percentage(trim_max), mcf,
                                                        contig_end_exclusion))
            elif m == "covered_fraction": est.append(E.new_estimator_covered_fraction(mcf))
            elif m == "covered_bases": est.append(E.new_estimator_covered_bases(mcf))
            elif m == "rpkm":
                if rpkm is not None:
                    raise SystemExit("The RPKM column cannot be specified more than once")
                rpkm = i; est.append(E.new_estimator_rpkm(mcf))
            elif m == "tpm":
                if tpm is not None:
                    raise SystemExit("The TPM column cannot be specified more than once")
                tpm = i; est.append(E.new_estimator_tpm(mcf))
            elif m == "variance": est.append(E.new_estimator_variance(mcf, contig_end_exclusion))
            elif m == "length": est.append(E.new_estimator_length())
            elif m == "relative_abundance":
                norm.append(i); est.append(E.new_estimator_mean(mcf, contig_end_exclusion, False))
            elif m == "count": est.append(E.new_estimator_read_count())
            elif m == "reads_per_base": est.append(E.new_estimator_reads_per_base())
            elif m == "anir": est.append(E.new_estimator_anir())
            else: raise SystemExit("unknown method %r" % m)
        if "coverage_histogram" in methods:
            if len(methods) > 1:
                raise SystemExit("Cannot specify the coverage_histogram method with any other coverage methods")
            taker, printer = CoverageTaker.new_pileup_coverage_coverage_printer(), host.PRINTER_STREAMED
        elif not norm and rpkm is None and tpm is None and output_format == "sparse":
            taker, printer = CoverageTaker.new_single_float_coverage_streaming_coverage_printer(), host.PRINTER_STREAMED
        else:
            taker = CoverageTaker.new_cached_single_float_coverage_taker(len(est))
            printer = host.PRINTER_SPARSE if output_format == "sparse" else host.PRINTER_DENSE
        if mcf != 0.0:  # :1472-1494
            bad = {host.READ_COUNT: "counts", host.LENGTH: "length", host.READS_PER_BASE: "reads_per_base",
                   host.ANIR: "anir"}
            for e in est:
                if e.kind in bad:
                    raise SystemExit("The '%s' coverage estimator cannot be used when --min-covered-fraction is > 0"
                                     % bad[e.kind])
        return EstimatorsAndTaker(est, taker, printer, norm, rpkm, tpm)

    def headers(self):
        hs = [h for e in self.estimators for h in e.column_headers()]
        for i in self.columns_to_normalise:
            hs[i] = "Relative Abundance (%)"
        return hs


# ---------------------------------------------------------------------------------------------------
# Reader stage.  Single-read thresholds run on the GPU (k_prep); mate pairing needs read names, which do not cross
# the C ABI, so it is the host layer's threaded C++ pre-pass (csrc/host_filter.cpp, filter.rs:117-228).
class _PairFilter(C.Structure):   # covh_pair_filter
    _fields_ = [("filter_single", C.c_int32), ("min_mapq", C.c_uint8), ("min_aligned_length_single", C.c_uint32),
                ("min_percent_identity_single", C.c_float), ("min_aligned_percent_single", C.c_float),
                ("min_aligned_length_pair", C.c_uint32), ("min_percent_identity_pair", C.c_float),
                ("min_aligned_percent_pair", C.c_float)]


def pair_mode_order(af: AlignmentFile, fp: FilterParameters, threads: int = 8):
    """Indices of the records ReferenceSortedBamFilter::read returns in pair mode (filter_out = true)."""
    from coverm_amd import native
    from coverm_amd.native import CovBatch
    L = native.lib()
    r = af.records
    n = r.n_records
    fs, _ = fp.filter_mode()
    pf = _PairFilter(int(fs), fp.min_mapq, fp.min_aligned_length_single, fp.min_percent_identity_single,
                     fp.min_aligned_percent_single, fp.min_aligned_length_pair, fp.min_percent_identity_pair,
                     fp.min_aligned_percent_pair)
    cb = CovBatch()
    for k in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar"):
        a = getattr(r, k)
        setattr(cb, k, a.ctypes.data if a.size else None)
    cb.n_records = n
    qoff = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum([len(q) for q in af.qname], out=qoff[1:])
    blob = b"".join(af.qname) or b"\0"
    mtid = np.ascontiguousarray(af.mtid, dtype=np.int32)
    out = C.c_void_p()
    n_out = C.c_uint64(0)
    L.covh_pair_mode_order.argtypes = [C.POINTER(CovBatch), C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(_PairFilter),
                                       C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.covh_free.argtypes = [C.c_void_p]
    rc = L.covh_pair_mode_order(C.byref(cb), mtid.ctypes.data if n else None, qoff.ctypes.data, blob, C.byref(pf), threads,
                                C.byref(out), C.byref(n_out))
    if rc == native.ERR_NM_MISSING:
        raise SystemExit("Mapping record encountered that does not have an 'NM' auxiliary tag in the SAM/BAM format")
    if rc:
        raise SystemExit("Unexpected data type of NM aux tag" if rc == native.ERR_NM_BADTYPE else "pair filter failed (%d)" % rc)
    order = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(max(1, n_out.value),))[:n_out.value].astype(np.int64)
    L.covh_free(out)
    return order


def _select(r: RecordBatch, idx) -> RecordBatch:
    n = (r.cigar_off[1:].astype(np.int64) - r.cigar_off[:-1].astype(np.int64))[idx]
    off = np.zeros(len(idx) + 1, dtype=np.uint32)
    np.cumsum(n, out=off[1:])
    cig = np.concatenate([r.cigar[r.cigar_off[i]:r.cigar_off[i + 1]] for i in idx]) if len(idx) else np.zeros(0, np.uint32)
    return RecordBatch(r.tid[idx], r.pos[idx], r.flag[idx], r.mapq[idx], r.nm[idx], r.nm_kind[idx], r.l_seq[idx], off,
                       cig.astype(np.uint32))


def reader_stage(af: AlignmentFile, fp: FilterParameters):
    """(records the scan sees, device FilterConfig, primary-alignment count if the host had to count it)."""
    ff = fp.flag_filters
    filt = FilterConfig(ff.include_improper_pairs, ff.include_supplementary, ff.include_secondary)
    records, prim = af.records, None
    if fp.doing_filtering():
        fs, fpairs = fp.filter_mode()
        if fs and not fpairs:
            filt.filter_single = True
            filt.min_mapq = fp.min_mapq
            filt.min_aligned_length = fp.min_aligned_length_single
            filt.min_percent_identity = fp.min_percent_identity_single
            filt.min_aligned_percent = fp.min_aligned_percent_single
        else:
            prim = int(((records.flag & 0x900) == 0).sum())   # filter.rs:129-131, every record read
            records = _select(records, pair_mode_order(af, fp))
    return records, filt, prim


import contextlib


@contextlib.contextmanager
def device_depth(af: AlignmentFile, records, filt, device: int = 0):
    """Per-gene coverage: one finished session per BAM; yields (depth_of(tid), primary-alignment count, session cfg)."""
    with Session(device, filt, 0, want_hist=False, want_identity=False) as s:
        s.set_targets(af.ref_lens)
        s.push(records)
        stats, summ = s.finish()
        yield s.depth, int(summ.num_detected_primary_alignments), s.cfg, (None if os.environ.get("COVERM_GENES_ON_HOST") else s)


def _run_genes(mode, files, et, fp, contig_end_exclusion, gff, feature_type, separator, single_genome, genome_definition,
               print_zeros, device, depth_provider):
    """run_contig / run_genome with --gff (coverm.rs:1557-1590, 2099-2109)."""
    genes = host.Genes.read_gff(gff, feature_type)
    namer_mode, genomes, c2g = 0, None, None
    if mode == "genome":
        if single_genome:
            namer_mode = 1
        elif separator is not None:
            namer_mode = 2
        else:
            if genome_definition is None:
                raise SystemExit("A genome definition is required when using --gff in genome mode")
            genomes, c2g = read_genome_definition(genome_definition)
            namer_mode = 3
    rms = []
    for af in files:
        records, filt, prim = reader_stage(af, fp)
        g_of = np.asarray([c2g.get(n, -1) for n in af.ref_names], dtype=np.int32) if namer_mode == 3 else None
        with depth_provider(af, records, filt, device) as prov:
            depth_of, prim_dev, cfg = prov[:3]
            rms.append(host.gene_coverage(af.ref_names, af.ref_lens, genes, af.stoit_name, records, cfg, depth_of,
                                          prim if prim is not None else prim_dev, et.taker, et.estimators, print_zeros,
                                          namer_mode, separator or "~", g_of, genomes, prov[3] if len(prov) > 3 else None))
    return rms


def device_sample(af: AlignmentFile, fp: FilterParameters, contig_end_exclusion: int, want_hist: bool,
                  want_identity: bool, mask=None, device: int = 0) -> SampleResult:
    """One BAM through a covermhip session (the only provider used outside tests)."""
    ff = fp.flag_filters
    filt = FilterConfig(ff.include_improper_pairs, ff.include_supplementary, ff.include_secondary)
    records = af.records
    prim = None
    if fp.doing_filtering():
        fs, fpairs = fp.filter_mode()
        if fs and not fpairs:
            filt.filter_single = True
            filt.min_mapq = fp.min_mapq
            filt.min_aligned_length = fp.min_aligned_length_single
            filt.min_percent_identity = fp.min_percent_identity_single
            filt.min_aligned_percent = fp.min_aligned_percent_single
        else:
            prim = int(((records.flag & 0x900) == 0).sum())   # filter.rs:129-131, every record read
            records = _select(records, pair_mode_order(af, fp))
    with Session(device, filt, contig_end_exclusion, want_hist, want_identity) as s:
        s.set_targets(af.ref_lens, mask)
        s.push(records)
        stats, summ = s.finish()
        hist = s.hist() if want_hist else None
        if prim is None:
            prim = int(summ.num_detected_primary_alignments)
    return SampleResult(af.stoit_name, stats, hist, prim)


def read_genome_definition(path: str):
    """read_genome_definition_file, genome_parsing.rs:71-141: `genome<TAB>contig [comment]` lines -> (genomes in file
    order, contig -> genome index).  The contig is the first whitespace-separated token of the second column."""
    genomes, idx, c2g = [], {}, {}
    with open(path, newline="") as fh:
        lines = fh.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    for line in lines:
        if line.endswith("\r"):
            line = line[:-1]
        f = line.split("\t")
        if len(f) != 2:
            raise SystemExit('The line "%s" in the genome definition file is not a genome name and contig name separated by a tab' % line)
        g, toks = f[0].strip(), f[1].split()
        if not toks:
            raise SystemExit("Failed to split contig name by whitespace in genome definition file")
        c = toks[0]
        if c in c2g and genomes[c2g[c]] != g:
            raise SystemExit("The contig name '%s' was assigned to multiple genomes" % c)
        if g not in idx:
            idx[g] = len(genomes); genomes.append(g)
        c2g.setdefault(c, idx[g])
    return genomes, c2g


def run(mode: str, files: Sequence[AlignmentFile], methods: Optional[Sequence[str]] = None,
        min_covered_fraction=None, contig_end_exclusion: int = 75, trim_min=5, trim_max=95,
        output_format: str = "dense", no_zeros: bool = False, proper_pairs_only: bool = False,
        exclude_supplementary: bool = False, include_secondary: bool = False, min_read_aligned_length: int = 0,
        min_read_percent_identity=None, min_read_aligned_percent=None, min_mapq: int = 255,
        min_read_aligned_length_pair: int = 0, min_read_percent_identity_pair=None,
        min_read_aligned_percent_pair=None, separator: Optional[str] = None, single_genome: bool = False,
        genome_definition: Optional[str] = None, device: int = 0,
        sample_provider: Callable[..., SampleResult] = device_sample, gff: Optional[str] = None,
        gff_feature_type: Optional[str] = None, depth_provider=None) -> str:
    """Runs `coverm <mode>` on decoded alignment files and returns what the reference prints to stdout."""
    if methods is None:
        methods = ["mean"] if mode == "contig" else ["relative_abundance"]   # cli.rs:2521, 2048
    if min_covered_fraction is None:
        min_covered_fraction = 0 if mode == "contig" else 10                  # cli.rs:2528, 2065
    et = EstimatorsAndTaker.generate(methods, min_covered_fraction, contig_end_exclusion, trim_min, trim_max,
                                     output_format)
    fp = FilterParameters(FlagFilter(not proper_pairs_only, not exclude_supplementary, include_secondary),
                          min_read_aligned_length, parse_percentage(min_read_percent_identity),
                          parse_percentage(min_read_aligned_percent), min_mapq, min_read_aligned_length_pair,
                          parse_percentage(min_read_percent_identity_pair),
                          parse_percentage(min_read_aligned_percent_pair))
    if list(methods) == ["metabat"]:   # add_metabat_filtering_if_required, :1680-1693
        fp.min_percent_identity_single = float(np.float32(0.97001))
        fp.flag_filters = FlagFilter(True, True, True)
    headers = et.headers()
    entry_type = "Contig" if mode == "contig" else "Genome"
    if gff is not None:   # coverm.rs:511-518, 1557-1590
        entry_type = "Gene\tContig" if mode == "contig" else "Gene\tContig\tGenome"
    host.print_headers(et.taker, et.printer, entry_type, headers)
    if gff is not None:
        rms = _run_genes(mode, files, et, fp, contig_end_exclusion, gff, gff_feature_type, separator, single_genome,
                         genome_definition, not no_zeros, device, depth_provider or device_depth)
        host.finalise_printing(et.taker, et.printer, entry_type, headers, rms, et.columns_to_normalise, et.rpkm_column,
                               et.tpm_column)
        return et.taker.text()
    want_hist, want_identity = host.wants(et.estimators)
    names, lens = files[0].ref_names, files[0].ref_lens

    genomes = genome_of_tid = None
    if mode == "genome" and separator is None and not single_genome:
        if genome_definition is None:
            raise SystemExit("genome mode over BAM files needs --separator, --single-genome or --genome-definition")
        genomes, c2g = read_genome_definition(genome_definition)
    if want_identity:   # contig.rs:208 and genome.rs:724 sum over primary reads, genome.rs:220 over not-supplementary ones
        want_identity = "nonsupp" if genomes is not None else "primary"

    samples = []
    for af in files:
        mask = None
        if genomes is not None:
            g_of = np.asarray([c2g.get(n, -1) for n in af.ref_names], dtype=np.int32)
            if (g_of >= 0).sum() == 0:
                raise SystemExit("Error: There are no found reference sequences that are a part of a genome")
            mask = (g_of >= 0).astype(np.uint8)
            genome_of_tid = g_of
        samples.append(sample_provider(af, fp, contig_end_exclusion, want_hist, want_identity, mask=mask,
                                       device=device))
    print_zeros = not no_zeros
    if mode == "contig":
        rms = host.contig_coverage(names, lens, samples, et.taker, et.estimators, print_zeros)
    elif separator is not None or single_genome:
        rms = host.mosdepth_genome_coverage(names, lens, samples, "0" if single_genome else separator, et.taker,
                                            print_zeros, et.estimators, single_genome)
    else:
        rms = host.mosdepth_genome_coverage_with_contig_names(names, lens, samples, genomes, genome_of_tid, et.taker,
                                                              print_zeros, et.estimators)
    host.finalise_printing(et.taker, et.printer, entry_type, headers, rms, et.columns_to_normalise, et.rpkm_column,
                           et.tpm_column)
    return et.taker.text()
