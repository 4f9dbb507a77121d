#!/usr/bin/env python3
"""Headline benchmark: aligned reads/s through `coverm contig` (BAM -> pileup -> per-contig) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1] at N = 1; configs[3] shape — one BAM per GPU — at N > 1):
50 M synthetic 150 bp reads over 5 000 contigs (1.0 Gbp), coordinate sorted, methods
`mean trimmed_mean covered_fraction variance` (SURVEY.md §8d profile, coverm_amd/synth.py).

`value` is measured on basis (i) below, as the bench contract prescribes (inputs resident in HBM when the clock starts); the same
JSON line carries the two wider bases, each beside a CPU figure taken on the SAME basis (`bases`):

  (i)   device_resident  one step = cov_finish (filter + CIGAR expansion + LDS pileup + statistics + every estimator's calculate_coverage
                         for every contig, all on the GPU) + fetch of the floats + the C++ scan loop's control flow (zero rows, ReadsMapped,
                         taker); --host-estimates: histogram fetch + C++ calculate_coverage instead; CPU = oracle scan over records in host memory
  (ii)  push_inclusive   records in page-locked HOST memory -> cov_push_batch -> cov_finish -> fetch -> finalise; CPU = the same scan
  (iii) end_to_end       BAM FILE -> TSV through the coverm-amd binary (streamed ingest) on a realistic-entropy BAM (random bases,
                         Phred-like qualities) of BASELINE config 5's size and flags; CPU = the same decoder (t threads) + oracle scan

`parity_checked`: the oracle's per-contig output over 100 % of the workload is compared with the GPU step's (every f32 bit-for-bit,
every integer statistic); the run exits non-zero if they differ.  For N > 1 each rank holds one sample and rank 0 receives the
per-contig coverages through one RCCL gather.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from coverm_amd import host, native, synth  # noqa: E402
from coverm_amd.engine import FilterConfig, RecordBatch, Session  # noqa: E402
from coverm_amd.host import CoverageEstimator as E  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
METHODS = ["mean", "trimmed_mean", "covered_fraction", "variance"]
ALL_METHODS = ["mean", "trimmed_mean", "covered_fraction", "covered_bases", "variance", "length", "count", "reads_per_base", "rpkm",
               "tpm", "anir"]
BIN = os.path.join(ROOT, "coverm_amd", "coverm-amd")


def estimators(excl=75):
    return [E.new_estimator_mean(0.0, excl, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, excl),
            E.new_estimator_covered_fraction(0.0), E.new_estimator_variance(0.0, excl)]


# ---------------------------------------------------------------------------------------------- CPU side (oracle = checker + baseline)
_ORC = None


def oracle_native():
    """The C oracle rebuilt on THIS box with -O3 -march=native for the cpu_baseline legs."""
    global _ORC
    if _ORC is None:
        from oracle import oracle as O
        src = os.path.join(ROOT, "oracle", "coverm_oracle.c")
        out = os.path.join(ROOT, "oracle", "_build", "libcoverm_oracle_native.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        try:
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-fno-fast-math", "-shared", "-o", out, src, "-lm"])
            lib = C.CDLL(out)
            flags = "-O3 -march=native"
        except Exception:
            lib = O.lib()
            flags = "Makefile default (-O3)"
        lib.orc_out_free.argtypes = [C.c_void_p]
        _ORC = (lib, flags)
    return _ORC


EMIT_DTYPE = np.dtype([("type", "<i4"), ("pad", "<i4"), ("a", "<i8"), ("b", "<u8"), ("cov", "<f4"), ("name_tid", "<i4")])


def oracle_contig_scan(ref_lens, batch, est_params, ff=(True, True, False), fp=None):
    """orc_contig_coverage (contig.rs:13-253 restated in C) over `batch`, one thread like the reference's scan.
    Returns (per-contig f32 coverages [n_contigs x n_est] with zero rows printed, reads mapped, seconds)."""
    from oracle import oracle as O
    from oracle.bamio import BamData
    lib, _ = oracle_native()
    n = batch.n_records
    z = np.zeros(1, np.int32)
    b = BamData([], np.asarray(ref_lens, np.int64), batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq, batch.nm, batch.nm_kind,
                batch.cigar_off, batch.cigar, z, z, z, [], "")
    order = None
    prim = int(((batch.flag & 0x900) == 0).sum())
    t_filter = 0.0
    if fp is not None:     # single-read reader stage (filter.rs:88-116): the oracle's own implementation, timed as part of the scan
        tf = time.perf_counter()
        order, prim = O.reader_stage(b, fp)
        t_filter = time.perf_counter() - tf
    r = O._Records()
    keep = dict(tid=batch.tid, pos=batch.pos, flag=batch.flag, mapq=batch.mapq, nm=batch.nm, nm_kind=batch.nm_kind, l_seq=batch.l_seq,
                cigar_off=batch.cigar_off, cigar=batch.cigar)
    for k, v in keep.items():
        setattr(r, k, v.ctypes.data if v.size else None)
    r.n_records = n
    if order is not None:
        order = np.ascontiguousarray(order, np.uint64)
        r.order = order.ctypes.data
        r.n_order = len(order)
    else:
        r.n_order = n
    tl = np.ascontiguousarray(ref_lens, np.int64)
    out = O._Out()
    rm = O._ReadsMapped()
    ffc = O.FlagFilter(*ff).c()
    t0 = time.perf_counter()
    rc = lib.orc_contig_coverage(C.byref(r), tl.ctypes.data_as(C.c_void_p), C.c_int32(len(tl)), O._params(est_params), C.c_int32(len(est_params)),
                                 C.c_int32(1), C.byref(ffc), C.c_uint64(prim), C.byref(out), C.byref(rm))
    dt = time.perf_counter() - t0 + t_filter
    assert rc == 0, "oracle error %d" % rc
    ev = np.ctypeslib.as_array(C.cast(out.e, C.POINTER(C.c_uint8)), shape=(out.n * EMIT_DTYPE.itemsize,)).view(EMIT_DTYPE)
    cov = ev["cov"][ev["type"] == 1].copy().reshape(len(tl), len(est_params))
    lib.orc_out_free(C.byref(out))
    return cov, (int(rm.num_mapped_reads), int(rm.num_reads)), dt


def oracle_dense_text(names, cov, reads_mapped, methods, stoit):
    """The oracle's dense table (coverage_printer.rs:359-553 restated in oracle.py) over per-contig coverages `cov`
    [n_contigs x n_methods]: header row, then one row per contig, RPKM / TPM normalised by the printer — the text `coverm contig`
    writes for one sample."""
    import io
    from oracle import oracle as O
    stream = io.StringIO()
    et = O.estimators_and_taker(methods, 0, 75, 5, 95, "dense", stream)
    headers = [h for e in et["estimators"] for h in O.COLUMN_HEADERS[e.kind]]
    taker = et["taker"]
    taker.start_stoit(stoit)
    for i, n in enumerate(names):
        taker.start_entry(i, n)
        for v in cov[i]:
            taker.add_single_coverage(v)
        taker.finish_entry()
    O.print_headers(et["printer"], "Contig", headers, stream)
    O.print_dense_cached("Contig", headers, taker, stream, [O.ReadsMapped(reads_mapped[0], reads_mapped[1])], et["columns_to_normalise"], et["rpkm"], et["tpm"])
    return stream.getvalue()


def oracle_estimators(excl=75, methods=METHODS):
    from oracle import oracle as O
    m = {"mean": lambda: O.est_mean(0.0, excl, False), "trimmed_mean": lambda: O.est_trimmed_mean(0.05, 0.95, 0.0, excl),
         "covered_fraction": lambda: O.est_covered_fraction(0.0), "covered_bases": lambda: O.est_covered_bases(0.0),
         "variance": lambda: O.est_variance(0.0, excl), "length": O.est_length, "count": O.est_read_count,
         "reads_per_base": O.est_reads_per_base, "rpkm": lambda: O.est_rpkm(0.0), "tpm": lambda: O.est_tpm(0.0), "anir": O.est_anir}
    return [m[k]() for k in methods]


def parity_check(ref, batch, gpu_cov, gpu_stats, gpu_hist):
    """GPU step output == oracle over the FULL workload: per-contig f32 of all four methods bit-for-bit, and the integer
    sufficient statistics + histograms (oracle's orc_integer_stats)."""
    from oracle import oracle as O
    from oracle.bamio import BamData
    cov, rmp, dt = oracle_contig_scan(ref.lengths, batch, oracle_estimators())
    mapped = rmp[0]
    n = len(ref.lengths)
    g = gpu_cov.reshape(n, len(METHODS))
    f32_equal = bool((g.view(np.uint32) == cov.view(np.uint32)).all())
    z = np.zeros(1, np.int32)
    b = BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq.astype(np.int32), batch.nm, batch.nm_kind,
                batch.cigar_off, batch.cigar, z, z, z, [], "")
    exp, exp_hist, _ = O.integer_stats(b, O.FlagFilter(True, True, False), None, 75)
    ints_equal = True
    for fld in ("n_primary", "n_pass", "n_nonsupp", "sum_nm", "sum_indel", "win_sum_d", "win_sum_d2", "win_covered", "full_covered",
                "win_min_d", "win_max_d", "hist_len"):
        ints_equal &= bool((gpu_stats[fld] == exp[fld]).all())
    hist_equal = len(gpu_hist) == len(exp_hist) and bool((gpu_hist == exp_hist).all())
    ok = f32_equal and ints_equal and hist_equal
    return dict(contigs=n, methods=METHODS, reads=batch.n_records, equal=ok, f32_bitwise_equal=f32_equal, integer_stats_equal=ints_equal,
                histograms_equal=hist_equal, oracle_scan_s=dt), (mapped, dt), (cov, rmp)


# ---------------------------------------------------------------------------------------------- pinned host arrays
def pinned_copy(batch):
    """The batch copied into page-locked memory from cov_host_alloc (what the decoder fills in the product)."""
    L = native.lib()
    L.cov_host_alloc.restype = C.c_void_p
    L.cov_host_alloc.argtypes = [C.c_size_t]
    L.cov_host_free.argtypes = [C.c_void_p]
    out, ptrs = {}, []
    for k in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar"):
        a = getattr(batch, k)
        p = L.cov_host_alloc(max(a.nbytes, 1))
        assert p, "cov_host_alloc failed"
        ptrs.append(p)
        v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(a.nbytes, 1),))[:a.nbytes].view(a.dtype)
        v[:] = a
        out[k] = v
    return RecordBatch(**out), ptrs


def finalise(ref, est, stats, summ, hist, name="sample0", estimates=None):
    """contig.rs:40-104 over one sample's device results: zero rows, ReadsMapped, the taker.  `estimates`: calculate_coverage already
    evaluated on the device (Session.estimates()), else the host evaluates it from the integer statistics + histogram."""
    taker = host.CoverageTaker.new_cached_single_float_coverage_taker(len(est))
    sample = host.SampleResult(name, stats, hist, int(summ.num_detected_primary_alignments))
    rm = host.contig_coverage(ref.names, ref.lengths, [sample], taker, est, True, estimates=None if estimates is None else [estimates])
    return taker, rm


# ---------------------------------------------------------------------------------------------- end to end
E2E_FLAGS = ["--min-read-percent-identity", "95", "--min-read-aligned-length", "50", "--proper-pairs-only"]


def run_binary(cmd, reps, sleep_s=4.0, env=None):
    """`reps` runs of the product binary; (median seconds, every run's seconds, stderr of the median run, its VmHWM bytes).
    A process that starts within a second of another one that just released tens of GB of device memory stalls ~1 s in its
    first large allocations (measured: 2.1 s against 1.1 s after a pause): repetitions start on an idle device."""
    runs = []
    e = dict(os.environ, COVERM_CLI_TIMING="1", **(env or {}))
    for rep in range(reps):
        if rep:
            time.sleep(sleep_s)
        t0 = time.perf_counter()
        p = subprocess.run(cmd, capture_output=True, text=True, env=e)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            raise RuntimeError("coverm-amd failed: " + p.stderr[-2000:])
        runs.append((dt, p.stderr))
    order = sorted(range(len(runs)), key=lambda k: runs[k][0])
    med = runs[order[len(order) // 2]]
    hwm = [l for l in med[1].splitlines() if "VmHWM" in l]
    rss = int(hwm[0].split()[-2]) * 1024 if hwm else None
    return med[0], [round(r[0], 3) for r in runs], med[1], rss


def timing_lines(stderr, n=8):
    return [l for l in stderr.splitlines() if "stream read" in l or "ingest" in l or "VmHWM" in l or "main:" in l or "pair filter" in l][:n]


def inflate_backend():
    """What the host readers inflate with on THIS box (csrc/host_bam.cpp binds libdeflate at run time and falls back to zlib)."""
    from coverm_amd import bam as cbam
    L = cbam._lib()
    L.covh_inflate_backend.restype = C.c_char_p
    return L.covh_inflate_backend().decode()


def cpu_decode(path, threads, reps):
    """The CPU side's decoder (csrc/host_bam.cpp covh_bam_open: threaded libdeflate inflate + parse) `reps` times; (best seconds,
    every run's seconds, records of the last run)."""
    from coverm_amd import bam as cbam
    L = cbam._lib()
    secs, recs = [], None
    for rep in range(reps):
        err = C.create_string_buffer(512)
        t0 = time.perf_counter()
        h = L.covh_bam_open(path.encode(), threads, 0, err, 512)      # the decode itself, timed without Python-side copies
        secs.append(time.perf_counter() - t0)
        assert h, err.value
        if rep == reps - 1:
            nrec, ncg = int(L.covh_bam_n_records(h)), int(L.covh_bam_n_cigar(h))
            cb = native.CovBatch()
            L.covh_bam_batch(h, C.byref(cb))
            cp = cbam._copy
            recs = RecordBatch(cp(cb.tid, np.int32, nrec), cp(cb.pos, np.int32, nrec), cp(cb.flag, np.uint16, nrec), cp(cb.mapq, np.uint8, nrec),
                               cp(cb.nm, np.uint32, nrec), cp(cb.nm_kind, np.uint8, nrec), cp(cb.l_seq, np.uint32, nrec),
                               cp(cb.cigar_off, np.uint32, nrec + 1), cp(cb.cigar, np.uint32, ncg))
        L.covh_bam_close(h)
    return min(secs), [round(x, 3) for x in secs], recs


def end_to_end(a, threads):
    """Basis (iii): BAM FILE -> TSV.  Config 5's size and flags (200 M reads, --min-read-percent-identity 95 --min-read-aligned-length 50
    --proper-pairs-only, all eleven methods) on a realistic-entropy BAM; GPU = the coverm-amd binary (device ingest), MEDIAN of three
    runs; CPU = the same decoder with the same thread count, then the oracle's scan (one thread, like the reference's), BEST of three
    each.  The binary's table must equal the oracle's dense table character for character (all eleven columns, RPKM and TPM
    normalised by the oracle's printer).  A second line repeats the GPU side on the same records written at BGZF level 6 (what
    samtools writes)."""
    from coverm_amd import bam as cbam
    from oracle import oracle as O
    res = dict(reads=a.e2e_reads, threads=threads)
    tmpdir = tempfile.mkdtemp(prefix="covbench", dir=a.tmp)
    try:
        free = shutil.disk_usage(tmpdir).free
        reads = a.e2e_reads
        if free < reads * 190:
            reads = max(1_000_000, int(free // 380))
            res["note_disk"] = "only %.1f GB free under %s: end-to-end leg reduced to %d reads" % (free / 1e9, tmpdir, reads)
        t0 = time.time()
        ref = synth.make_reference(a.contigs, a.bp, seed=1)
        batch = synth.make_reads(ref, reads, seed=3)
        res["generation_s"] = time.time() - t0
        path = os.path.join(tmpdir, "config5.bam")
        t0 = time.time()
        cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=2, level=1, threads=threads)
        res["bam_write_s"] = time.time() - t0
        size = os.path.getsize(path)
        res.update(reads=reads, bam_bytes=size, bam_bytes_per_read=size / reads, seq_qual="random bases, Phred-like binned qualities, Illumina-style names",
                   bam_location=tmpdir + (" (tmpfs: storage speed excluded, as with a warm page cache)" if tmpdir.startswith("/dev/shm") else ""))
        out_tsv = os.path.join(tmpdir, "gpu.tsv")
        cmd = [BIN, "contig", "-b", path, "-m"] + ALL_METHODS + E2E_FLAGS + ["-t", str(threads), "-o", out_tsv]
        # five runs: the first one of a fresh file is cold (1.2-1.6 s), and about one run in eight starts with a runtime initialisation of 0.2 s instead
        # of 0.07 s (profiles/r06_*_ab_200M.json: `sessions_s`) — with three runs either one decided the median
        gpu_s, rep_seconds, gpu_err, gpu_rss = run_binary(cmd, 5)
        gpu_text = open(out_tsv).read()
        # the same command as ONE process (csrc/cli_main.cc: by default the work runs in a child and the command returns when the table is
        # written, the kernel taking the runtime's queues, device mappings and page-locked slots apart behind it; here the caller waits for that too)
        one_s, one_reps, _, _ = run_binary(cmd, 3, env={"COVERM_NO_FAST_EXIT": "1"})
        one_s = min(one_reps)
        mapped = [l for l in gpu_err.splitlines() if "reads mapped out of" in l]
        # ---- CPU, same basis: same decoder + oracle scan, best of three each
        dec_s, dec_all, recs = cpu_decode(path, threads, 3)
        fp = O.FilterParameters(O.FlagFilter(False, True, False), 50, float(np.float32(0.95)), 0.0, 255, 0, 0.0, 0.0)
        est = oracle_estimators(75, ALL_METHODS)
        scans = []
        for rep in range(3):
            cov, rmp, scan_s = oracle_contig_scan(ref.lengths, recs, est, ff=(False, True, False), fp=fp)
            scans.append(scan_s)
        scan_s = min(scans)
        want_text = oracle_dense_text(ref.names, cov, rmp, ALL_METHODS, "config5")
        same = gpu_text == want_text
        res.update(
            gpu=dict(seconds=gpu_s, seconds_is="median of five runs", reads_per_s=rmp[0] / gpu_s, reads_per_s_is="considered (aligned, filter-passing) reads per second: the metric's unit",
                     records_per_s=reads / gpu_s, rep_seconds=rep_seconds, max_rss_bytes=gpu_rss,
                     one_process_seconds=one_s, one_process_rep_seconds=one_reps,
                     one_process_is="best of three runs with COVERM_NO_FAST_EXIT=1: no launcher / child split, the caller also waits for the runtime's teardown",
                     command=" ".join(["coverm-amd"] + cmd[1:]), stderr_mapped=mapped[:1], stderr_timing=timing_lines(gpu_err)),
            cpu=dict(decode_s=dec_s, decode_runs=dec_all, scan_s=scan_s, scan_runs=[round(x, 3) for x in scans], seconds_is="best of three runs each",
                     reads_per_s_serial=rmp[0] / (dec_s + scan_s), reads_per_s_overlapped=rmp[0] / max(dec_s, scan_s),
                     records_per_s_serial=reads / (dec_s + scan_s), records_per_s_overlapped=reads / max(dec_s, scan_s),
                     inflate_backend=inflate_backend(), nproc=os.cpu_count(), usable_cpus=usable_cpus(), decode_threads=threads, scan_threads=1,
                     decoder="csrc/host_bam.cpp covh_bam_open, %d threads, %s" % (threads, inflate_backend()), scan="oracle/coverm_oracle.c, 1 thread, %s" % oracle_native()[1],
                     note="the reference overlaps htslib's inflate pool with its single scan thread: its rate lies between the two figures, "
                          "at or below the overlapped one"),
            speedup_vs_cpu_overlapped=(reads / gpu_s) / (reads / max(dec_s, scan_s)), speedup_vs_cpu_serial=(reads / gpu_s) / (reads / (dec_s + scan_s)),
            speedup_vs_cpu_overlapped_one_process=max(dec_s, scan_s) / one_s,
            target=">= 10x the CPU path (BASELINE.json north_star)", tables_equal=same,
            tables_compared="the binary's TSV == the oracle's dense table, text equality over all %d columns x %d contigs" % (len(ALL_METHODS), len(ref.names)),
            considered_reads=rmp[0])
        del recs
        # ---- the same records at BGZF level 6 (what samtools / htslib write): fewer bytes over PCIe, longer matches for the device's LZ stage
        if not a.no_level6:
            os.remove(path)
            p6 = os.path.join(tmpdir, "config5.bam")       # same stem: same sample name in the table
            t0 = time.time()
            cbam.write_bam(p6, ref.names, ref.lengths, batch, with_seq=2, level=6, threads=threads)
            w6 = time.time() - t0
            size6 = os.path.getsize(p6)
            g6, reps6, err6, rss6 = run_binary(cmd, 5)
            same6 = open(out_tsv).read() == want_text
            d6, d6_all, _ = cpu_decode(p6, threads, 3)
            res["level6"] = dict(bam_bytes=size6, bam_bytes_per_read=size6 / reads, bam_write_s=w6, gpu_seconds=g6, seconds_is="median of five runs", rep_seconds=reps6,
                                 reads_per_s=rmp[0] / g6, records_per_s=reads / g6, max_rss_bytes=rss6, cpu_inflate_backend=inflate_backend(), stderr_timing=timing_lines(err6), cpu_decode_s=d6, cpu_decode_runs=d6_all, cpu_scan_s=scan_s,
                                 speedup_vs_cpu_overlapped=(reads / g6) / (reads / max(d6, scan_s)), speedup_vs_cpu_serial=(reads / g6) / (reads / (d6 + scan_s)),
                                 tables_equal=same6)
            res["tables_equal"] = res["tables_equal"] and same6
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return res


def binary_config_legs(a, threads, ref, batch, oracle_cfg2):
    """BASELINE configs 2, 3 and (one device's share of) 4 at FULL size through the product binary: the headline's own 50 M-read
    sample written as a BAM file, `coverm-amd contig` (config 2 / 4: the four methods) and `coverm-amd genome --genome-definition`
    (config 3: 500 genomes, relative_abundance rpkm tpm), each table compared with the oracle's text."""
    from coverm_amd import bam as cbam
    from oracle import oracle as O
    from oracle.bamio import BamData
    res = {}
    tmpdir = tempfile.mkdtemp(prefix="covbench", dir=a.tmp)
    try:
        path = os.path.join(tmpdir, "sample0.bam")
        t0 = time.time()
        cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=2, level=1, threads=threads)
        res["bam_write_s"] = time.time() - t0
        res["bam_bytes"] = os.path.getsize(path)
        out = os.path.join(tmpdir, "out.tsv")
        cmd2 = [BIN, "contig", "-b", path, "-m"] + METHODS + ["-t", str(threads), "-o", out]
        s2, reps2, err2, rss2 = run_binary(cmd2, 3, sleep_s=2.0)
        cov, rmp = oracle_cfg2
        same2 = open(out).read() == oracle_dense_text(ref.names, cov, rmp, METHODS, "sample0")
        res["config2_contig"] = dict(seconds=s2, seconds_is="median of three runs", rep_seconds=reps2, reads_per_s=rmp[0] / s2, records_per_s=batch.n_records / s2, max_rss_bytes=rss2,
                                     tables_equal=same2, stderr_timing=timing_lines(err2, 4),
                                     note="also config 4's per-device share (one 50 M-read BAM per device)")
        gd = os.path.join(tmpdir, "genomes.tsv")
        with open(gd, "w") as fh:
            fh.write("".join("%s\t%s\n" % (ref.genomes[g], n) for n, g in zip(ref.names, ref.genome_of_contig)))
        cmd3 = [BIN, "genome", "-b", path, "--genome-definition", gd, "-m", "relative_abundance", "rpkm", "tpm", "-t", str(threads), "-o", out]
        s3, reps3, err3, rss3 = run_binary(cmd3, 3, sleep_s=2.0)
        got3 = open(out).read()
        z = np.zeros(1, np.int32)
        b = BamData(ref.names, ref.lengths, batch.tid, batch.pos, batch.flag, batch.mapq, batch.l_seq.astype(np.int32), batch.nm, batch.nm_kind,
                    batch.cigar_off, batch.cigar, z, z, z, [], "")
        t0 = time.perf_counter()
        want3 = O.run_cli("genome", [path], bams=[b], methods=["relative_abundance", "rpkm", "tpm"], genome_definition=gd)
        res["config3_genome"] = dict(seconds=s3, seconds_is="median of three runs", rep_seconds=reps3, reads_per_s=rmp[0] / s3, records_per_s=batch.n_records / s3, max_rss_bytes=rss3,
                                     genomes=len(ref.genomes), tables_equal=got3 == want3, oracle_cli_s=time.perf_counter() - t0, stderr_timing=timing_lines(err3, 4))
        res["tables_equal"] = bool(same2 and got3 == want3)
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return res


SAMPLE_WRITER = r'''
import sys
sys.path.insert(0, sys.argv[1])
from coverm_amd import bam as cbam, synth
contigs, bp, reads, seed, threads, path = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
ref = synth.make_reference(contigs, bp, seed=1)
b = synth.make_reads(ref, reads, seed=seed)
cbam.write_bam(path, ref.names, ref.lengths, b, with_seq=2, level=1, threads=threads)
print("N_RECORDS %d" % b.n_records, flush=True)
'''


def sample_writers(n_samples, reads, threads):
    """How many samples to generate at once: bounded by the samples, by the CPUs (a generator is one numpy thread, then `threads` / P
    compressor threads) and by the memory a 50 M-read sample takes while it is built (~120 bytes per read)."""
    avail = 1 << 62
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except Exception:
        pass
    return max(1, min(n_samples, max(1, threads // 2), int(avail // 2 // max(1, reads * 120))))


def write_samples(a, seeds, paths, threads):
    """The N distinct samples of config 4 (one synthetic BAM per seed), written by up to P processes at once — at N = 8 the samples cost
    4-5 minutes one after the other (20 s of single-threaded numpy + the BGZF writer each) and the other ranks' CPUs are idle meanwhile.
    Same bytes as synth.make_reads + bam.write_bam in this process (the writer's output does not depend on its thread count).  Returns the
    records written."""
    P = sample_writers(len(seeds), a.reads, threads)
    per = max(1, threads // P)
    todo = list(zip(seeds, paths))
    running, nrec = [], 0
    while todo or running:
        while todo and len(running) < P:
            seed, path = todo.pop(0)
            running.append(subprocess.Popen([sys.executable, "-c", SAMPLE_WRITER, ROOT, str(a.contigs), str(a.bp), str(a.reads), str(seed), str(per), path],
                                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        pr = running.pop(0)
        so, se = pr.communicate()
        if pr.returncode != 0 or "N_RECORDS" not in so:
            for q in running:
                q.kill()
            raise RuntimeError("sample writer failed: " + se[-1500:])
        nrec += int(so.split("N_RECORDS")[1].split()[0])
    return nrec, P


def multi_device_legs(a, threads, world, devs=None):
    """The PRODUCT's multi-GPU path (coverm-amd --devices 0..N-1, one process, one reader + session per device) at N > 1:
    config 4 (N BAMs x 50 M reads, one per device) and config 5 (one 200 M-read BAM cut into N tid spans, RCCL gather of the
    per-contig blocks), reads/s and per-device ingest stamps, each table compared with the single-device run of the same command."""
    from coverm_amd import bam as cbam
    res = dict(devices=world)
    tmpdir = tempfile.mkdtemp(prefix="covbench", dir=a.tmp)
    devs = devs or "0-%d" % (world - 1)
    res["devices_arg"] = devs
    try:
        ref = synth.make_reference(a.contigs, a.bp, seed=1)
        # ---- config 4: N DISTINCT samples (seeds 10 .. 10 + N - 1), one per device: N readers pull N different files through
        # the host's memory system at once — what N devices really contend for (N names for one file would share one page-cache copy)
        t0 = time.time()
        paths = [os.path.join(tmpdir, "sample%d.bam" % k) for k in range(world)]
        nrec, writers = write_samples(a, [10 + k for k in range(world)], paths, threads)
        nbytes = sum(os.path.getsize(q) for q in paths)
        gen4 = time.time() - t0
        res["sample_writers_at_once"] = writers
        out = os.path.join(tmpdir, "out.tsv")
        cmd1 = [BIN, "contig", "-b"] + paths + ["-m"] + METHODS + ["-t", str(threads), "-o", out]
        s1, _, _, _ = run_binary(cmd1 + ["--devices", devs.split(",")[0].split("-")[0]], 1)      # the same files, one device, one after the other
        text1 = open(out).read()
        sn, repsn, errn, rssn = run_binary(cmd1 + ["--devices", devs], 3)
        same4 = open(out).read() == text1
        res["config4_samples"] = dict(bams=world, reads=nrec, bam_bytes=nbytes, generation_and_write_s=gen4, seconds=sn, seconds_is="median of three runs", rep_seconds=repsn,
                                      records_per_s=nrec / sn, records_per_s_is="all records of the files (aligned or not); no oracle runs in this leg to count the considered ones", host_read_GBps=nbytes / sn / 1e9, single_device_all_bams_seconds=s1, speedup_vs_single_device=s1 / sn,
                                      max_rss_bytes=rssn, tables_equal=bool(same4),
                                      stderr_timing=[l for l in errn.splitlines() if "device ingest:" in l or "[covermhip] ingest" in l][:2 * world],
                                      note="N distinct BAMs (seeds 10 ..); per device: file read = time its reader threads spent in pread, staging waits = time they "
                                           "waited for the device to take a slot (reader-bound when the first dominates, link- or device-bound when the second does)")
        for q in paths:
            os.remove(q)
        # ---- config 5: one big BAM, N tid spans
        big = synth.make_reads(ref, a.e2e_reads, seed=3)
        path = os.path.join(tmpdir, "config5.bam")
        cbam.write_bam(path, ref.names, ref.lengths, big, with_seq=2, level=1, threads=threads)
        nbig = big.n_records
        del big
        cmd1 = [BIN, "contig", "-b", path, "-m"] + ALL_METHODS + E2E_FLAGS + ["-t", str(threads), "-o", out]
        s1, reps1, _, _ = run_binary(cmd1, 3)
        text1 = open(out).read()
        sn, repsn, errn, rssn = run_binary(cmd1 + ["--devices", devs], 3)
        res["config5_spans"] = dict(reads=nbig, bam_bytes=os.path.getsize(path), seconds=sn, seconds_is="median of three runs", rep_seconds=repsn, records_per_s=nbig / sn,
                                    single_device_seconds=s1, single_device_rep_seconds=reps1, speedup_vs_single_device=s1 / sn, max_rss_bytes=rssn,
                                    host_read_GBps=os.path.getsize(path) / sn / 1e9, tables_equal=open(out).read() == text1,
                                    stderr_timing=[l for l in errn.splitlines() if "device ingest:" in l or "[covermhip] ingest" in l][:2 * world])
        res["tables_equal"] = bool(same4 and res["config5_spans"]["tables_equal"])
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("COVERM_BENCH_READS", 50_000_000)))
    ap.add_argument("--contigs", type=int, default=5000)
    ap.add_argument("--bp", type=int, default=1_000_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (parity check, CPU baselines) and the end-to-end leg")
    ap.add_argument("--e2e-reads", type=int, default=int(os.environ.get("COVERM_BENCH_E2E_READS", 200_000_000)))
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-level6", action="store_true", help="skip the second end-to-end line (same records written at BGZF level 6)")
    ap.add_argument("--no-binary-legs", action="store_true", help="skip configs 2 / 3 at full size through the coverm-amd binary")
    ap.add_argument("--host-estimates", action="store_true", help="evaluate calculate_coverage on the host from the integer statistics + histogram (the path before cov_set_estimators), for A/B")
    ap.add_argument("--no-multi-device-e2e", action="store_true", help="N > 1: skip the coverm-amd --devices legs (configs 4 and 5 through the product's multi-GPU path)")
    ap.add_argument("--time-limit", type=float, default=float(os.environ.get("COVERM_BENCH_TIME_LIMIT", 1500)),
                    help="seconds this command may take: the plan (legs and what each is expected to cost) is printed up front, and a leg that cannot finish inside what is left is skipped and recorded as skipped instead of dying in the caller's timeout")
    ap.add_argument("--tmp", default=os.environ.get("COVERM_BENCH_TMP", default_tmp()),
                    help="where the end-to-end leg writes its BAM (default: /dev/shm when it has room, so that storage speed is not part of the figure)")
    a = ap.parse_args()
    BUDGET["t0"] = time.time() - float(os.environ.get("COVERM_BENCH_ELAPSED", 0))      # (the relaunch below hands over what it used)
    BUDGET["limit"] = a.time_limit

    share = os.environ.get("COVERM_BENCH_SHARE_GPU") == "1"
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` on its own: become N ranks (one per GPU) under torch.distributed.run, as the driver's launch line does
        check_devices(a.gpus, share)
        os.execvpe(sys.executable, relaunch_cmd(a.gpus, sys.argv[1:]), dict(os.environ, MASTER_ADDR="127.0.0.1", COVERM_BENCH_ELAPSED=str(time.time() - BUDGET["t0"])))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE): the line would be mislabelled" % (a.gpus, world))
    check_devices(world, share)
    if rank == 0:
        print_plan(a, world, share)
    # Functional check of the N > 1 path on a single-GPU box: COVERM_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # exchanges over gloo (RCCL refuses two ranks on one device).  Never set by the driver; such a line is not a measurement.
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = torch.device("cpu") if share else dev          # where the exchanged tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            import datetime
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(minutes=45))   # ranks > 0 wait while rank 0 runs the binary's legs

    # ---- synthetic sample of this rank (one BAM per GPU; seed 2 at N=1, 10+rank otherwise)
    t0 = time.time()
    ref = synth.make_reference(a.contigs, a.bp, seed=1)
    seed = 2 if world == 1 else 10 + rank
    cache = os.environ.get("COVERM_BENCH_CACHE")      # profiling passes repeat the same command six times: the sample is generated once
    cpath = os.path.join(cache, "reads_%d_%d_%d_%d.npz" % (a.reads, a.contigs, a.bp, seed)) if cache else None
    if cpath and os.path.exists(cpath):
        z = np.load(cpath)
        batch = RecordBatch(*[z[k] for k in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")])
    else:
        batch = synth.make_reads(ref, a.reads, seed=seed)
        if cpath:
            np.savez(cpath, **{k: getattr(batch, k) for k in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")})
    gen_s = time.time() - t0
    dt = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in
          ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")}
    torch.cuda.synchronize()

    est = estimators()
    want_hist, want_id = host.wants(est)
    sess = Session(local_rank, FilterConfig(), 75, want_hist, want_id)
    sess.set_targets(ref.lengths)
    dev_est = not a.host_estimates
    if dev_est:
        sess.set_estimators(est)      # CoverageEstimator::calculate_coverage of every contig inside cov_finish (k_estimate)
    sess.push_device(dt, batch.n_records)
    n_cov = len(ref.lengths) * len(est)
    gather_buf = [torch.empty(n_cov, dtype=torch.float32, device=xdev) for _ in range(world)] if (dist and rank == 0) else None

    def step():
        stats, summ = sess.finish()
        if dev_est:
            hist = None
            taker, rm = finalise(ref, est, stats, summ, None, "sample%d" % rank, estimates=sess.estimates())
        else:
            hist = sess.hist()
            taker, rm = finalise(ref, est, stats, summ, hist, "sample%d" % rank)
        if dist:
            # per-contig coverages of this sample -> rank 0 (one gather over RCCL/xGMI)
            cov = taker.cached_coverages(0)   # n_contigs x n_estimators f32, contig order (zeros printed)
            t = torch.from_numpy(cov).to(xdev)
            dist.gather(t, gather_buf, dst=0)
        return summ, rm, taker, stats, hist

    for _ in range(a.warmup):
        summ, rm, taker, stats, hist = step()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = {k: 0.0 for k in native.KERNEL_NAMES.values()}
    for _ in range(a.steps):
        summ, rm, taker, stats, hist = step()
        for k, v in sess.kernel_ms().items():
            kms[k] += v[0]
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    considered = int(summ.n_considered)
    if dist:
        tt = torch.tensor([elapsed, float(considered)], dtype=torch.float64, device=xdev)
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        total_reads = int(tt[1].item())
    else:
        total_reads = considered
    gpu_cov = taker.cached_coverages(0)
    if hist is None:
        hist = sess.hist()        # (for the parity check of the integer statistics: the step itself no longer needs the bins on the host)
    pipe_bytes = sess.algorithmic_bytes()
    n_tiles = sess_tiles(sess, ref)
    sess.close()
    del dt
    torch.cuda.empty_cache()

    exit_code = 0
    if rank == 0:
        for k in kms:
            kms[k] /= a.steps
        R = batch.n_records
        ncig = int(batch.cigar_off[-1]) - int(batch.cigar_off[0])
        # algorithmic HBM bytes per launch (DESIGN.md "Algorithmic bytes")
        kbytes = {"k_prep": R * 24 + ncig * 4 + R * 8,                    # SoA + CIGAR read once, run words written
                  "k_pileup": R * 8 + n_tiles * 32 + len(ref.lengths) * 160,  # run words + tile descriptors + results
                  "k_ranges": n_tiles * 40, "k_identity": R * 32, "k_hist": 0, "k_hist_compact": 0}
        dom = max(kms, key=kms.get)
        pipe_ms = sum(kms.values())
        aligned_bp = synth.aligned_bases(batch) * (considered / max(1, R))
        prof = profile_numbers()
        per_kernel = {}
        for k in ("k_prep", "k_pileup"):
            ach = kbytes[k] / (kms[k] * 1e-3) / 1e9 if kms[k] > 0 else 0.0
            per_kernel[k] = {"kernel_ms": kms[k], "algorithmic_bytes": kbytes[k], "hbm_achieved_GBps": ach, "hbm_frac_of_8TBps": ach / HBM_PEAK_GBPS,
                             "hbm_traffic_bytes_from_committed_profile": prof.get("traffic", {}).get(k), "pipes_from_committed_profile": prof.get("pipes", {}).get(k)}
        domk = per_kernel.get(dom, {})
        # The rubric's roofline: algorithmic bytes of the dominant kernel / its launch duration (HIP events of THIS run) against the HBM peak.
        # k_prep_lean streams the record store (round 6: ~0.65 of the HBM peak by its algorithmic bytes, VALU issue ~80 % busy); k_pileup_fast keeps the
        # depth array in LDS and is bound by VALU issue (~86 % busy), not by HBM: its figure against the HBM peak is small by construction.  What the
        # counters of the committed profile say about both is carried beside the figure (`kernels.*.pipes_from_committed_profile`).
        roof = {"bound": "hbm", "kernel": {"k_pileup": "k_pileup_fast", "k_prep": "k_prep_lean (+ k_prep_generic, k_post_prep: one timed group)"}.get(dom, dom), "achieved": domk.get("hbm_achieved_GBps"), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": (domk.get("hbm_achieved_GBps") or 0.0) / HBM_PEAK_GBPS, "traffic": domk.get("hbm_traffic_bytes_from_committed_profile"),
                "traffic_source": "committed rocprofv3 PMC profile (profiles/pmc_traffic.json, %s), not this run" % prof.get("traffic", {}).get("_source"),
                "note": "the k_prep group (k_prep_lean streams the record store) reaches %.3f of the HBM peak, k_pileup_fast (depth array in LDS, VALU-bound, not HBM-bound) %.3f; their launch times differ by %.1f %%; "
                        "see kernels" % (per_kernel["k_prep"]["hbm_frac_of_8TBps"], per_kernel["k_pileup"]["hbm_frac_of_8TBps"],
                                                         100.0 * abs(kms["k_prep"] - kms["k_pileup"]) / max(kms["k_prep"], kms["k_pileup"], 1e-9))}
        roof["ingest"] = ingest_roofline()
        roof.update({"kernel_ms": kms.get(dom), "all_kernels_ms": kms, "kernels": per_kernel,
                     "pipeline": {"algorithmic_bytes": pipe_bytes, "kernels_ms": pipe_ms, "achieved_GBps": pipe_bytes / (pipe_ms * 1e-3) / 1e9 if pipe_ms else 0.0,
                                  "frac_of_8TBps": pipe_bytes / (pipe_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if pipe_ms else 0.0}})
        out = {
            "metric": "aligned reads/s through coverm contig (mean trimmed_mean covered_fraction variance)",
            "value": total_reads * a.steps / elapsed, "unit": "aligned reads/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32 depth / u64 sums / f32 estimators", "data": "synthetic",
            "estimators_evaluated_on": "device (k_estimate inside cov_finish: the reference's f32 expressions, one rounding per operation; floats checked bit for bit against the oracle in parity_checked)" if dev_est else "host (csrc/host_coverage.cpp calculate)",
            "value_basis": "(i) device_resident: records already in HBM when the clock starts (bench contract); bases (ii) and (iii) below include the host side",
            "config": {"workload": "coverm contig, %d-read synthetic sorted BAM over %d contigs (%.2f Gbp) per GPU, "
                                   "--methods %s, records resident in HBM" % (a.reads, a.contigs,
                                                                             ref.lengths.sum() / 1e9, " ".join(METHODS)),
                       "reads_per_gpu": a.reads, "contigs": a.contigs, "reference_bp": int(ref.lengths.sum()),
                       "samples": world, "rccl_ranks": (dist.get_world_size() if dist else 1), "exchange_backend": (dist.get_backend() if dist else None),
                       "sharding": ("FUNCTIONAL CHECK ONLY: ranks share one GPU, gloo exchange" if share else
                                                       "one sample per GPU, RCCL gather of per-contig coverages") if world > 1 else "single GPU"},
            "gbp_per_s": aligned_bp * world * a.steps / elapsed / 1e9,
            "roofline": roof,
            "host": {"generation_s": gen_s, "nproc": os.cpu_count(), "usable_cpus": usable_cpus()},
        }
        if not a.no_cpu_baseline and world == 1:     # CPU legs (parity at full size, baselines, end to end) on rank 0 at N = 1 only
            threads = max(1, int(os.environ.get("COVERM_BENCH_THREADS", usable_cpus())))
            par, (cpu_mapped, cpu_dt), oracle_cfg2 = parity_check(ref, batch, gpu_cov, stats, hist)
            out["parity_checked"] = par
            if not par["equal"]:
                exit_code = 3
            cpu_rate = cpu_mapped / cpu_dt
            out["cpu_baseline"] = dict(value=cpu_rate, unit="aligned reads/s", cores=1, kind="port", nproc=os.cpu_count(), usable_cpus=usable_cpus(),
                                       inflate_backend=inflate_backend() + " (end-to-end legs only: this leg starts from decoded records)",
                                       sample="100%% of the workload (%d records, whole contigs), %.1f s; oracle/coverm_oracle.c (%s) = literal C port of "
                                              "CoverM 0.8.0's scan loop + estimators, records already decoded in host memory; not the coverm binary"
                                              % (R, cpu_dt, oracle_native()[1]))
            # ---- basis (ii): records in page-locked host memory -> push -> finish -> fetch -> finalise
            hb, ptrs = pinned_copy(batch)
            s2 = Session(local_rank, FilterConfig(), 75, want_hist, want_id)
            s2.set_targets(ref.lengths)
            if dev_est:
                s2.set_estimators(est)
            push_s = []
            for it in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s2.reset()
                s2.push(hb)
                st2, su2 = s2.finish()
                if dev_est:
                    finalise(ref, est, st2, su2, None, estimates=s2.estimates())
                else:
                    finalise(ref, est, st2, su2, s2.hist())
                push_s.append(time.perf_counter() - t0)
            s2.close()
            L = native.lib()
            for p in ptrs:
                L.cov_host_free(p)
            best = min(push_s[1:])
            out["bases"] = {
                "device_resident": {"gpu_reads_per_s": total_reads * a.steps / elapsed, "ms": elapsed / a.steps * 1e3,
                                    "cpu_reads_per_s": cpu_rate, "cpu": "oracle scan, 1 thread, records in host memory"},
                "push_inclusive": {"gpu_reads_per_s": considered / best, "ms": best * 1e3, "bytes_over_pcie": int(pipe_bytes - len(ref.lengths) * 160),
                                   "pcie_GBps": (pipe_bytes - len(ref.lengths) * 160) / best / 1e9,
                                   "cpu_reads_per_s": cpu_rate, "cpu": "oracle scan, 1 thread, records in host memory (same starting point)",
                                   "speedup_vs_cpu": considered / best / cpu_rate},
            }
            if not a.no_binary_legs and not leg_allowed("binary_configs", a, world, out):
                pass
            elif not a.no_binary_legs:
                try:
                    out["binary_configs"] = binary_config_legs(a, threads, ref, batch, oracle_cfg2)
                    if not out["binary_configs"].get("tables_equal", False):
                        exit_code = 3
                except Exception as ex:
                    out["binary_configs"] = {"error": repr(ex)[:1000]}
            if not a.no_e2e and world == 1 and not leg_allowed("end_to_end", a, world, out["bases"]):
                pass
            elif not a.no_e2e and world == 1:
                try:
                    out["bases"]["end_to_end"] = end_to_end(a, threads)
                    if not out["bases"]["end_to_end"].get("tables_equal", False):
                        exit_code = 3
                except Exception as ex:   # the headline legs above stand on their own
                    out["bases"]["end_to_end"] = {"error": repr(ex)[:1000]}
        fake = os.environ.get("COVERM_BENCH_MULTI_DEVICE_CHECK")     # e.g. "0,0": the --devices legs on a single-GPU box (functional check, not a measurement)
        if fake and world == 1:
            try:
                out["multi_device_end_to_end_functional_check"] = multi_device_legs(a, max(1, usable_cpus()), len(fake.split(",")), devs=fake)
                if not out["multi_device_end_to_end_functional_check"].get("tables_equal", False):
                    exit_code = 3
            except Exception as ex:
                out["multi_device_end_to_end_functional_check"] = {"error": repr(ex)[:1000]}
                exit_code = 3
        if world > 1 and not a.no_multi_device_e2e and not share and not leg_allowed("multi_device_end_to_end", a, world, out):
            pass
        elif world > 1 and not a.no_multi_device_e2e and not share:
            # the product's own multi-GPU path (one process, N devices): run by rank 0 while the other ranks wait on the host (wait_for_root below)
            try:
                threads = max(1, int(os.environ.get("COVERM_BENCH_THREADS", usable_cpus())))
                out["multi_device_end_to_end"] = multi_device_legs(a, threads, world)
                if not out["multi_device_end_to_end"].get("tables_equal", False):
                    exit_code = 3
            except Exception as ex:
                out["multi_device_end_to_end"] = {"error": repr(ex)[:1000]}
        # The scalars a reader of the record needs, as the LAST keys of the line (a log that keeps only the tail of the line keeps these):
        # end to end (BAM file -> TSV, config-5 size and flags) at BGZF level 1 and 6 — GPU median of three, CPU best of three, x = GPU rate
        # over the CPU's decode and scan perfectly overlapped —, configs 2 and 3 through the binary, and whether every table compared equal.
        e2e = (out.get("bases") or {}).get("end_to_end") or {}
        l6 = e2e.get("level6") or {}
        bc = out.get("binary_configs") or {}
        r3 = lambda x: None if x is None else round(float(x), 3)
        out.update({
            "e2e_reads": e2e.get("reads"),
            "e2e_l1_s": r3((e2e.get("gpu") or {}).get("seconds")), "e2e_l1_x_overlapped": r3(e2e.get("speedup_vs_cpu_overlapped")),
            "e2e_l1_x_serial": r3(e2e.get("speedup_vs_cpu_serial")),
            "e2e_l1_one_process_s": r3((e2e.get("gpu") or {}).get("one_process_seconds")), "e2e_l1_x_overlapped_one_process": r3(e2e.get("speedup_vs_cpu_overlapped_one_process")),
            "e2e_l6_s": r3(l6.get("gpu_seconds")), "e2e_l6_x_overlapped": r3(l6.get("speedup_vs_cpu_overlapped")), "e2e_l6_x_serial": r3(l6.get("speedup_vs_cpu_serial")),
            "cpu_decode_s": r3((e2e.get("cpu") or {}).get("decode_s")), "cpu_decode_l6_s": r3(l6.get("cpu_decode_s")), "cpu_scan_s": r3((e2e.get("cpu") or {}).get("scan_s")),
            "cpu_threads": e2e.get("threads"), "cpu_inflate_backend": (e2e.get("cpu") or {}).get("inflate_backend"),
            "cfg2_binary_s": r3((bc.get("config2_contig") or {}).get("seconds")), "cfg3_binary_s": r3((bc.get("config3_genome") or {}).get("seconds")),
            "parity_equal": (out.get("parity_checked") or {}).get("equal"),
            "tables_equal": (bool(e2e.get("tables_equal")) and bool(bc.get("tables_equal"))) if (e2e and bc and "error" not in e2e and "error" not in bc and "skipped" not in e2e and "skipped" not in bc) else None,
            "device_resident_ms": r3(out["ms_per_step"]), "roofline_kernel": roof.get("kernel"), "roofline_frac": round(float(roof.get("frac") or 0.0), 4),
            "pileup_kernel_ms": r3((roof.get("all_kernels_ms") or {}).get("k_pileup")), "pileup_hbm_frac": round(float(((roof.get("kernels") or {}).get("k_pileup") or {}).get("hbm_frac_of_8TBps") or 0.0), 4),
            "prep_kernel_ms": r3((roof.get("all_kernels_ms") or {}).get("k_prep")), "prep_hbm_frac": round(float(((roof.get("kernels") or {}).get("k_prep") or {}).get("hbm_frac_of_8TBps") or 0.0), 4),
        })
        print(json.dumps(out), flush=True)
    if dist:
        # rank 0 comes here after the product's multi-device legs; the others wait for it on the host (a key in the group's store), not
        # inside a collective: an RCCL barrier would spin on their GPUs while coverm-amd --devices uses them
        from coverm_amd import distributed as cdist
        cdist.wait_for_root(dist, rank, "coverm_bench_legs_done")
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(exit_code)


# ---- time budget: what a leg is expected to cost on a lease box with ~16 usable CPUs, from the rounds' own runs
# (generation of a 50 M-read sample ~25 s, of the 200 M-read one ~100 s + ~60 s per BGZF level to write it; CPU decode legs 3 x 9 s per level)
BUDGET = {"t0": time.time(), "limit": 1500.0}


def leg_estimates(a, world):
    scale = a.reads / 50e6
    e2e = a.e2e_reads / 200e6
    est = {"sample + device-resident steps": 45 * scale + 10}
    if world == 1 and not a.no_cpu_baseline:
        est["parity + cpu baseline + push-inclusive"] = 40 * scale
        if not a.no_binary_legs:
            est["binary_configs"] = 75 * scale
        if not a.no_e2e:
            est["end_to_end"] = (215 if a.no_level6 else 370) * e2e      # (five runs of the binary per level + two as one process)
    if world > 1 and not a.no_multi_device_e2e:
        est["multi_device_end_to_end"] = (60 + 35 * world) * scale + 260 * e2e
    return est


def print_plan(a, world, share):
    est = leg_estimates(a, world)
    tot = sum(est.values())
    print("[bench] time limit %.0f s; plan for N = %d%s: %s = ~%.0f s%s" % (
        BUDGET["limit"], world, " (ranks share one GPU: functional check)" if share else "", ", ".join("%s ~%.0f s" % kv for kv in est.items()), tot,
        "" if tot <= BUDGET["limit"] else " -- MORE than the limit: the last legs will be skipped and recorded as skipped"), file=sys.stderr, flush=True)


def leg_allowed(name, a, world, record):
    """True when `name` can still finish inside the limit; else writes the reason into record[name] and says so on stderr."""
    need = leg_estimates(a, world).get(name, 0.0)
    left = BUDGET["limit"] - (time.time() - BUDGET["t0"])
    if need * 1.15 <= left:
        return True
    record[name] = {"skipped": "time budget", "estimate_s": round(need, 1), "left_s": round(left, 1), "time_limit_s": BUDGET["limit"]}
    print("[bench] skipping %s: needs ~%.0f s, %.0f s of the %.0f s limit are left" % (name, need, left, BUDGET["limit"]), file=sys.stderr, flush=True)
    return False


def relaunch_cmd(n, argv, port=None):
    """The command `python bench.py --gpus N` turns itself into when no launcher started it: N ranks on this node, rendezvous on
    127.0.0.1 (the container hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def check_devices(n, share):
    """N ranks need N GPUs: fewer is an error, not a silently smaller run.  COVERM_BENCH_SHARE_GPU=1 (functional check of the N > 1
    path on a single-GPU box; never a measurement) puts every rank on device 0."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the coverage engine has no CPU fallback")
    have = torch.cuda.device_count()
    if have < n and not share:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (COVERM_BENCH_SHARE_GPU=1 runs the ranks on one device as a functional check)" % (n, have))


def default_tmp():
    try:
        if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (64 << 30):
            return "/dev/shm"
    except Exception:
        pass
    return tempfile.gettempdir()


def usable_cpus():
    """CPUs this process may really use: affinity mask and the cgroup CPU quota (the GPU lease boxes show 256 logical CPUs but
    grant 16 through cpu.max; more decoder threads than that only add contention)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def sess_tiles(sess, ref):
    """Tiles of the pileup kernels: 1024 bases per wave."""
    tile = 1024
    return int(((ref.lengths + tile - 1) // tile).sum())


def ingest_roofline():
    """The device ingest's dominant kernels against the HBM roofline, from the newest COMMITTED ingest profile (the files are named in
    `source`: profiles/r06_ingest_kernel_stats.csv — rocprofv3 --kernel-trace --stats of `coverm-amd contig` over a 20 M-read level-1 BAM, run
    with rounds of 81 920 BGZF blocks: one full round + one of 12 382 — and profiles/r06_ingest_pmc_summary.json — separate --pmc passes).
    The library's default round has been 61 440 blocks since the end of round 6; the figures here are per 81 920, as profiled.  Not
    measured by this run: the bench's timed region is the coverage path; labelled as such.  Algorithmic bytes per full round: compressed
    bytes read once + inflated bytes written once (k_inflate_wave); token positions + every match byte read and written once (k_lz_stage;
    k_lz_resolve in profiles that predate it); the inflated bytes read once (k_crc32_wave)."""
    out = {"source": "no committed ingest profile found",
           "round_blocks": 81920, "peak_GBps": HBM_PEAK_GBPS}
    try:
        import csv
        full_ms = {}
        stats = next(p for p in ("r06_ingest_kernel_stats.csv", "r05_ingest_kernel_stats.csv", "r04_ingest_kernel_stats.csv") if os.path.exists(os.path.join(ROOT, "profiles", p)))
        pmc = next(p for p in ("r06_ingest_pmc_summary.json", "r05_ingest_pmc_summary.json", "r04_ingest_pmc_summary.json") if os.path.exists(os.path.join(ROOT, "profiles", p)))
        out["source"] = "committed profile (profiles/%s, %s), not this run" % (stats, pmc)
        with open(os.path.join(ROOT, "profiles", stats)) as fh:
            for r in csv.DictReader(fh):
                full_ms[r["Name"].split("(")[0].replace("void ", "").strip()] = float(r["MaxNs"]) / 1e6      # the full round is the longer of the two launches
        with open(os.path.join(ROOT, "profiles", pmc)) as fh:
            pm = json.load(fh).get("derived_full_round", {})
        blocks = 81920
        algo = {"covi::k_inflate_wave": blocks * (21100 + 62900), "covi::k_lz_stage": blocks * (5900 * 2 + 2 * 50600), "covi::k_lz_resolve": blocks * (5900 * 2 + 2 * 50600),
                "covi::k_crc32_wave": blocks * 65280}
        for k, b in algo.items():
            ms = full_ms.get(k)
            if not ms:
                continue
            ach = b / (ms * 1e-3) / 1e9
            out[k.split("::")[1]] = {"ms_per_full_round": ms, "algorithmic_bytes": b, "achieved_GBps": ach, "frac": ach / HBM_PEAK_GBPS,
                                     "hbm_read_bytes_counters": (pm.get(k) or {}).get("hbm_read_bytes_gfx950_corrected"),
                                     "hbm_write_bytes_counters": (pm.get(k) or {}).get("hbm_write_bytes")}
    except Exception as ex:
        out["error"] = repr(ex)[:300]
    return out


def profile_numbers():
    """Per-kernel HBM bytes and pipe utilisation from the committed rocprofv3 PMC summary of this workload (profiles/), if present.
    They describe the committed build, not this run — labelled as such in the JSON."""
    out = {}
    for name, key in (("pmc_traffic.json", "traffic"), ("pmc_pipes.json", "pipes")):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                out[key] = json.load(fh)
        except Exception:
            out[key] = {}
    return out


if __name__ == "__main__":
    main()
