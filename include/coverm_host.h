/*
 * coverm_host.h — host side ABOVE the covermhip C ABI, in C++ with a C surface.
 *
 * The reference keeps these layers in Rust; no Rust toolchain exists in this build environment, so
 * they are written in C++ with the same names, argument meaning and error behaviour:
 *
 *   CoverageEstimator::{new_estimator_*, calculate_coverage, print_coverage, print_zero_coverage}
 *                                   src/mosdepth_genome_coverage_estimators.rs:107-224, 530-839, 936-991
 *   contig_coverage                 src/contig.rs:13-253           (flush + zero-row logic :40-104, 255-277)
 *   mosdepth_genome_coverage_with_contig_names   src/genome.rs:17-322
 *   mosdepth_genome_coverage        src/genome.rs:419-929
 *   CoverageTaker implementations   src/coverage_takers.rs:74-219, 265-377
 *   CoveragePrinter                 src/coverage_printer.rs:20-553
 *   EstimatorsAndTaker / FilterParameters  src/bin/coverm.rs:1315-1504, 1648-1704
 *   BAM/SAM reading (rust-htslib in the reference)  src/bam_generator.rs:103-144, 356-371
 *
 * The scan loops here do not touch reads: they consume the per-contig integer statistics a
 * covermhip session produced (cov_contig_stats + histograms) where the reference would have called
 * add_contig(ups_and_downs, ...).  Every f32 is formed from those integers with the reference's
 * operation order, so results are bit-identical, not merely within tolerance.
 */
#ifndef COVERM_HOST_H
#define COVERM_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "covermhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* enum CoverageEstimator variant order, estimators.rs:4-81 */
typedef enum {
    COVH_MEAN = 0,
    COVH_TRIMMED_MEAN = 1,
    COVH_PILEUP_COUNTS = 2,
    COVH_COVERED_FRACTION = 3,
    COVH_COVERED_BASES = 4,
    COVH_RPKM = 5,
    COVH_TPM = 6,
    COVH_VARIANCE = 7,
    COVH_LENGTH = 8,
    COVH_READ_COUNT = 9,
    COVH_READS_PER_BASE = 10,
    COVH_ANIR = 11
} covh_kind;

/* Constructor parameters (estimators.rs:107-224). */
typedef struct {
    int32_t kind;
    float min_fraction_covered_bases;
    uint64_t contig_end_exclusion; /* must equal the session's cov_config.contig_end_exclusion */
    int32_t exclude_mismatches;
    float trim_min, trim_max;
} covh_estimator;

typedef struct {
    uint64_t num_mapped_reads, num_reads; /* ReadsMapped, lib.rs:54-57 */
} covh_reads_mapped;

/* One sample's device results, as returned by cov_finish / cov_fetch_hist. */
typedef struct {
    const char *stoit_name;
    const cov_contig_stats *stats; /* n_targets entries */
    const uint64_t *hist;          /* may be NULL when no estimator needs histograms */
    uint64_t num_detected_primary_alignments;
} covh_sample;

/* BAM header as the scan loops use it (names + lengths). */
typedef struct {
    uint32_t n_targets;
    const char *names;        /* concatenated, not NUL separated */
    const uint32_t *name_off; /* n_targets + 1 */
    const uint64_t *target_len;
} covh_header;

typedef enum {
    COVH_TAKER_STREAM = 0, /* SingleFloatCoverageStreamingCoveragePrinter */
    COVH_TAKER_PILEUP = 1, /* PileupCoverageCoveragePrinter */
    COVH_TAKER_CACHED = 2  /* CachedSingleFloatCoverageTaker */
} covh_taker_kind;

typedef struct covh_taker covh_taker;     /* owns an output text buffer */
covh_taker *covh_taker_new(int kind, size_t num_coverages);
void covh_taker_free(covh_taker *t);
const char *covh_taker_text(const covh_taker *t, size_t *len); /* streamed output so far */
void covh_taker_clear_text(covh_taker *t);
/* Cached taker: the f32 coverages recorded for sample `stoit`, in recording order (entries x estimators);
 * returns the number available (copies at most cap). */
size_t covh_taker_cached_coverages(const covh_taker *t, size_t stoit, float *out, size_t cap);

/* trait CoverageTaker (coverage_takers.rs:29-38): start_stoit once per BAM; per entry start_entry -> add_* (one per
 * estimator, in estimator order) -> finish_entry.  The scan drivers below make these calls themselves. */
void covh_taker_start_stoit(covh_taker *t, const char *stoit_name);
void covh_taker_start_entry(covh_taker *t, size_t entry_order_id, const char *entry_name);
void covh_taker_add_single_coverage(covh_taker *t, float coverage);
void covh_taker_add_coverage_entry(covh_taker *t, uint64_t num_reads, uint64_t num_bases);
void covh_taker_finish_entry(covh_taker *t);
int covh_taker_names_mismatch(const covh_taker *t); /* an entry id arrived with two different names (:140-148) */
/* Cached taker: CoverageTakerTypeIterator (coverage_takers.rs:265-377), flattened; returns the item count. */
size_t covh_taker_iterate(const covh_taker *t, uint64_t *entry_index, uint64_t *stoit_index, float *coverages, size_t cap);

/* Which need a histogram / identity sums from the device (COV_WANT_* for the session). */
uint32_t covh_wants(const covh_estimator *est, size_t n_est);

/* The three scan entry points.  Return COV_OK or a cov_status; *err (optional) receives a static message. */
int covh_contig_coverage(const covh_header *h, const covh_sample *samples, size_t n_samples, covh_taker *taker,
                         const covh_estimator *est, size_t n_est, int print_zero_coverage_contigs,
                         covh_reads_mapped *reads_mapped_out);
/* The same loop with calculate_coverage already evaluated on the device: estimates[i] = the n_targets x n_est floats cov_fetch_estimates
 * returned for sample i (after cov_set_estimators with these estimators), or NULL for a sample to evaluate here; estimates itself may be
 * NULL.  The floats are the same bit for bit, so rows, zero rows and ReadsMapped (contig.rs:40-104) are those of covh_contig_coverage.
 * COVH_PILEUP_COUNTS and COVH_TPM are never evaluated on the device: COV_ERR_INVALID_ARG when floats are given with one of them. */
int covh_contig_coverage_estimated(const covh_header *h, const covh_sample *samples, size_t n_samples, covh_taker *taker,
                                   const covh_estimator *est, size_t n_est, int print_zero, covh_reads_mapped *rm_out,
                                   const float *const *estimates);
int covh_genome_coverage_with_contig_names(const covh_header *h, const covh_sample *samples, size_t n_samples,
                                           const int32_t *genome_of_tid, const char *const *genome_names,
                                           size_t n_genomes, covh_taker *taker, int print_zero_coverage_genomes,
                                           const covh_estimator *est, size_t n_est,
                                           covh_reads_mapped *reads_mapped_out);
int covh_genome_coverage_separator(const covh_header *h, const covh_sample *samples, size_t n_samples,
                                   uint8_t split_char, covh_taker *taker, int print_zero_coverage_genomes,
                                   const covh_estimator *est, size_t n_est, int single_genome,
                                   covh_reads_mapped *reads_mapped_out);
const char *covh_last_error(void);

/* CoveragePrinter (coverage_printer.rs).  printer: 0 streamed, 1 sparse cached, 2 dense cached, 3 MetaBAT.
 * Appends to the taker's text buffer. */
void covh_print_headers(covh_taker *t, int printer, const char *entry_type, const char *const *headers, size_t n);
void covh_finalise_printing(covh_taker *t, int printer, const char *entry_type, const char *const *headers,
                            size_t n_headers, const covh_reads_mapped *reads_mapped, size_t n_samples,
                            const int64_t *columns_to_normalise, size_t n_norm, int64_t rpkm_column,
                            int64_t tpm_column);

/* Rust `Display` for f32 / f64 (shortest round-trip, positional notation).  Returns bytes written. */
size_t covh_format_f32(float v, char *buf, size_t cap);
size_t covh_format_f64(double v, char *buf, size_t cap);

/* ---- BGZF / BAM / SAM reader (rust-htslib's role: bam_generator.rs:113-119, 125-129, 356-371).
 * Decodes the whole file with `threads` inflate/parse threads and exposes the records as a cov_batch whose
 * arrays stay owned by the handle.  want_names also keeps read names (pair-mode filtering, filter.rs:164). */
typedef struct covh_bam covh_bam;
covh_bam *covh_bam_open(const char *path, int threads, int want_names, char *err, size_t errcap);
void covh_bam_close(covh_bam *h);
/* Keep the largest inflate buffer mapped between files (off by default; turning it off frees it).  A tool that
 * reads many BAMs in a row saves the unmap + first-touch of gigabytes per file. */
void covh_bam_set_buffer_cache(int on);
/* Decode record arrays into page-locked memory from cov_host_alloc (off by default; falls back to ordinary memory
 * when no device is usable), so that cov_push_batch is a plain DMA. */
void covh_bam_set_pinned(int on);
/* 1: covh_bam_gpu_ingest* gives its page-locked staging slots back to the system as soon as their last uploads are through (a session
 * that reads no further file; page-locked memory still held at process exit costs ~0.13 s per GiB).  Default 0: the slots are parked for
 * the next file. */
void covh_bam_set_release_staging(int on);
/* 1 when COVERM_CLI_TIMING is set: the library's and the host layer's timing stamps go to stderr (read once per process). */
int covh_timing_on(void);
/* How many device ingests (covh_bam_gpu_ingest*) the caller runs at once, one per GPU.  With more than two and no COVERM_INGEST_IO the
 * file is mapped and its span registered with the device once, up front, instead of copied through page-locked staging slots: N feeders
 * share one host memory system, and a staged byte crosses it three times, a mapped one once (unmeasured on more
 * than one device).  Default 1. */
void covh_bam_set_concurrent_feeders(int n);
uint32_t covh_bam_n_targets(const covh_bam *h);
const char *covh_bam_target_name(const covh_bam *h, uint32_t i);
uint64_t covh_bam_target_len(const covh_bam *h, uint32_t i);
uint64_t covh_bam_n_records(const covh_bam *h);
uint64_t covh_bam_n_cigar(const covh_bam *h);
void covh_bam_batch(const covh_bam *h, cov_batch *out);
const int32_t *covh_bam_mtid(const covh_bam *h);
const uint32_t *covh_bam_qname_off(const covh_bam *h); /* n_records + 1, NULL without want_names */
const char *covh_bam_qnames(const covh_bam *h);
/* Writes a batch as a coordinate-sorted BGZF BAM (benchmark/test inputs), NM typed C/S/I by magnitude.  with_seq: 0 = SEQ
 * '*'; 1 = read names r<i>, SEQ all 'A', QUAL 0xff (compresses ~18x: inflate cost far below a real BAM's); 2 = realistic
 * entropy: random bases, Phred-like binned qualities, Illumina-style names (~65 compressed bytes per 150 bp read).
 * Chunked, bounded memory.  Returns 0 on success. */
int covh_bam_write(const char *path, uint32_t n_targets, const char *const *names, const uint64_t *lens,
                   const cov_batch *batch, int with_seq, int level, int threads);

/* ---- streamed reader: the same decode window by window with bounded memory (three ~32 MiB-compressed windows and three
 * page-locked SoA batches circulate), so that inflate, record parsing and the H2D copy of finished batches overlap inside one
 * file.  BAM only (BGZF); no read names (pair-mode filtering and --gff use covh_bam_open).  span_count > 1 selects the
 * span_index-th of span_count tid ranges, cut where the tid changes nearest to k/span_count of the file; the rank's first
 * BGZF block is located by probing the file, so each rank inflates only its own part.  A sorted BAM's records
 * fall into exactly one span each (records with tid -1 go to the last one).
 *   h = covh_bam_stream_open(path, threads, 0, 1, err, cap);   header accessors are valid at once
 *   while (covh_bam_stream_next(h, &batch) == 1) cov_push_batch(session, &batch);   arrays stay valid until the next call
 * covh_bam_stream_next returns 1 (batch), 0 (end of file / span), -1 (covh_bam_stream_error) or -2 (the same, and the error is
 * "the record keys decrease inside this span": see covh_bam_gpu_ingest_span). */
typedef struct covh_bam_stream covh_bam_stream;
covh_bam_stream *covh_bam_stream_open(const char *path, int threads, uint32_t span_index, uint32_t span_count, char *err, size_t errcap);
uint32_t covh_bam_stream_n_targets(const covh_bam_stream *h);
const char *covh_bam_stream_target_name(const covh_bam_stream *h, uint32_t i);
uint64_t covh_bam_stream_target_len(const covh_bam_stream *h, uint32_t i);
int covh_bam_stream_next(covh_bam_stream *h, cov_batch *out);
const char *covh_bam_stream_error(const covh_bam_stream *h);
/* Which inflate the HOST readers of this process use (covh_bam_open, covh_bam_stream_*, coverm-amd filter): "libdeflate" when
 * libdeflate.so.0 could be bound at run time, else "zlib" — so that a CPU baseline timed with them can say what it was timed on. */
const char *covh_inflate_backend(void);
uint64_t covh_bam_stream_n_records(const covh_bam_stream *h);  /* records handed out so far */
uint64_t covh_bam_stream_peak_bytes(const covh_bam_stream *h); /* largest total of window + batch buffers held */
void covh_bam_stream_timing(const covh_bam_stream *h, double *out5); /* s: read, inflate, parse, inflate-side wait, parse-side wait */
void covh_bam_stream_close(covh_bam_stream *h);

/* ---- device ingest driver (cov_ingest_* of covermhip.h): header on the host, everything else on the GPU.
 *   hd = covh_bam_read_header(path, ...)  -> reference names / lengths (cov_set_targets) and where the records start
 *   covh_bam_gpu_ingest(path, threads, session, hd, check_crc, &n, timing, err, cap)
 *       0 = the records are in the session's store; 1 = this file needs the CPU reader (reason in err, nothing appended:
 *       use covh_bam_stream_* / covh_bam_open); -1 = error.  threads read the file into page-locked staging buffers. */
typedef struct covh_bam_header covh_bam_header;
covh_bam_header *covh_bam_read_header(const char *path, char *err, size_t errcap);
void covh_bam_header_free(covh_bam_header *h);
uint32_t covh_bam_header_n_targets(const covh_bam_header *h);
const char *covh_bam_header_target_name(const covh_bam_header *h, uint32_t i);
uint64_t covh_bam_header_target_len(const covh_bam_header *h, uint32_t i);
uint64_t covh_bam_header_first_record(const covh_bam_header *h); /* offset in the inflated stream */
/* covh_bam_gpu_ingest_span: one tid span of the file (span definition of covh_bam_stream_open): only the span's part of the file
 * is read and fed; 0 with *n_records = 0 for an empty span.  -2 (instead of -1) when the record keys decrease inside the span: a span
 * trusts the file's order, so the caller decides (coverm-amd then sends the file through one device whole, where the reference's
 * own rule, contig.rs:118-132, judges it). */
int covh_bam_gpu_ingest_span(const char *path, int threads, cov_session *s, const covh_bam_header *hd, int check_crc, uint32_t span_index,
                             uint32_t span_count, uint64_t *n_records, double *timing8, char *err, size_t errcap);
int covh_bam_gpu_ingest(const char *path, int threads, cov_session *s, const covh_bam_header *hd, int check_crc, uint64_t *n_records,
                        double *timing8, char *err, size_t errcap);   /* s: file read, staging waits, cov_ingest_end, total, buffers, block-header walk, cov_ingest_feed; [7] = where the DMA read the bytes (0 staging slots, 1 the mapped file, 2 mapped and registered up front) */

/* ---- reader-stage PAIR filter (ReferenceSortedBamFilter::read pair branch, filter.rs:117-228, filter_out = true).
 * The single-read branch runs on the device (cov_config.filter_single); the pair branch needs read names, which never
 * cross the covermhip ABI, so it runs here, threaded over references.  Thresholds as FilterParameters holds them
 * (coverm.rs:1648-1704); filter_single = ReferenceSortedBamFilter::filter_single (filter.rs:48-55). */
typedef struct {
    int32_t filter_single;
    uint8_t min_mapq; /* 255 = off */
    uint32_t min_aligned_length_single;
    float min_percent_identity_single, min_aligned_percent_single;
    uint32_t min_aligned_length_pair;
    float min_percent_identity_pair, min_aligned_percent_pair;
} covh_pair_filter;
/* Indices of the records the reference's filter would return, in its order (first mate then second mate, pairs
 * ordered by where the second mate sits in the file).  *order_out is released with covh_free.  Returns COV_OK, or
 * COV_ERR_NM_MISSING / COV_ERR_NM_BADTYPE where the reference's nm() would have panicked. */
int covh_pair_mode_order(const cov_batch *b, const int32_t *mtid, const uint32_t *qname_off, const char *qnames,
                         const covh_pair_filter *f, int threads, uint64_t **order_out, uint64_t *n_out);
/* ReferenceSortedBamFilter::read as a whole (filter.rs:84-228), serial: the single-read branch when f->filter_single && !filter_pairs,
 * the pair branch otherwise; filter_out = 0 is `coverm filter --inverse` (unmapped records and — in the pair branch — improper pairs are
 * returned, and a record / pair is returned iff it FAILS its judgement).  The selection of the `filter` subcommand. */
int covh_reader_filter_order(const cov_batch *b, const int32_t *mtid, const uint32_t *qname_off, const char *qnames, const covh_pair_filter *f,
                             int filter_pairs, int include_supplementary, int include_secondary, int filter_out, uint64_t **order_out, uint64_t *n_out);
/* `coverm filter` (bin/coverm.rs:408-472): the records of the BAM file `in_path` that the reader filter returns, byte for byte (names,
 * bases, qualities, tags) and in its order, under the input's header, as a new BGZF-compressed BAM.  Host code, like the reference's.
 * Returns 0, or -1 with a message. */
int covh_bam_filter_file(const char *in_path, const char *out_path, const covh_pair_filter *f, int filter_pairs, int include_supplementary,
                         int include_secondary, int filter_out, int level, int threads, uint64_t *n_in, uint64_t *n_out, char *err, size_t errcap);
void covh_free(void *p);
/* Gathers records order[0..n) of src into a new batch (page-locked memory when a device is usable). */
int covh_batch_select(const cov_batch *src, const uint64_t *order, uint64_t n, int threads, cov_batch *out);
void covh_batch_free(cov_batch *b);

/* ---- per-gene coverage (--gff; src/genes.rs:182-567).  The contig's depth comes through a callback (cov_copy_depth of a
 * finished session in the product); GFF parsing, gene resolution, per-gene window statistics and per-gene read aggregates
 * (reads are assigned to the gene their leftmost position falls in) are this host code.  One call per BAM. */
typedef struct covh_genes covh_genes;
covh_genes *covh_genes_read_gff(const char *path, const char *feature_type /* NULL = every feature */, char *err, size_t errcap);
covh_genes *covh_genes_from_arrays(const char *const *ids, const char *const *contigs, const uint64_t *start,
                                   const uint64_t *end, size_t n); /* 0-based half-open */
size_t covh_genes_count(const covh_genes *g);
void covh_genes_get(const covh_genes *g, size_t i, const char **id, const char **contig, uint64_t *start, uint64_t *end);
void covh_genes_free(covh_genes *g);
typedef int (*covh_depth_fn)(void *ctx, uint32_t tid, int32_t *depth_out); /* target_len[tid] entries; COV_OK or a status */
typedef struct {
    int32_t mode; /* 0 no genome column (contig mode), 1 single genome "genome1", 2 prefix before `separator`, 3 table */
    uint8_t separator;
    const int32_t *genome_of_tid;    /* mode 3: -1 = contig in no genome (its genes are not reported) */
    const char *const *genome_names; /* mode 3 */
} covh_genome_namer;
/* records = what the scan sees (after a pair-mode reader stage, if any); cfg carries the flag filter, the single-read
 * thresholds (filter_single) and nothing else that matters here; num_detected_primary_alignments as the reader counted it.
 * device_session (a finished session holding these records) selects the device reductions (cov_interval_stats_compute: one
 * wave per gene over the depth kept in HBM); with NULL the depth callback is used and the reductions run on the host. */
int covh_gene_coverage(const covh_header *h, const covh_genes *genes, const covh_genome_namer *namer, const char *stoit_name,
                       const cov_batch *records, const cov_config *cfg, cov_session *device_session, covh_depth_fn depth,
                       void *depth_ctx, uint64_t num_detected_primary_alignments, covh_taker *taker, const covh_estimator *est,
                       size_t n_est, int print_zero_coverage_genes, covh_reads_mapped *reads_mapped_out);

/* ---- the orchestrator of `coverm contig|genome --bam-files ...` (src/bin/coverm.rs:1315-1704, 2088-2131, 1539-1628): argv as the
 * coverm-amd binary takes it (argv[1] = "contig" | "genome", the reference's flag names, plus --device N / --devices a,b,...
 * and --no-stream).  Writes the table to --output-file or stdout; returns the process exit code (1 after printing an error). */
int covh_cli_main(int argc, char **argv);
/* For a process that exits right after covh_cli_main (the coverm-amd binary): leave the sessions to the OS instead of
 * destroying them one allocation at a time. */
void covh_cli_set_fast_exit(int on);

/* ---- trait MosdepthGenomeCoverageEstimator (estimators.rs:245-265), one export per method, for hosts that keep the reference's
 * scan-loop shape (a Rust `impl MosdepthGenomeCoverageEstimator` forwards 1:1).  add_contig takes the contig's
 * integer statistics from cov_finish where the reference passes the delta array.  `hist` is copied by the call (the bins of this
 * contig only): a binding may hand in a temporary buffer, as the reference's add_contig borrows its slice for the call alone. */
typedef struct covh_estimator_state covh_estimator_state;
covh_estimator_state *covh_estimator_new(const covh_estimator *params);                 /* CoverageEstimator::new_estimator_* */
void covh_estimator_free(covh_estimator_state *s);
void covh_estimator_setup(covh_estimator_state *s);                                     /* fn setup(&mut self) */
void covh_estimator_add_contig_stats(covh_estimator_state *s, const cov_contig_stats *stats, uint64_t target_len,
                                     const uint64_t *hist /* cov_fetch_hist array or NULL */, uint64_t num_mapped_reads,
                                     double sum_identity);                              /* fn add_contig(&mut self, ...) */
float covh_estimator_calculate_coverage(covh_estimator_state *s, const uint64_t *unobserved_contig_lengths, size_t n);
void covh_estimator_print_coverage(const covh_estimator_state *s, float coverage, covh_taker *t);
void covh_estimator_print_zero_coverage(const covh_estimator_state *s, covh_taker *t, uint64_t entry_length);
covh_estimator_state *covh_estimator_copy(const covh_estimator_state *s);               /* fn copy(&self) */
uint64_t covh_estimator_num_mapped_reads(const covh_estimator_state *s);

/* calculate_coverage for one entry built from explicit sums (used by unit tests). */
typedef struct {
    uint64_t win_len, win_sum_d, win_sum_d2, win_covered, full_len, full_covered, n_reads, mismatches;
    uint32_t win_min_d;
    uint32_t hist_len;
    const uint64_t *hist;
    double sum_identity;
} covh_entry;
float covh_calculate_coverage(const covh_estimator *e, const covh_entry *entry, const uint64_t *unobserved,
                              size_t n_unobserved);

#ifdef __cplusplus
}
#endif
#endif
