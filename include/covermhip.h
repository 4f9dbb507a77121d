/*
 * covermhip.h — C ABI of the MI355X-native coverage engine (libcovermhip.so).
 *
 * This is the drop-in boundary for CoverM's BAM -> pileup -> per-contig hot path.  The reference
 * (wwood/CoverM v0.8.0) has no FFI of its own; the seam is the body of its three scan loops
 *   contig_coverage                               src/contig.rs:13-253
 *   mosdepth_genome_coverage_with_contig_names    src/genome.rs:17-322
 *   mosdepth_genome_coverage                      src/genome.rs:419-797
 * and of `CoverageEstimator::add_contig` (src/mosdepth_genome_coverage_estimators.rs:366-528).
 * A host (Rust via `extern "C"`, C++, Python/ctypes) keeps decoding BAM on the CPU and keeps
 * `calculate_coverage` / `CoverageTaker` untouched; it streams packed record batches through this
 * ABI and receives, per contig, the exact integer sufficient statistics from which every
 * estimator's f32 is finalised (INTEGRATION.md shows the Rust binding).
 *
 * Plain C, POD only, caller-owned buffers.  Every function returns a cov_status; on failure
 * cov_last_error() describes it.  A session is bound to one HIP device and one sample (BAM); it is
 * not thread-safe (single producer), and owns its own HIP stream.
 */
#ifndef COVERMHIP_H
#define COVERMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COVERMHIP_ABI_VERSION 1

typedef struct cov_session cov_session;

/* Status codes.  The UNSORTED / NM_* / POS_OOB codes map 1:1 onto the reference's panics. */
typedef enum {
    COV_OK = 0,
    COV_ERR_UNSORTED = 1,   /* contig.rs:129-132, genome.rs:133-136, 549-552 */
    COV_ERR_NM_MISSING = 2, /* lib.rs:149-156 */
    COV_ERR_NM_BADTYPE = 3, /* lib.rs:144-147 */
    COV_ERR_POS_OOB = 4,    /* index panic at contig.rs:178 (M/=/X run starting at or past contig end) */
    COV_ERR_BAD_CIGAR = 6,  /* CIGAR op code > 8 */
    COV_ERR_BAD_TID = 7,    /* considered record whose tid is outside the header ("Corrupt BAM file?", contig.rs:145) */
    COV_ERR_INVALID_ARG = 16,
    COV_ERR_HIP = 17,       /* HIP runtime failure or no usable device: the engine never falls back to the CPU */
    COV_ERR_STATE = 18,
    COV_ERR_INGEST_FALLBACK = 19 /* cov_ingest_end: the file needs the CPU reader (see cov_last_error); nothing was appended */
} cov_status;

/* nm_kind values: how the record's NM aux tag was encoded (lib.rs:138-158). */
#define COV_NM_ABSENT 0
#define COV_NM_UNSIGNED 1 /* BAM type C, S or I */
#define COV_NM_BADTYPE 2  /* any other type: the reference panics */

/* cov_config.want bits */
#define COV_WANT_HIST 1u     /* per-contig depth histogram (trimmed_mean, coverage_histogram) */
#define COV_WANT_IDENTITY 2u /* ordered f64 identity sums (anir) */
/* with COV_WANT_IDENTITY: compute only one of the two sums (the other comes back 0).  contig_coverage and the
 * separator / single-genome scan use the primary-read sum (contig.rs:208, genome.rs:724), the contig-names genome scan
 * the not-supplementary one (genome.rs:220). */
#define COV_WANT_IDENTITY_PRIMARY_ONLY 4u
#define COV_WANT_IDENTITY_NONSUPP_ONLY 8u
/* A device ingest will follow (cov_ingest_begin): its streams, events and fixed tables — ~50 ms of runtime calls — are created by a helper thread
 * from cov_create on, beside the caller's next steps, instead of inside cov_ingest_begin.  Nothing else changes. */
#define COV_WANT_INGEST 16u

typedef struct {
    int32_t device; /* HIP device ordinal */
    /* FlagFilter, lib.rs:60-79 (CLI defaults: improper in, supplementary in, secondary out; coverm.rs:1661-1665) */
    uint8_t include_improper_pairs;
    uint8_t include_supplementary;
    uint8_t include_secondary;
    /* Single-read alignment thresholds = the stateless branch of ReferenceSortedBamFilter::read
     * (filter.rs:88-116, 243-279).  filter_single = 1 iff that branch is the active reader
     * (filtering_single && !filtering_pairs, filter.rs:48-61). */
    uint8_t filter_single;
    uint8_t min_mapq; /* 255 = no MAPQ test (filter.rs:13) */
    uint8_t reserved0[3];
    uint32_t min_aligned_length;
    float min_percent_identity; /* fraction in [0,1] */
    float min_aligned_percent;  /* fraction in [0,1] */
    /* --contig-end-exclusion: window [excl, L-excl) for the mean / trimmed_mean / variance /
     * histogram family (estimators.rs:386-398, 436-449). */
    uint64_t contig_end_exclusion;
    uint32_t want; /* COV_WANT_* */
    uint32_t reserved1;
} cov_config;

/*
 * One batch of BAM records in file order, struct-of-arrays.  Field provenance (rust-htslib accessors
 * used by the reference): tid/pos contig.rs:124,166; flag lib.rs:67-78; mapq filter.rs:251; NM
 * lib.rs:138-158; l_seq = record.seq().len() filter.rs:277; cigar contig.rs:168.
 * cigar words are raw BAM: len << 4 | op, op order MIDNSHP=X.  cigar_off has n_records + 1 entries
 * indexing `cigar` (cigar_off[0] may be non-zero; n_cigar = cigar_off[n] - cigar_off[0]).
 */
typedef struct {
    const int32_t *tid;
    const int32_t *pos;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint32_t *nm;
    const uint8_t *nm_kind;
    const uint32_t *l_seq;
    const uint32_t *cigar_off;
    const uint32_t *cigar;
    uint64_t n_records;
} cov_batch;

/*
 * Per-contig result.  "considered" = record survives the reader stage (filter_single), passes
 * FlagFilter and is mapped — i.e. reaches the CIGAR walk of the scan loops.  Depth statistics are
 * over per-base depth d[p] = prefix sum of the +1/-1 events of every M/=/X run (contig.rs:171-186):
 *   window  = [excl, L-excl) if 2*excl < L, else empty           (estimators.rs:386-398)
 *   full    = [0, L)                                              (estimators.rs:492-501)
 */
typedef struct {
    uint64_t n_primary;  /* considered && !secondary && !supplementary        contig.rs:157-159 */
    uint64_t n_pass;     /* considered                                         genome.rs:173-174 */
    uint64_t n_nonsupp;  /* considered && !supplementary                       genome.rs:677-682 */
    uint64_t sum_nm;     /* sum of NM over considered records                  contig.rs:207 */
    uint64_t sum_indel;  /* sum of I and D lengths                             contig.rs:189,197 */
    double sum_identity_primary; /* in file order; contig.rs:208-211, genome.rs:724-727 (COV_WANT_IDENTITY) */
    double sum_identity_nonsupp; /* in file order; genome.rs:220-223                      (COV_WANT_IDENTITY) */
    uint64_t win_sum_d;   /* sum of d over window (wrapping u64) */
    uint64_t win_sum_d2;  /* sum of d*d over window (wrapping u64) */
    uint64_t win_covered; /* #{p in window : d > 0} */
    uint64_t full_covered; /* #{p in [0,L) : d > 0} */
    uint64_t first_record; /* file-order index of the first / last considered record of this contig */
    uint64_t last_record;  /*   (sortedness is judged from these, see cov_finish) */
    uint32_t win_min_d;    /* min / max depth over the window (0 / 0 when the window is empty) */
    uint32_t win_max_d;
    uint32_t hist_len;     /* COV_WANT_HIST: win_max_d + 1 bins at hist[hist_off ..], else 0 */
    uint32_t reserved;
    uint64_t hist_off;
} cov_contig_stats;

typedef struct {
    uint64_t num_detected_primary_alignments; /* every record pushed with !secondary && !supplementary
                                                 (bam_generator.rs:114-118, filter.rs:94-96) */
    uint64_t n_records;
    uint64_t n_considered;
    uint64_t hist_total; /* number of u64 bins cov_fetch_hist() will write */
} cov_summary;

/* Named kernels of the device pipeline, for cov_kernel_ms(). */
typedef enum {
    COV_K_PREP = 0,     /* filter + CIGAR summary + per-contig record counters */
    COV_K_RANGES = 1,   /* tile -> candidate record range */
    COV_K_PILEUP = 2,   /* LDS-tiled events + scan + statistics (the dominant kernel) */
    COV_K_IDENTITY = 3, /* ordered f64 identity sums */
    COV_K_HIST = 4,     /* histogram arena layout + zero fill (before the pileup) */
    COV_K_HIST_COMPACT = 5, /* compact histogram (behind the pileup) */
    COV_K_ESTIMATE = 6, /* CoverageEstimator::calculate_coverage of every contig (cov_set_estimators) */
    COV_K_COUNT = 7
} cov_kernel_id;

/* --- lifecycle ------------------------------------------------------------------------------- */
int cov_abi_version(void);
/* Creates a session on cfg->device.  Fails with COV_ERR_HIP (never falls back) if no GPU. */
cov_status cov_create(const cov_config *cfg, cov_session **out);
void cov_destroy(cov_session *s);
const char *cov_last_error(const cov_session *s); /* s may be NULL for cov_create failures */

/* BAM header: n reference sequences and their lengths (HeaderView::target_len, contig.rs:145). */
cov_status cov_set_targets(cov_session *s, uint32_t n_targets, const uint64_t *target_len);

/* Optional: mask[t] == 0 removes target t from the CIGAR walk, the NM requirement and the depth
 * statistics while its records still take part in the sortedness check — exactly what
 * mosdepth_genome_coverage_with_contig_names does for contigs outside every genome
 * (genome.rs:170-171).  NULL (default) = every target participates. */
cov_status cov_set_target_mask(cov_session *s, const uint8_t *mask);

/* Appends a batch of records held in HOST memory (copied to the device on the session stream).  The arrays may
 * be changed or freed as soon as the call returns.  Arrays in page-locked memory (cov_host_alloc, hipHostMalloc)
 * move by DMA at link rate without the runtime's per-call pinning of pageable pages. */
cov_status cov_push_batch(cov_session *s, const cov_batch *host_batch);
/* Appends a batch whose arrays already live in DEVICE memory (no copy; the caller keeps them
 * alive and unmodified until cov_finish returns).  Used when ingest writes straight to HBM. */
cov_status cov_push_batch_device(cov_session *s, const cov_batch *device_batch);

/*
 * Runs the device pipeline over everything pushed and writes one cov_contig_stats per target into
 * `stats` (n_targets entries, caller-owned).  Returns COV_ERR_UNSORTED / COV_ERR_NM_* /
 * COV_ERR_POS_OOB exactly when the reference's scan would have panicked.  The session keeps its
 * records, so cov_finish may be called again (e.g. for benchmarking) and cov_copy_depth works
 * afterwards; cov_reset drops the records.
 */
cov_status cov_finish(cov_session *s, cov_contig_stats *stats, cov_summary *summary);
/* After cov_finish with COV_WANT_HIST: concatenated histograms, hist[stats[t].hist_off + d] =
 * #{p in window of contig t : depth == d}, summary.hist_total entries. */
cov_status cov_fetch_hist(cov_session *s, uint64_t *hist);
/* Bit-exact per-base depth of one contig (target_len[tid] entries), for parity checks. */
cov_status cov_copy_depth(cov_session *s, uint32_t tid, int32_t *depth_out);
cov_status cov_reset(cov_session *s);
/* Optional: size the record store for n_records / n_cigar up front (a streamed ingest that knows roughly what is coming
 * avoids regrowing the HBM arrays while it pushes). */
cov_status cov_reserve(cov_session *s, uint64_t n_records, uint64_t n_cigar);

/*
 * Device ingest (optional accelerator for hosts with few cores): the BGZF-compressed BAM goes to HBM as it is; the GPU inflates
 * every block (one lane per block, RFC 1951 decoder with LDS tables), checks its CRC-32, finds the record boundaries
 * (speculative per segment, verified to equal the serial hop) and writes the records straight into the session's record store —
 * the same store cov_push_batch appends to, so cov_finish does not care where the records came from.  The inflated stream
 * exists only one window at a time (the blocks one inflate launch holds, ~3 GB; a record cut by a window's end is carried
 * into the next), so device memory stays bounded whatever the file size.  The host only reads the
 * file into page-locked buffers, hops the 18-byte BGZF block headers and parses the BAM header (reference names / lengths,
 * offset of the first record in the inflated stream).
 *   cov_set_targets(s, ...)                         reference lengths: the record-boundary test uses them
 *   cov_ingest_begin(s, file_bytes, first_record_offset, check_crc)       first_record_offset: in the inflated stream, behind the BAM header
 *   loop: cov_ingest_slot_wait(s, slot) -> fill the slot's buffer -> cov_ingest_feed(s, slot, buf, file_offset, n, blocks, n_blocks)
 *         (COV_INGEST_SLOTS staging buffers rotate; slot_wait may be called from a reader thread while another thread feeds;
 *          `blocks` = the BGZF blocks COMPLETED by this piece: offsets are absolute in the file / the inflated stream; the
 *          copy is asynchronous, the inflate kernels of these blocks run behind it)
 *   cov_ingest_end(s, &n_records)
 * Anything irregular (inflate or CRC failure, boundaries that do not verify, a CG:B,I long-CIGAR placeholder) makes
 * cov_ingest_end return COV_ERR_INGEST_FALLBACK with nothing appended: decode that file on the host and cov_push_batch it.
 */
#define COV_INGEST_SLOTS 4
typedef struct {
    uint64_t in_off;  /* first byte of the block's raw DEFLATE data, offset in the file */
    uint64_t out_off; /* offset of its inflated bytes in the inflated stream (running sum of ISIZE) */
    uint32_t in_len, isize, crc, pad;
} cov_bgzf_block;
cov_status cov_ingest_begin(cov_session *s, uint64_t compressed_bytes, uint64_t first_record_offset, int check_crc);
/* Optional, between cov_ingest_begin and the first feed — one rank of a span-sharded file: only records with key_lo <= key < key_hi
 * are taken (key = tid; 0x7fffffff for records without a reference, which sort last); search_first_record: the bytes fed start
 * at a BGZF block somewhere inside the file (first_record_offset is then 0 and the first record boundary is searched);
 * open_end: the bytes fed end before the file does, so a record cut by their end is expected — it must lie beyond key_hi,
 * else COV_ERR_INGEST_FALLBACK; [file_lo, file_hi) = the file bytes that will be fed (sizes the record store once). */
cov_status cov_ingest_span(cov_session *s, int64_t key_lo, int64_t key_hi, int search_first_record, int open_end, uint64_t file_lo, uint64_t file_hi);
cov_status cov_ingest_slot_wait(cov_session *s, int slot);
cov_status cov_ingest_feed(cov_session *s, int slot, const void *host_bytes, uint64_t file_offset, uint64_t n_bytes,
                           const cov_bgzf_block *blocks, uint32_t n_blocks);
cov_status cov_ingest_end(cov_session *s, uint64_t *n_records);
/* Gives up an ingest that was begun but cannot be ended (a malformed block header met by the driver, a read error): waits for
 * everything queued on the device, appends nothing, leaves the session ready for cov_push_batch.  cov_reset and cov_push_batch
 * call it themselves when an ingest is still open. */
cov_status cov_ingest_abort(cov_session *s);
cov_status cov_ingest_release(cov_session *s); /* frees the compressed / inflated buffers (kept between files otherwise) */
cov_status cov_ingest_copy_inflated(cov_session *s, uint64_t offset, uint64_t n, void *out); /* test hook; single-window files only */
/* ---- reader-stage PAIR filter on the device: ReferenceSortedBamFilter::read, pair branch (src/filter.rs:117-228, filter_out = true) +
 * read_pair_passes_filter (:281-336) over the records the device ingest put into the session's store.
 *   cov_ingest_want_mates(s, 1)      before cov_ingest_begin: the extraction also keeps next_refID and a 96-bit hash of each read name
 *   ... ingest one or more files ...
 *   cov_pair_filter_apply(s, &f, &n_selected, &n_primary)
 * replaces the store by the records the reference's filter would return, in its order (pairs by their second record; first record,
 * then second); *n_primary = num_detected_primary_alignments (filter.rs:129-131: every primary record of the input).  cov_finish then
 * runs with cov_config.filter_single = 0.  COV_ERR_NM_MISSING / COV_ERR_NM_BADTYPE where the reference's nm() would have panicked;
 * COV_ERR_INGEST_FALLBACK (store untouched) when more than 2^20 records carry a read name that occurs more than twice among the
 * primary proper-pair records of one reference (fewer are replayed exactly): run the host filter (covh_pair_mode_order) on the CPU
 * reader's records then.  Same layout as covh_pair_filter. */
typedef struct {
    int32_t filter_single;     /* single-read thresholds apply to both mates as well (filter.rs:48-55) */
    uint8_t min_mapq;          /* 255 = off */
    uint8_t pad[3];
    uint32_t min_aligned_length_single;
    float min_percent_identity_single, min_aligned_percent_single;
    uint32_t min_aligned_length_pair;
    float min_percent_identity_pair, min_aligned_percent_pair;
} cov_pair_filter;
cov_status cov_ingest_want_mates(cov_session *s, int on);
cov_status cov_pair_filter_apply(cov_session *s, const cov_pair_filter *f, uint64_t *n_selected, uint64_t *n_primary);
/* Test hook: the session's own record store copied back into caller-sized host arrays (host == NULL: only the counts). */
cov_status cov_copy_records(cov_session *s, const cov_batch *host, uint64_t *n_records, uint64_t *n_cigar);

/* Multi-GPU, one process: after cov_finish on every session (one per device, same targets), ONE RCCL gather moves each
 * rank's per-contig result block (fixed size: 160 B per contig + counters) to the device of sessions[root] over xGMI and
 * from there to the host in one DMA; cov_gathered then decodes rank r's block exactly as cov_finish decodes its own
 * (same error reporting).  librccl is bound at run time; COV_ERR_HIP if it is missing.  A device listed more than once
 * (functional checks on a single-GPU box) is served by plain device copies, since RCCL refuses two ranks on one device. */
cov_status cov_gather(cov_session *const *sessions, uint32_t n, uint32_t root);
cov_status cov_gathered(cov_session *root_session, uint32_t rank, cov_contig_stats *stats, cov_summary *summary);

/* Average duration (ms) of one launch of kernel `k` during the last cov_finish, measured with HIP
 * events recorded on the session's stream around that kernel; *launches receives the count. */
cov_status cov_kernel_ms(const cov_session *s, cov_kernel_id k, double *ms_total, uint32_t *launches);
/* Which paths the last cov_finish took — diagnostics, so that a parity case built for one branch of the device pipeline can assert that the
 * branch ran (tests/test_gpu_step_shapes.py).  listed_steps: steps of 64 records k_prep_lean left to k_prep_generic (generic_only = 1: a
 * sample of fewer than 128 records per contig, k_prep_generic walked all of them and nothing was listed); slow_tiles: tiles k_pileup_fast
 * left to k_pileup_stream; bucket_records: records whose CIGAR went through the bucket list (CX_MIN_OPS operations and more). */
typedef struct { uint32_t listed_steps, generic_only, slow_tiles, bucket_records; } cov_path_counts;
cov_status cov_last_paths(const cov_session *s, cov_path_counts *out);
/* Algorithmic HBM bytes of the last cov_finish (DESIGN.md "Algorithmic bytes"): record SoA + CIGAR
 * words read once, plus result structs written. */
cov_status cov_algorithmic_bytes(const cov_session *s, uint64_t *bytes);

/* ---- CoverageEstimator::calculate_coverage on the device (estimators.rs:530-839), for `coverm contig`: one entry per contig, unobserved
 * lengths [0] (contig.rs:62-66).  cov_set_estimators before cov_finish; cov_finish then also evaluates every estimator for every contig
 * (one wave per contig, the reference's float expressions in the reference's order, one rounding per operation) and
 * cov_fetch_estimates copies the n_targets x n_est floats out (row = contig, 0 for a contig without a considered record; RPKM before the
 * printer's normalisation, as calculate_coverage returns it).  The floats equal the host path's (coverm_host.h covh_contig_coverage over
 * cov_contig_stats) bit for bit; a host that takes them skips the per-contig finalisation and, unless it prints histograms, the
 * histogram fetch.  Kinds = enum CoverageEstimator's variant order (coverm_host.h covh_kind).  Not offered: COV_EST_TPM (f64 exp / ln of the
 * host's libm) and COV_EST_PILEUP_COUNTS (prints the histogram itself) — COV_ERR_INVALID_ARG, evaluate those on the host.  Needs
 * COV_WANT_HIST for a trimmed mean and COV_WANT_IDENTITY for ANIr, no target mask (genome modes aggregate contigs first: host). */
typedef struct {
    int32_t kind;                      /* COV_EST_* */
    float min_fraction_covered_bases;
    uint64_t contig_end_exclusion;     /* as the session's cov_config.contig_end_exclusion */
    int32_t exclude_mismatches;
    float trim_min, trim_max;
} cov_estimator;                       /* same layout as coverm_host.h's covh_estimator */
#define COV_EST_MEAN 0
#define COV_EST_TRIMMED_MEAN 1
#define COV_EST_PILEUP_COUNTS 2
#define COV_EST_COVERED_FRACTION 3
#define COV_EST_COVERED_BASES 4
#define COV_EST_RPKM 5
#define COV_EST_TPM 6
#define COV_EST_VARIANCE 7
#define COV_EST_LENGTH 8
#define COV_EST_READ_COUNT 9
#define COV_EST_READS_PER_BASE 10
#define COV_EST_ANIR 11
#define COV_EST_MAX 16
cov_status cov_set_estimators(cov_session *s, const cov_estimator *est, uint32_t n_est); /* n_est = 0: off (the default) */
cov_status cov_fetch_estimates(cov_session *s, float *out);                              /* after cov_finish: n_targets * n_est floats */

/* ---- bounded record store.  The reference holds one contig at a time and flushes it when the tid changes (contig.rs:128-155), so a
 * sample may be arbitrarily large.  The session's record store keeps as many contigs as fit under a cap (2^31 records / 2^31 CIGAR
 * words by default; COVERM_KNOBS="store_cap_records=N,store_cap_cigar=M", read by cov_create).  When cov_push_batch, cov_push_batch_device
 * or the device ingest would take it past the cap, the pipeline runs over what the store holds, the contigs that are complete stay on
 * the host, the records from the first considered record of the contig in flight onwards move to the front of the store and the call
 * goes on ("a spill"); cov_finish / cov_fetch_hist / cov_gather then return the whole sample (record indices count from the sample's first
 * record).  What remains: ONE reference's records must fit (2^32 - 16 records and CIGAR words); after a spill cov_copy_depth,
 * cov_interval_stats_compute, cov_copy_records and cov_pair_filter_apply return COV_ERR_STATE (they need every record) and a device ingest
 * that has to hand its file to the CPU reader fails with COV_ERR_STATE instead of COV_ERR_INGEST_FALLBACK; a store that carries mate
 * columns (cov_ingest_want_mates) or an adopted device batch is never spilled.  cov_store_spills: spills since the last cov_reset. */
uint32_t cov_store_spills(const cov_session *s);


/* ---- per-interval statistics (per-gene coverage, the reference's src/genes.rs:508-535).  After cov_finish: the depth of
 * every target is materialised once in HBM (4 B per base, kept until the next push / reset / finish) and each interval
 * [start, end) of target `tid` is reduced by one wave.  contig_end_exclusion applies to the INTERVAL's own ends: the
 * window is [start+excl, end-excl) when 2*excl < end-start, else empty; full_covered has no exclusion. */
typedef struct { uint32_t tid, pad; uint64_t start, end; } cov_interval;
typedef struct {
    uint64_t win_sum_d, win_sum_d2, win_covered, full_covered;
    uint32_t win_min_d, win_max_d; /* over the window; 0 when it is empty */
    uint32_t hist_len, pad;        /* window non-empty: win_max_d + 1 bins at hist[hist_off ..] (when want_hist), else 0 */
    uint64_t hist_off;
} cov_interval_stats;
cov_status cov_interval_stats_compute(cov_session *s, const cov_interval *intervals, uint64_t n, uint64_t contig_end_exclusion,
                                      int want_hist, cov_interval_stats *out, uint64_t *hist_total);
cov_status cov_fetch_interval_hist(cov_session *s, uint64_t *hist);

/* Page-locked host memory for record batches, pooled: a freed block is kept for the next request of similar size
 * (a decoder filling one batch per BAM file gets its arrays back without re-pinning).  cov_host_alloc returns NULL
 * when no HIP device is usable; cov_host_free returns 0 if `p` did not come from cov_host_alloc; cov_host_trim
 * gives the parked blocks back to the system. */
/* Binds the CALLING THREAD (and the threads it creates afterwards) to the CPUs of the NUMA node device `device` is attached to;
 * returns that node, or -1 when nothing was changed.  Call it in the thread that is going to read and feed a file for that device. */
int cov_bind_thread_to_device_node(int device);
void *cov_host_alloc(size_t bytes);
/* Host memory the caller owns (e.g. an mmap of the BAM file, page aligned) made readable by the session's device, so that
 * cov_ingest_feed can take its bytes from there without a staging copy; unregister once the session has synchronised
 * (after cov_ingest_end / cov_ingest_abort).  COV_ERR_HIP when the runtime refuses the range: use staging buffers then. */
cov_status cov_host_register(cov_session *s, void *p, size_t bytes);
cov_status cov_host_unregister(cov_session *s, void *p);
int cov_host_free(void *p);
void cov_host_trim(void);

#ifdef __cplusplus
}
#endif
#endif /* COVERMHIP_H */
