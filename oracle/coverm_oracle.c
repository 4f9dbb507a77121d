/*
 * TEST INFRASTRUCTURE ONLY — CPU oracle for the CoverM BAM -> pileup -> per-contig/per-genome path.
 *
 * A deliberately literal, scalar, single-threaded restatement of the reference's algorithm
 * (wwood/CoverM v0.8.0, /root/reference).  It keeps the reference's structure — one Vec<i32>
 * of deltas per contig, one sequential prefix-sum pass PER ESTIMATOR, Vec<u64> histograms
 * grown on demand — so that it is both the parity checker and the timed "port" CPU baseline.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (coverm_amd/, libcovermhip.so) never links or calls it.
 *
 * Parity pinned: tests/test_oracle_golden.py checks this file against the reference's own
 * golden vectors (contig.rs:325-577, genome.rs:1088-1986, filter.rs:342-844,
 * tests/test_cmdline.rs --bam-files cases) on the reference's fixture BAMs.
 *
 * Each function cites the reference file:line it follows.  Integer semantics follow a Rust
 * *release* build (wrapping u64/usize arithmetic, saturating float->int casts).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef int64_t i64;

/* ---- estimator kinds: order of `enum CoverageEstimator`, estimators.rs:4-81 ---- */
enum {
    ORC_MEAN = 0,
    ORC_TRIMMED_MEAN = 1,
    ORC_PILEUP_COUNTS = 2,
    ORC_COVERED_FRACTION = 3,
    ORC_COVERED_BASES = 4,
    ORC_RPKM = 5,
    ORC_TPM = 6,
    ORC_VARIANCE = 7,
    ORC_LENGTH = 8,
    ORC_READ_COUNT = 9,
    ORC_READS_PER_BASE = 10,
    ORC_ANIR = 11
};

/* Estimator parameters as given to the constructors, estimators.rs:107-224. */
typedef struct {
    int32_t kind;
    float min_fraction_covered_bases;
    u64 contig_end_exclusion;
    int32_t exclude_mismatches;
    float trim_min, trim_max;
} orc_est_param;

typedef struct {
    orc_est_param p;
    u64 total_count, total_bases, num_covered_bases, num_mapped_reads, total_mismatches;
    u64 observed_contig_length;
    double sum_identity;
    u64 num_reads;
    u64 *counts;
    size_t counts_len, counts_cap;
} orc_est;

/* Emission protocol = the CoverageTaker trait calls, coverage_takers.rs:29-38. */
enum { ORC_EM_START_ENTRY = 0, ORC_EM_SINGLE = 1, ORC_EM_PILEUP = 2, ORC_EM_FINISH = 3 };
typedef struct {
    int32_t type;
    int32_t pad;
    i64 a;     /* START_ENTRY: entry_order_id ; PILEUP: num_reads (coverage depth i) */
    u64 b;     /* PILEUP: num_bases */
    float cov; /* SINGLE: coverage */
    int32_t name_tid; /* START_ENTRY: tid whose name (or genome prefix / genome index) names the entry */
} orc_emit;

typedef struct {
    orc_emit *e;
    size_t n, cap;
} orc_out;

enum {
    ORC_OK = 0,
    ORC_ERR_UNSORTED = 1,     /* contig.rs:129-132, genome.rs:133-136, 549-552 */
    ORC_ERR_NM_MISSING = 2,   /* lib.rs:149-156 */
    ORC_ERR_NM_BADTYPE = 3,   /* lib.rs:144-147 */
    ORC_ERR_POS_OOB = 4,      /* index panic at contig.rs:178 */
    ORC_ERR_NO_SEPARATOR = 5, /* genome.rs:802 */
    ORC_ERR_BAD_CIGAR = 6
};

static void out_push(orc_out *o, orc_emit em) {
    if (o->n == o->cap) {
        o->cap = o->cap ? o->cap * 2 : 1024;
        o->e = (orc_emit *)realloc(o->e, o->cap * sizeof(orc_emit));
    }
    o->e[o->n++] = em;
}
static void em_start(orc_out *o, i64 id, int32_t name_tid) {
    orc_emit e; memset(&e, 0, sizeof e); e.type = ORC_EM_START_ENTRY; e.a = id; e.name_tid = name_tid; out_push(o, e);
}
static void em_single(orc_out *o, float c) {
    orc_emit e; memset(&e, 0, sizeof e); e.type = ORC_EM_SINGLE; e.cov = c; out_push(o, e);
}
static void em_pileup(orc_out *o, i64 i, u64 b) {
    orc_emit e; memset(&e, 0, sizeof e); e.type = ORC_EM_PILEUP; e.a = i; e.b = b; out_push(o, e);
}
static void em_finish(orc_out *o) {
    orc_emit e; memset(&e, 0, sizeof e); e.type = ORC_EM_FINISH; out_push(o, e);
}

/* Rust `x as usize` for f32: saturating, NaN -> 0. */
static u64 f32_to_usize(float x) {
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 18446744073709551616.0f) return UINT64_MAX;
    return (u64)x;
}

/* estimators.rs:268-364 */
static void est_setup(orc_est *e) {
    e->total_count = e->total_bases = e->num_covered_bases = e->num_mapped_reads = 0;
    e->total_mismatches = e->observed_contig_length = 0;
    e->sum_identity = 0.0;
    e->num_reads = 0;
    e->counts_len = 0; /* *counts = vec![] */
}

static void counts_resize(orc_est *e, size_t n) {
    if (n > e->counts_cap) {
        size_t c = e->counts_cap ? e->counts_cap : 64;
        while (c < n) c *= 2;
        e->counts = (u64 *)realloc(e->counts, c * sizeof(u64));
        e->counts_cap = c;
    }
    for (size_t i = e->counts_len; i < n; i++) e->counts[i] = 0;
    e->counts_len = n;
}

/* estimators.rs:366-528 */
static void est_add_contig(orc_est *e, const int32_t *ud, size_t len, u64 n_reads, u64 mismatches,
                           double sum_identity) {
    switch (e->p.kind) {
    case ORC_MEAN: { /* :374-409 */
        e->num_mapped_reads += n_reads;
        e->total_mismatches += mismatches;
        u64 excl = e->p.contig_end_exclusion;
        if (excl * 2 < (u64)len) e->total_bases += (u64)len - 2 * excl;
        else return;
        int32_t cumulative_sum = 0;
        size_t start_from = (size_t)excl, end_at = len - (size_t)excl - 1;
        for (size_t i = 0; i < len; i++) {
            cumulative_sum += ud[i];
            if (i >= start_from && i <= end_at) {
                if (cumulative_sum > 0) e->num_covered_bases += 1;
                e->total_count += (u64)(i64)cumulative_sum;
            }
        }
        break;
    }
    case ORC_TRIMMED_MEAN:
    case ORC_PILEUP_COUNTS:
    case ORC_VARIANCE: { /* :410-466 */
        e->num_mapped_reads = n_reads; /* sic: assignment, :434 */
        u64 excl = e->p.contig_end_exclusion;
        if (excl * 2 < (u64)len) e->observed_contig_length += (u64)len - 2 * excl;
        else return;
        int32_t cumulative_sum = 0;
        size_t start_from = (size_t)excl, end_at = len - (size_t)excl - 1;
        for (size_t i = 0; i < len; i++) {
            cumulative_sum += ud[i];
            if (i >= start_from && i <= end_at) {
                if (cumulative_sum > 0) e->num_covered_bases += 1;
                if (e->counts_len <= (size_t)cumulative_sum) counts_resize(e, (size_t)cumulative_sum + 1);
                e->counts[(size_t)cumulative_sum] += 1;
            }
        }
        break;
    }
    case ORC_COVERED_FRACTION:
    case ORC_COVERED_BASES:
    case ORC_RPKM:
    case ORC_TPM: { /* :467-502 — no contig-end exclusion */
        e->num_mapped_reads += n_reads;
        e->total_bases += (u64)len;
        int32_t cumulative_sum = 0;
        for (size_t i = 0; i < len; i++) {
            cumulative_sum += ud[i];
            if (cumulative_sum > 0) e->num_covered_bases += 1;
        }
        break;
    }
    case ORC_LENGTH:
    case ORC_READS_PER_BASE: /* :503-513 */
        e->observed_contig_length += (u64)len;
        e->num_mapped_reads += n_reads;
        break;
    case ORC_READ_COUNT: /* :514-518 */
        e->num_mapped_reads += n_reads;
        break;
    case ORC_ANIR: /* :519-525 */
        e->num_reads += n_reads;
        e->sum_identity += sum_identity;
        break;
    }
}

/* estimators.rs:226-242 */
static u64 unobserved_bases(const u64 *unobs, size_t n, u64 excl) {
    u64 s = 0, e = 2 * excl;
    for (size_t i = 0; i < n; i++) s += (unobs[i] < e) ? unobs[i] : unobs[i] - e;
    return s;
}
static u64 sum_u64(const u64 *v, size_t n) {
    u64 s = 0;
    for (size_t i = 0; i < n; i++) s += v[i];
    return s;
}

/* estimators.rs:530-839 */
static float est_calculate(orc_est *e, const u64 *unobs, size_t n_unobs) {
    const float minfrac = e->p.min_fraction_covered_bases;
    switch (e->p.kind) {
    case ORC_MEAN: { /* :532-565 */
        u64 T = e->total_bases + unobserved_bases(unobs, n_unobs, e->p.contig_end_exclusion);
        if (T == 0 || ((float)e->num_covered_bases / (float)T) < minfrac) return 0.0f;
        float num = e->p.exclude_mismatches ? (float)(e->total_count - e->total_mismatches)
                                            : (float)e->total_count;
        return num / (float)T;
    }
    case ORC_TRIMMED_MEAN: { /* :566-647 */
        u64 U = unobserved_bases(unobs, n_unobs, e->p.contig_end_exclusion);
        u64 T = e->observed_contig_length + U;
        if (T == 0) return 0.0f;
        if (((float)e->num_covered_bases / (float)T) < minfrac) return 0.0f;
        u64 min_index = f32_to_usize(floorf(e->p.trim_min * (float)T));
        u64 max_index = f32_to_usize(ceilf(e->p.trim_max * (float)T));
        if (e->num_covered_bases == 0) return 0.0f;
        e->counts[0] += U;
        u64 num_accounted_for = 0, total = 0;
        int started = 0;
        for (size_t i = 0; i < e->counts_len; i++) {
            u64 num_covered = e->counts[i];
            num_accounted_for += num_covered;
            if (num_accounted_for >= min_index) {
                if (started) {
                    if (num_accounted_for > max_index) {
                        u64 num_excess = num_accounted_for - num_covered;
                        u64 num_wanted = (max_index >= num_excess) ? max_index - num_excess + 1 : 0;
                        total += num_wanted * (u64)i;
                        break;
                    } else {
                        total += num_covered * (u64)i;
                    }
                } else if (num_accounted_for > max_index) {
                    total = (max_index - min_index + 1) * (u64)i; /* no break, :626-629 */
                    started = 1;
                } else if (num_accounted_for < min_index) {
                } else {
                    u64 num_wanted = num_accounted_for - min_index + 1;
                    total = num_wanted * (u64)i;
                    started = 1;
                }
            }
        }
        return (float)total / (float)(max_index - min_index);
    }
    case ORC_PILEUP_COUNTS: { /* :648-678 */
        if (e->observed_contig_length == 0) return 0.0f;
        u64 T = e->observed_contig_length + unobserved_bases(unobs, n_unobs, e->p.contig_end_exclusion);
        if (((float)e->num_covered_bases / (float)T) < minfrac) return 0.0f;
        return (float)(T - e->num_covered_bases + 1);
    }
    case ORC_COVERED_FRACTION: { /* :679-695 */
        u64 T = e->total_bases + sum_u64(unobs, n_unobs);
        if (T == 0 || ((float)e->num_covered_bases / (float)T) < minfrac) return 0.0f;
        return (float)e->num_covered_bases / (float)T;
    }
    case ORC_COVERED_BASES: { /* :696-712 */
        u64 T = e->total_bases + sum_u64(unobs, n_unobs);
        if (T == 0 || ((float)e->num_covered_bases / (float)T) < minfrac) return 0.0f;
        return (float)e->num_covered_bases;
    }
    case ORC_RPKM: { /* :713-737 */
        u64 T = e->total_bases + sum_u64(unobs, n_unobs);
        if (T == 0 || ((float)e->num_covered_bases / (float)T) < minfrac) return 0.0f;
        return (float)(e->num_mapped_reads * 1000000000ULL) / (float)T;
    }
    case ORC_TPM: { /* :738-763 */
        u64 T = e->total_bases + sum_u64(unobs, n_unobs);
        if (T == 0 || ((float)e->num_covered_bases / (float)T) < minfrac) return 0.0f;
        return (float)exp(log((double)e->num_mapped_reads) - log((double)T));
    }
    case ORC_VARIANCE: { /* :764-813 */
        u64 U = unobserved_bases(unobs, n_unobs, e->p.contig_end_exclusion);
        u64 T = e->observed_contig_length + U;
        if (T == 0) return 0.0f;
        if (((float)e->num_covered_bases / (float)T) < minfrac || T < 3 || e->counts_len == 0) return 0.0f;
        e->counts[0] += U;
        size_t k = 0;
        while (e->counts[k] == 0) k++;
        u64 ex = 0, ex2 = 0;
        for (size_t x = 0; x < e->counts_len; x++) {
            if (e->counts[x] == 0) continue;
            u64 nc = e->counts[x];
            ex += (u64)(x - k) * nc;
            ex2 += (u64)(x - k) * (u64)(x - k) * nc;
        }
        return ((float)ex2 - (float)(ex * ex) / (float)T) / (float)(T - 1);
    }
    case ORC_LENGTH: /* :814-817 */
        return (float)(e->observed_contig_length + sum_u64(unobs, n_unobs));
    case ORC_READ_COUNT: /* :818 */
        return (float)e->num_mapped_reads;
    case ORC_READS_PER_BASE: /* :819-826 */
        return (float)e->num_mapped_reads / (float)(e->observed_contig_length + sum_u64(unobs, n_unobs));
    case ORC_ANIR: /* :827-836 */
        if (e->num_reads == 0) return 0.0f;
        return (float)(e->sum_identity / (double)e->num_reads);
    }
    return 0.0f;
}

/* estimators.rs:936-969 */
static void est_print_coverage(const orc_est *e, float coverage, orc_out *o) {
    if (e->p.kind != ORC_PILEUP_COUNTS) { em_single(o, coverage); return; }
    for (size_t i = 0; i < e->counts_len; i++) {
        u64 cov;
        if (i == 0) {
            u64 c = f32_to_usize(floorf(coverage));
            cov = (c == 0) ? 0 : c - 1;
        } else cov = e->counts[i];
        em_pileup(o, (i64)i, cov);
    }
}
/* estimators.rs:971-991 */
static void est_print_zero(const orc_est *e, orc_out *o, u64 entry_length) {
    if (e->p.kind == ORC_PILEUP_COUNTS) return;
    if (e->p.kind == ORC_LENGTH) em_single(o, (float)entry_length);
    else em_single(o, 0.0f);
}

/* ------------------------------------------------------------------ record stream */
typedef struct {
    const int32_t *tid, *pos;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint32_t *nm;
    const uint8_t *nm_kind; /* 0 absent, 1 unsigned (C/S/I), 2 other type */
    const uint32_t *l_seq;
    const uint32_t *cigar_off; /* R+1 */
    const uint32_t *cigar;     /* len<<4 | op, op order MIDNSHP=X */
    u64 n_records;
    /* reader order after the (optional) filter stage: indices into the arrays above */
    const u64 *order;
    u64 n_order;
} orc_records;

typedef struct {
    int32_t include_improper_pairs, include_supplementary, include_secondary;
} orc_flag_filter;

/* lib.rs:67-78 */
static int flag_passes(const orc_flag_filter *f, uint16_t flag) {
    if (!f->include_secondary && (flag & 0x100)) return 0;
    if (!f->include_supplementary && (flag & 0x800)) return 0;
    if (!f->include_improper_pairs && !(flag & 0x2)) return 0;
    return 1;
}

/* lib.rs:138-158.  Returns ORC_OK or the panic it would raise. */
static int nm_of(const orc_records *r, u64 i, u64 *out) {
    if (r->nm_kind[i] == 1) { *out = r->nm[i]; return ORC_OK; }
    return r->nm_kind[i] == 0 ? ORC_ERR_NM_MISSING : ORC_ERR_NM_BADTYPE;
}

/* filter.rs:243-279.  *pass receives the verdict; return value is the panic code if any. */
static int single_read_passes_filter(const orc_records *r, u64 i, uint32_t min_aligned_length,
                                     float min_percent_identity, float min_aligned_percent,
                                     uint8_t min_mapq, int *pass) {
    if (min_mapq != 255 && (r->mapq[i] < min_mapq || r->mapq[i] == 255)) { *pass = 0; return ORC_OK; }
    u64 edit;
    int rc = nm_of(r, i, &edit);
    if (rc) return rc;
    uint32_t aligned = 0;
    for (uint32_t c = r->cigar_off[i]; c < r->cigar_off[i + 1]; c++) {
        uint32_t op = r->cigar[c] & 15, len = r->cigar[c] >> 4;
        if (op == 0 || op == 1 || op == 2 || op == 8 || op == 7) aligned += len;
    }
    *pass = aligned >= min_aligned_length &&
            (float)aligned / (float)r->l_seq[i] >= min_aligned_percent &&
            1.0f - (float)edit / (float)aligned >= min_percent_identity;
    return ORC_OK;
}

/*
 * filter.rs:88-116 — the single-read branch of ReferenceSortedBamFilter::read with
 * filter_out = true.  Writes surviving record indices to order_out (capacity n_records),
 * counts primaries like :94-96.  (Pair mode, filter.rs:117-228, is restated in
 * oracle/oracle.py because it needs read names.)
 */
int orc_filter_single(const orc_records *r, const orc_flag_filter *ff, uint32_t min_aligned_length,
                      float min_percent_identity, float min_aligned_percent, uint8_t min_mapq,
                      u64 *order_out, u64 *n_out, u64 *num_detected_primary) {
    u64 n = 0, prim = 0;
    for (u64 i = 0; i < r->n_records; i++) {
        uint16_t flag = r->flag[i];
        if (!(flag & 0x800) && !(flag & 0x100)) prim++;
        int unmapped = (flag & 0x4) != 0;
        int passes1 = !unmapped && (ff->include_supplementary || !(flag & 0x800)) &&
                      (ff->include_secondary || !(flag & 0x100));
        if (passes1) {
            int pass = 0;
            int rc = single_read_passes_filter(r, i, min_aligned_length, min_percent_identity,
                                               min_aligned_percent, min_mapq, &pass);
            if (rc) return rc;
            if (pass) order_out[n++] = i;
        }
    }
    *n_out = n;
    *num_detected_primary = prim;
    return ORC_OK;
}

/* bam_generator.rs:113-119 — plain reader: identity order, primaries counted per record. */
u64 orc_count_primary(const orc_records *r) {
    u64 prim = 0;
    for (u64 i = 0; i < r->n_records; i++)
        if (!(r->flag[i] & 0x800) && !(r->flag[i] & 0x100)) prim++;
    return prim;
}

/* The CIGAR walk shared verbatim by contig.rs:166-202, genome.rs:179-214, genome.rs:683-718. */
static int cigar_walk(const orc_records *r, u64 i, int32_t *ud, size_t L, u64 *indels, u64 *aligned_len) {
    size_t cursor = (size_t)(i64)r->pos[i];
    for (uint32_t c = r->cigar_off[i]; c < r->cigar_off[i + 1]; c++) {
        uint32_t op = r->cigar[c] & 15;
        size_t len = r->cigar[c] >> 4;
        switch (op) {
        case 0: case 8: case 7: { /* M X = */
            if (cursor >= L) return ORC_ERR_POS_OOB;
            ud[cursor] += 1;
            size_t final_pos = cursor + len;
            if (final_pos < L) ud[final_pos] -= 1;
            cursor += len;
            *aligned_len += len;
            break;
        }
        case 2: cursor += len; *indels += len; *aligned_len += len; break; /* D */
        case 3: cursor += len; break;                                       /* N */
        case 1: *indels += len; *aligned_len += len; break;                 /* I */
        case 4: case 5: case 6: break;                                      /* S H P */
        default: return ORC_ERR_BAD_CIGAR;
        }
    }
    return ORC_OK;
}

typedef struct {
    u64 num_mapped_reads, num_reads;
} orc_reads_mapped;

static orc_est *est_new(const orc_est_param *p, size_t n) {
    orc_est *e = (orc_est *)calloc(n ? n : 1, sizeof(orc_est));
    for (size_t i = 0; i < n; i++) { e[i].p = p[i]; est_setup(&e[i]); }
    return e;
}
static void est_free(orc_est *e, size_t n) {
    for (size_t i = 0; i < n; i++) free(e[i].counts);
    free(e);
}

/* ------------------------------------------------------------------ contig.rs:13-253 */
typedef struct {
    orc_est *est; size_t n_est;
    orc_out *out;
    int print_zero;
    const i64 *target_len; int32_t n_targets;
} contig_ctx;

/* contig.rs:255-277 */
static void print_previous_zero_coverage_contigs(contig_ctx *cx, int32_t last_tid, int32_t current_tid) {
    for (int32_t my_tid = last_tid + 1; my_tid < current_tid; my_tid++) {
        em_start(cx->out, my_tid, my_tid);
        for (size_t k = 0; k < cx->n_est; k++) est_print_zero(&cx->est[k], cx->out, (u64)cx->target_len[my_tid]);
        em_finish(cx->out);
    }
}

/* contig.rs:40-104 */
static void process_previous_contigs(contig_ctx *cx, int32_t last_tid, int32_t tid, const int32_t *ud, size_t L,
                                     u64 n_reads, u64 edit, u64 indels, double *sum_identity, u64 *mapped_total) {
    if (last_tid != -2) {
        for (size_t k = 0; k < cx->n_est; k++)
            est_add_contig(&cx->est[k], ud, L, n_reads, edit - indels, *sum_identity);
        float *cov = (float *)malloc(sizeof(float) * (cx->n_est ? cx->n_est : 1));
        const u64 zero = 0;
        int nonzero = 0;
        for (size_t k = 0; k < cx->n_est; k++) {
            cov[k] = est_calculate(&cx->est[k], &zero, 1);
            if (cov[k] > 0.0f) nonzero = 1;
        }
        if (nonzero) *mapped_total += n_reads;
        if (cx->print_zero || nonzero) {
            em_start(cx->out, last_tid, last_tid);
            for (size_t k = 0; k < cx->n_est; k++) est_print_coverage(&cx->est[k], cov[k], cx->out);
            em_finish(cx->out);
        }
        for (size_t k = 0; k < cx->n_est; k++) est_setup(&cx->est[k]);
        *sum_identity = 0.0;
        free(cov);
    }
    if (cx->print_zero) print_previous_zero_coverage_contigs(cx, last_tid == -2 ? -1 : last_tid, tid);
}

int orc_contig_coverage(const orc_records *r, const i64 *target_len, int32_t n_targets,
                        const orc_est_param *params, int32_t n_est, int32_t print_zero,
                        const orc_flag_filter *ff, u64 num_detected_primary, orc_out *out,
                        orc_reads_mapped *rm) {
    contig_ctx cx;
    cx.est = est_new(params, (size_t)n_est); cx.n_est = (size_t)n_est;
    cx.out = out; cx.print_zero = print_zero; cx.target_len = target_len; cx.n_targets = n_targets;
    int32_t last_tid = -2;
    int32_t *ud = NULL; size_t L = 0;
    u64 mapped_total = 0, n_in_contig = 0, indels = 0, edit_total = 0;
    double sum_identity = 0.0;
    int rc = ORC_OK;
    for (u64 oi = 0; oi < r->n_order; oi++) {
        u64 i = r->order ? r->order[oi] : oi;
        uint16_t flag = r->flag[i];
        if (!flag_passes(ff, flag)) continue;               /* :119 */
        int32_t tid = r->tid[i];
        if (flag & 0x4) continue;                            /* :125 */
        if (tid != last_tid) {
            if (tid < last_tid) { rc = ORC_ERR_UNSORTED; goto done; }
            process_previous_contigs(&cx, last_tid, tid, ud, L, n_in_contig, edit_total, indels, &sum_identity,
                                     &mapped_total);
            free(ud);
            L = (size_t)target_len[tid];
            ud = (int32_t *)calloc(L ? L : 1, sizeof(int32_t)); /* vec![0; target_len], :144-145 */
            last_tid = tid;
            n_in_contig = 0; edit_total = 0; indels = 0; sum_identity = 0.0;
        }
        int primary = !(flag & 0x800) && !(flag & 0x100);
        if (primary) n_in_contig++;                          /* :157-159 */
        u64 aligned_len = 0;
        rc = cigar_walk(r, i, ud, L, &indels, &aligned_len);
        if (rc) goto done;
        u64 edit;
        rc = nm_of(r, i, &edit);                             /* :206 */
        if (rc) goto done;
        edit_total += edit;
        if (primary && aligned_len > 0)
            sum_identity += ((double)aligned_len - (double)edit) / (double)aligned_len; /* :208-211 */
    }
    process_previous_contigs(&cx, last_tid, n_targets, ud, L, n_in_contig, edit_total, indels, &sum_identity,
                             &mapped_total);
    rm->num_mapped_reads = mapped_total;
    rm->num_reads = num_detected_primary;
done:
    free(ud);
    est_free(cx.est, cx.n_est);
    return rc;
}

/* ------------------------------------------------------------------ genome.rs:17-322 */
int orc_genome_coverage_with_contig_names(const orc_records *r, const i64 *target_len, int32_t n_targets,
                                          const int32_t *genome_of_tid /* -1 = None */, int32_t n_genomes,
                                          const orc_est_param *params, int32_t n_est, int32_t print_zero,
                                          const orc_flag_filter *ff, u64 num_detected_primary, orc_out *out,
                                          orc_reads_mapped *rm) {
    size_t ne = (size_t)n_est;
    orc_est **per_genome = (orc_est **)calloc((size_t)n_genomes ? (size_t)n_genomes : 1, sizeof(orc_est *));
    for (int32_t g = 0; g < n_genomes; g++) per_genome[g] = est_new(params, ne);   /* :92-97 */
    u64 *reads_in_genome = (u64 *)calloc((size_t)n_genomes ? (size_t)n_genomes : 1, sizeof(u64));
    uint8_t *seen = (uint8_t *)calloc((size_t)n_targets ? (size_t)n_targets : 1, 1); /* seen_ref_ids */
    uint32_t last_tid = 0;
    int doing_first = 1;
    int32_t *ud = NULL; size_t L = 0;
    u64 n_in_contig = 0, edit_total = 0, indels = 0;
    double sum_identity = 0.0;
    int rc = ORC_OK;
    for (u64 oi = 0; oi < r->n_order; oi++) {
        u64 i = r->order ? r->order[oi] : oi;
        uint16_t flag = r->flag[i];
        if (!flag_passes(ff, flag)) continue;
        if (flag & 0x4) continue;
        uint32_t tid = (uint32_t)r->tid[i];
        if (tid != last_tid || doing_first) {
            if (doing_first) doing_first = 0;
            else {
                if (tid < last_tid) { rc = ORC_ERR_UNSORTED; goto done; }
                int32_t g = genome_of_tid[last_tid];
                if (g >= 0)
                    for (size_t k = 0; k < ne; k++)
                        est_add_contig(&per_genome[g][k], ud, L, n_in_contig, edit_total - indels, sum_identity);
            }
            free(ud);
            L = (size_t)target_len[tid];
            ud = (int32_t *)calloc(L ? L : 1, sizeof(int32_t));
            n_in_contig = 0; edit_total = 0; indels = 0; sum_identity = 0.0;
            last_tid = tid;
            seen[tid] = 1;
        }
        int32_t g = genome_of_tid[tid];
        if (g >= 0) {                                        /* :170-225 */
            reads_in_genome[g] += 1;
            n_in_contig += 1;
            u64 aligned_len = 0;
            rc = cigar_walk(r, i, ud, L, &indels, &aligned_len);
            if (rc) goto done;
            u64 edit;
            rc = nm_of(r, i, &edit);
            if (rc) goto done;
            edit_total += edit;
            if (!(flag & 0x800) && aligned_len > 0)          /* :220 */
                sum_identity += ((double)aligned_len - (double)edit) / (double)aligned_len;
        }
    }
    {
        u64 mapped_total = 0;
        if (doing_first && num_detected_primary == 0) {
            /* warn only, :230-234 */
        } else {
            int32_t g = genome_of_tid[last_tid];             /* :237-248 */
            if (g >= 0)
                for (size_t k = 0; k < ne; k++)
                    est_add_contig(&per_genome[g][k], ud, L, n_in_contig, edit_total - indels, sum_identity);
            for (int32_t gi = 0; gi < n_genomes; gi++) {     /* :252-302 */
                size_t n_unobs = 0;
                u64 *unobs = (u64 *)malloc(sizeof(u64) * ((size_t)n_targets ? (size_t)n_targets : 1));
                u64 genome_len = 0;
                for (int32_t t = 0; t < n_targets; t++)
                    if (genome_of_tid[t] == gi) {
                        genome_len += (u64)target_len[t];
                        if (!seen[t]) unobs[n_unobs++] = (u64)target_len[t];
                    }
                float *cov = (float *)malloc(sizeof(float) * (ne ? ne : 1));
                int nonzero = 0;
                for (size_t k = 0; k < ne; k++) {
                    cov[k] = est_calculate(&per_genome[gi][k], unobs, n_unobs);
                    if (cov[k] > 0.0f) nonzero = 1;
                }
                if (nonzero) mapped_total += reads_in_genome[gi];
                if (print_zero || nonzero) {
                    em_start(out, gi, -1 - gi); /* name = genomes[gi] */
                    for (size_t k = 0; k < ne; k++) {
                        if (cov[k] > 0.0f) est_print_coverage(&per_genome[gi][k], cov[k], out);
                        else est_print_zero(&per_genome[gi][k], out, genome_len);
                    }
                    em_finish(out);
                }
                free(cov); free(unobs);
            }
        }
        rm->num_mapped_reads = mapped_total;
        rm->num_reads = num_detected_primary;
    }
done:
    free(ud); free(seen); free(reads_in_genome);
    for (int32_t g = 0; g < n_genomes; g++) est_free(per_genome[g], ne);
    free(per_genome);
    return rc;
}

/* ------------------------------------------------------------------ genome.rs:419-929 */
typedef struct {
    const char *names;          /* concatenated target names */
    const uint32_t *name_off;   /* n_targets + 1 */
    const i64 *target_len;
    int32_t n_targets;
    uint8_t split_char;
    int single_genome;
    int *err;
} sep_hdr;

typedef struct { const char *p; size_t n; int is_none; } bslice;

/* genome.rs:799-805 */
static bslice extract_genome(const sep_hdr *h, uint32_t tid) {
    const char *name = h->names + h->name_off[tid];
    size_t n = h->name_off[tid + 1] - h->name_off[tid];
    const char *q = (const char *)memchr(name, h->split_char, n);
    bslice s; s.is_none = 0;
    if (!q) { *h->err = ORC_ERR_NO_SEPARATOR; s.p = name; s.n = n; return s; }
    s.p = name; s.n = (size_t)(q - name);
    return s;
}
static int bs_eq(bslice a, bslice b) { return a.n == b.n && memcmp(a.p, b.p, a.n) == 0; }

typedef struct { u64 *v; size_t n, cap; size_t first_tid; } unobs_vec;
static void uv_push(unobs_vec *u, u64 x) {
    if (u->n == u->cap) { u->cap = u->cap ? u->cap * 2 : 16; u->v = (u64 *)realloc(u->v, u->cap * sizeof(u64)); }
    u->v[u->n++] = x;
}

/* genome.rs:807-853 */
static void fill_genome_length_backwards(const sep_hdr *h, uint32_t current_tid, bslice target_genome, unobs_vec *u) {
    u->n = 0;
    if (current_tid == 0) { u->first_tid = 0; return; }
    uint32_t my_tid = current_tid - 1;
    while (h->single_genome || bs_eq(extract_genome(h, my_tid), target_genome)) {
        uv_push(u, (u64)h->target_len[my_tid]);
        if (my_tid == 0) { u->first_tid = 0; return; }
        my_tid--;
    }
    u->first_tid = (size_t)my_tid + 1;
}
/* genome.rs:477-499 */
static void fill_genome_length_backwards_to_last(const sep_hdr *h, uint32_t current_tid, uint32_t last_tid,
                                                 bslice target_genome, unobs_vec *u) {
    if (current_tid == 0) return;
    uint32_t my_tid = last_tid + 1;
    while (my_tid < current_tid) {
        if (h->single_genome || bs_eq(extract_genome(h, my_tid), target_genome)) {
            uv_push(u, (u64)h->target_len[my_tid]);
            my_tid++;
        } else break;
    }
}
/* genome.rs:448-475 */
static void fill_genome_length_forwards(const sep_hdr *h, uint32_t current_tid, bslice target_genome, unobs_vec *u) {
    if (target_genome.is_none) return;
    uint32_t my_tid = current_tid + 1;
    while (my_tid < (uint32_t)h->n_targets) {
        if (h->single_genome || bs_eq(extract_genome(h, my_tid), target_genome)) {
            uv_push(u, (u64)h->target_len[my_tid]);
            my_tid++;
        } else break;
    }
}

/* genome.rs:859-929.  Entry names: genome prefix of the tid recorded in name_tid. */
static void print_previous_zero_coverage_genomes2(const sep_hdr *h, bslice last_genome, bslice current_genome,
                                                  uint32_t current_tid, const orc_est *est, size_t ne, orc_out *out) {
    bslice my_current_genome = current_genome;
    uint32_t tid = current_tid;
    size_t cap = 16, n = 0;
    size_t *first_tids = (size_t *)malloc(cap * sizeof(size_t));
    uint32_t *name_tids = (uint32_t *)malloc(cap * sizeof(uint32_t));
    u64 *unobs_len = (u64 *)malloc(cap * sizeof(u64));
    u64 unobserved_length = 0;
    int have_last_first = 0; uint32_t last_first_id = 0;
    uint32_t my_current_name_tid = current_tid;
#define PUSH_GENOME(ID, NAMETID, LEN) do { if (n == cap) { cap *= 2; \
        first_tids = (size_t *)realloc(first_tids, cap * sizeof(size_t)); \
        name_tids = (uint32_t *)realloc(name_tids, cap * sizeof(uint32_t)); \
        unobs_len = (u64 *)realloc(unobs_len, cap * sizeof(u64)); } \
        first_tids[n] = (ID); name_tids[n] = (NAMETID); unobs_len[n] = (LEN); n++; } while (0)
    for (;;) {
        bslice genome = extract_genome(h, tid);
        if (!last_genome.is_none && bs_eq(genome, last_genome)) break;
        else if (!bs_eq(genome, my_current_genome)) {
            if (have_last_first) {
                if (last_genome.is_none || !bs_eq(genome, last_genome))
                    PUSH_GENOME(last_first_id, my_current_name_tid, unobserved_length);
            }
            my_current_genome = genome; my_current_name_tid = tid;
            have_last_first = 1; last_first_id = tid;
            unobserved_length = (u64)h->target_len[tid];
        } else if (!bs_eq(genome, current_genome)) {
            have_last_first = 1; last_first_id = tid;
            unobserved_length += (u64)h->target_len[tid];
        }
        if (tid == 0) break;
        tid--;
    }
    if (have_last_first) PUSH_GENOME(last_first_id, my_current_name_tid, unobserved_length);
#undef PUSH_GENOME
    for (size_t i = n; i-- > 0;) {
        em_start(out, (i64)first_tids[i], (int32_t)name_tids[i]);
        for (size_t k = 0; k < ne; k++) est_print_zero(&est[k], out, unobs_len[i]);
        em_finish(out);
    }
    free(first_tids); free(name_tids); free(unobs_len);
}

/* genome.rs:331-416.  last_genome_name_tid: a tid whose genome prefix is last_genome
 * (or -1000000 for the single-genome dummy name "genome1", genome.rs:739-741). */
static int print_last_genomes(const sep_hdr *h, u64 n_in_contig, bslice last_genome, int32_t last_genome_name_tid,
                              unobs_vec *u, const int32_t *ud, size_t L, u64 edit_total, u64 indels,
                              double sum_identity, bslice current_genome, orc_est *est, size_t ne, orc_out *out,
                              int print_zero, uint32_t tid_to_print_zeros_to) {
    for (size_t k = 0; k < ne; k++) est_add_contig(&est[k], ud, L, n_in_contig, edit_total - indels, sum_identity);
    float *cov = (float *)malloc(sizeof(float) * (ne ? ne : 1));
    int positive = 0;
    for (size_t k = 0; k < ne; k++) {
        cov[k] = est_calculate(&est[k], u->v, u->n);
        if (cov[k] > 0.0f) positive = 1;
    }
    if (print_zero || positive) {
        if (!last_genome.is_none) {
            em_start(out, (i64)u->first_tid, last_genome_name_tid);
            for (size_t k = 0; k < ne; k++) {
                if (cov[k] > 0.0f) est_print_coverage(&est[k], cov[k], out);
                else est_print_zero(&est[k], out, 9);
            }
            em_finish(out);
        }
    }
    for (size_t k = 0; k < ne; k++) est_setup(&est[k]);
    if (print_zero && !h->single_genome)
        print_previous_zero_coverage_genomes2(h, last_genome, current_genome, tid_to_print_zeros_to, est, ne, out);
    free(cov);
    return positive;
}

#define ORC_NAME_GENOME1 (-1000000)

int orc_genome_coverage_separator(const orc_records *r, const char *names, const uint32_t *name_off,
                                  const i64 *target_len, int32_t n_targets, uint8_t split_char,
                                  int32_t single_genome, const orc_est_param *params, int32_t n_est,
                                  int32_t print_zero, const orc_flag_filter *ff, u64 num_detected_primary,
                                  orc_out *out, orc_reads_mapped *rm) {
    int err = ORC_OK;
    sep_hdr h; h.names = names; h.name_off = name_off; h.target_len = target_len; h.n_targets = n_targets;
    h.split_char = split_char; h.single_genome = single_genome; h.err = &err;
    size_t ne = (size_t)n_est;
    orc_est *est = est_new(params, ne);
    uint32_t last_tid = 0;
    int doing_first = 1;
    bslice last_genome; last_genome.p = ""; last_genome.n = 0; last_genome.is_none = 1;
    int32_t last_genome_name_tid = -1;
    unobs_vec u; memset(&u, 0, sizeof u);
    int32_t *ud = NULL; size_t L = 0;
    u64 mapped_total = 0, n_in_contig = 0, n_in_genome = 0, edit_total = 0, indels = 0;
    double sum_identity = 0.0;
    int rc = ORC_OK;
    bslice empty; empty.p = ""; empty.n = 0; empty.is_none = 0;
    for (u64 oi = 0; oi < r->n_order; oi++) {
        u64 i = r->order ? r->order[oi] : oi;
        uint16_t flag = r->flag[i];
        if (!flag_passes(ff, flag)) continue;
        if (flag & 0x4) continue;
        uint32_t tid = (uint32_t)r->tid[i];
        bslice current_genome = single_genome ? empty : extract_genome(&h, tid);
        if (err) { rc = err; goto done; }
        if (tid != last_tid || doing_first) {
            if (!doing_first && tid < last_tid) { rc = ORC_ERR_UNSORTED; goto done; }
            if (doing_first) {
                for (size_t k = 0; k < ne; k++) est_setup(&est[k]);
                fill_genome_length_backwards(&h, tid, current_genome, &u);
                last_genome = current_genome; last_genome_name_tid = (int32_t)tid;
                doing_first = 0;
                if (print_zero && !single_genome) {
                    bslice none; none.p = ""; none.n = 0; none.is_none = 1;
                    print_previous_zero_coverage_genomes2(&h, none, current_genome, tid, est, ne, out);
                }
            } else if (bs_eq(current_genome, last_genome)) {
                for (size_t k = 0; k < ne; k++)
                    est_add_contig(&est[k], ud, L, n_in_contig, edit_total - indels, sum_identity);
                fill_genome_length_backwards_to_last(&h, tid, last_tid, current_genome, &u);
            } else {
                fill_genome_length_backwards_to_last(&h, tid, last_tid, last_genome, &u);
                int positive = print_last_genomes(&h, n_in_contig, last_genome, last_genome_name_tid, &u, ud, L,
                                                  edit_total, indels, sum_identity, current_genome, est, ne, out,
                                                  print_zero, tid);
                if (positive) mapped_total += n_in_genome;
                n_in_genome = 0;
                last_genome = current_genome; last_genome_name_tid = (int32_t)tid;
                fill_genome_length_backwards(&h, tid, current_genome, &u);
            }
            if (err) { rc = err; goto done; }
            free(ud);
            L = (size_t)target_len[tid];
            ud = (int32_t *)calloc(L ? L : 1, sizeof(int32_t));
            n_in_contig = 0; edit_total = 0; indels = 0; sum_identity = 0.0;
            last_tid = tid;
        }
        if (!(flag & 0x800)) { n_in_contig++; n_in_genome++; }   /* :677-682 */
        u64 aligned_len = 0;
        rc = cigar_walk(r, i, ud, L, &indels, &aligned_len);
        if (rc) goto done;
        u64 edit;
        rc = nm_of(r, i, &edit);
        if (rc) goto done;
        edit_total += edit;
        if (!(flag & 0x800) && !(flag & 0x100) && aligned_len > 0)  /* :724 */
            sum_identity += ((double)aligned_len - (double)edit) / (double)aligned_len;
    }
    if (doing_first && num_detected_primary == 0) {
        /* warn only, :731-735 */
    } else {
        if (single_genome) { last_genome = empty; last_genome_name_tid = ORC_NAME_GENOME1; } /* "genome1" */
        fill_genome_length_forwards(&h, last_tid, last_genome, &u);
        int positive = print_last_genomes(&h, n_in_contig, last_genome, last_genome_name_tid, &u, ud, L, edit_total,
                                          indels, sum_identity, empty, est, ne, out, print_zero,
                                          (uint32_t)(n_targets - 1));
        if (err) { rc = err; goto done; }
        if (positive) mapped_total += n_in_genome;
    }
    rm->num_mapped_reads = mapped_total;
    rm->num_reads = num_detected_primary;
done:
    free(ud); free(u.v);
    est_free(est, ne);
    return rc;
}

/* ------------------------------------------------------------------ helpers for tests */
void orc_out_free(orc_out *o) { free(o->e); o->e = NULL; o->n = o->cap = 0; }

/* Per-contig delta array exactly as contig.rs:144-202 builds it (no pad slot), for depth parity
 * checks.  `ud` has target_len[want_tid] entries, zeroed by the caller. */
int orc_contig_deltas(const orc_records *r, const i64 *target_len, const orc_flag_filter *ff, int32_t want_tid,
                      int32_t *ud) {
    size_t L = (size_t)target_len[want_tid];
    for (u64 oi = 0; oi < r->n_order; oi++) {
        u64 i = r->order ? r->order[oi] : oi;
        uint16_t flag = r->flag[i];
        if (!flag_passes(ff, flag) || (flag & 0x4) || r->tid[i] != want_tid) continue;
        u64 indels = 0, aligned = 0;
        int rc = cigar_walk(r, i, ud, L, &indels, &aligned);
        if (rc) return rc;
    }
    return ORC_OK;
}

/* Standalone estimator access (add_contig + calculate_coverage on a caller-supplied delta array);
 * used by property tests and by the cpu_baseline timing of HOT LOOP B alone. */
float orc_estimate_one(const orc_est_param *p, const int32_t *ud, u64 len, u64 n_reads, u64 mismatches,
                       double sum_identity, const u64 *unobs, u64 n_unobs) {
    orc_est *e = est_new(p, 1);
    est_add_contig(e, ud, (size_t)len, n_reads, mismatches, sum_identity);
    float c = est_calculate(e, unobs, (size_t)n_unobs);
    est_free(e, 1);
    return c;
}

/* As orc_estimate_one, but through print_coverage (estimators.rs:936-969), so that PileupCounts emits its (depth, bases)
 * entries: what genes.rs:536-549 does for every gene.  Returns the coverage; `out` receives the emits. */
float orc_estimate_emit(const orc_est_param *p, const int32_t *ud, u64 len, u64 n_reads, u64 mismatches,
                        double sum_identity, const u64 *unobs, u64 n_unobs, orc_out *out) {
    orc_est *e = est_new(p, 1);
    est_add_contig(e, ud, (size_t)len, n_reads, mismatches, sum_identity);
    float c = est_calculate(e, unobs, (size_t)n_unobs);
    est_print_coverage(e, c, out);
    est_free(e, 1);
    return c;
}

/* ------------------------------------------------------------------ integer sufficient statistics
 * What the device must return per contig (covermhip.h cov_contig_stats), computed the reference's way:
 * one delta array per contig built by the CIGAR walk (contig.rs:144-202), then a sequential prefix sum
 * with the window test of estimators.rs:393-404 / 447-465 and the full-length loop of :496-501.
 * mask[t] == 0 reproduces genome.rs:170-171 (contigs outside every genome: counted as seen, not walked). */
typedef struct {
    u64 n_primary, n_pass, n_nonsupp, sum_nm, sum_indel;
    double id_primary, id_nonsupp;
    u64 win_sum_d, win_sum_d2, win_covered, full_covered, first_record, last_record;
    uint32_t win_min_d, win_max_d, hist_len, seen;
    u64 hist_off;
} orc_stats;

int orc_integer_stats(const orc_records *r, const i64 *target_len, int32_t n_targets, const uint8_t *mask,
                      const orc_flag_filter *ff, u64 excl, orc_stats *out, u64 **hist_out, u64 *hist_total) {
    memset(out, 0, sizeof(orc_stats) * (size_t)(n_targets ? n_targets : 1));
    u64 *hist = NULL; size_t hn = 0, hcap = 0;
    int32_t last_tid = -2;
    int32_t *ud = NULL; size_t L = 0;
    int rc = ORC_OK;
    for (u64 oi = 0; oi <= r->n_order; oi++) {
        int flush = (oi == r->n_order);
        u64 i = 0; uint16_t flag = 0; int32_t tid = -1;
        if (!flush) {
            i = r->order ? r->order[oi] : oi;
            flag = r->flag[i];
            if (!flag_passes(ff, flag) || (flag & 0x4)) continue;
            tid = r->tid[i];
            if (tid != last_tid) {
                if (tid < last_tid) { rc = ORC_ERR_UNSORTED; goto done; }
                flush = 1;
            }
        }
        if (flush && last_tid >= 0 && (!mask || mask[last_tid])) {
            orc_stats *o = &out[last_tid];
            int32_t c = 0;
            for (size_t p = 0; p < L; p++) { c += ud[p]; if (c > 0) o->full_covered++; }
            if (2 * excl < (u64)L) {
                size_t s = (size_t)excl, e = L - (size_t)excl - 1;
                c = 0;
                uint32_t mn = 0xffffffffu, mx = 0;
                for (size_t p = 0; p < L; p++) {
                    c += ud[p];
                    if (p >= s && p <= e) {
                        if (c > 0) o->win_covered++;
                        o->win_sum_d += (u64)(i64)c;
                        o->win_sum_d2 += (u64)(i64)c * (u64)(i64)c;
                        if ((uint32_t)c < mn) mn = (uint32_t)c;
                        if ((uint32_t)c > mx) mx = (uint32_t)c;
                    }
                }
                o->win_min_d = mn; o->win_max_d = mx; o->hist_len = mx + 1;
                o->hist_off = hn;
                if (hn + mx + 1 > hcap) { hcap = (hn + mx + 1) * 2; hist = (u64 *)realloc(hist, hcap * sizeof(u64)); }
                memset(hist + hn, 0, (size_t)(mx + 1) * sizeof(u64));
                c = 0;
                for (size_t p = 0; p < L; p++) { c += ud[p]; if (p >= s && p <= e) hist[hn + (size_t)c]++; }
                hn += mx + 1;
            }
        }
        if (oi == r->n_order) break;
        if (tid != last_tid) {
            free(ud);
            L = (size_t)target_len[tid];
            ud = (int32_t *)calloc(L ? L : 1, sizeof(int32_t));
            last_tid = tid;
            out[tid].first_record = i;
            out[tid].seen = 1;
        }
        orc_stats *o = &out[tid];
        o->last_record = i;
        o->n_pass++;
        int primary = !(flag & 0x800) && !(flag & 0x100);
        if (primary) o->n_primary++;
        if (!(flag & 0x800)) o->n_nonsupp++;
        if (mask && !mask[tid]) continue;
        u64 indels = 0, aligned = 0;
        rc = cigar_walk(r, i, ud, L, &indels, &aligned);
        if (rc) goto done;
        u64 edit;
        rc = nm_of(r, i, &edit);
        if (rc) goto done;
        o->sum_nm += edit; o->sum_indel += indels;
        if (aligned > 0) {
            double idv = ((double)aligned - (double)edit) / (double)aligned;
            if (primary) o->id_primary += idv;
            if (!(flag & 0x800)) o->id_nonsupp += idv;
        }
    }
done:
    free(ud);
    *hist_out = hist; *hist_total = hn;
    return rc;
}
void orc_free(void *p) { free(p); }
