// Device side of libcovermhip: hand-written gfx950 (CDNA4, wave64) kernels for the CoverM pileup path.
//
// Pipeline over one sample's records (already resident in HBM as SoA, see covermhip.h cov_batch):
//
//   k_prep      one thread / record : reader-stage filter (filter.rs:88-116,243-279), FlagFilter
//                                     (lib.rs:67-78), CIGAR summary (contig.rs:166-202), per-contig
//                                     record counters; emits an 8-byte "run word" per record
//   k_ranges    one thread / tile   : candidate record range of each 16384-base tile (binary search
//                                     on the sorted start positions)
//   k_pileup    one workgroup / tile: +1/-1 events accumulated with LDS atomics into a 64 KiB i32
//                                     tile that never leaves the CU, block-wide prefix sum, window /
//                                     full-length statistics, change-point histogram in LDS
//   k_identity  one wave / contig   : file-order f64 identity sums (anir)
//   k_hist_*    histogram layout, zero-fill and compaction
//
// No MFMA: there is no contraction on this path.  The per-base depth array — the reference's
// Vec<i32> ups_and_downs (contig.rs:144) and its prefix sum (estimators.rs:393-404) — exists only
// in LDS, so HBM traffic is the record stream itself (DESIGN.md "Algorithmic bytes").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace covk {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int TILE = 16384;   // bases per tile: 64 KiB of i32 in LDS, two workgroups per CU
constexpr int HB = 2048;      // LDS histogram bins; deeper positions go straight to the global histogram
constexpr u32 F_POS_UNSORTED = 1u;

// Per-contig device accumulators (128 B).
struct DevContig {
    u64 n_primary, n_pass, n_nonsupp, sum_nm, sum_indel;   // k_prep
    double id_primary, id_nonsupp;                          // k_identity
    u64 sum_d, sum_d2, cov_win, cov_full, proc_win;         // k_pileup
    u64 hist_off;                                           // k_hist_layout
    u32 first_rec, last_rec;                                // considered records (file order)
    u32 rec_start, rec_end;                                 // span of ALL records carrying this tid
    u32 n_groups;                                           // number of maximal runs of this tid in file order
    u32 max_span;                                           // longest reference span of a considered record
    u32 flags;
    u32 hist_cap;                                           // upper bound on depth (max candidate count of a tile)
    u32 min_d, max_d;                                       // k_pileup, window positions of processed tiles
    u32 hist_len;                                           // k_hist_compact_layout
    u32 pad;
    u64 chist_off;                                          // offset in the compact histogram
};
static_assert(sizeof(DevContig) == 160, "DevContig layout");

struct DevGlobal {
    u64 n_primary_all;   // bam_generator.rs:114-118 / filter.rs:94-96
    u64 n_considered;
    u64 first_error;     // min over erroring records of (record_index << 8 | code); ~0 = none
    u64 hist_cap_total;  // arena bins in use
    u64 chist_total;     // compact histogram bins
    u32 internal_error;  // depth exceeded its proven bound (would indicate a bug), etc.
    u32 pad;
};

struct FilterCfg {
    u32 include_improper_pairs, include_supplementary, include_secondary;
    u32 filter_single, min_mapq, min_aligned_length;
    float min_percent_identity, min_aligned_percent;
};

struct Records {
    const int32_t *tid, *pos;
    const uint16_t *flag;
    const uint8_t *mapq;
    const u32 *nm;
    const uint8_t *nm_kind;
    const u32 *l_seq;
    const u32 *cigar_off;
    const u32 *cigar;
    u32 n;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane_id() >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (u32)__shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (u32)__shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ void report_error(DevGlobal *g, u32 rec, u32 code) {
    atomicMin(&g->first_error, ((u64)rec << 8) | code);
}

// ------------------------------------------------------------------------------------ k_init
__global__ void k_init(DevContig *ctg, u32 n_targets, DevGlobal *g) {
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) {
        g->n_primary_all = 0; g->n_considered = 0; g->first_error = ~0ull; g->hist_cap_total = 0;
        g->chist_total = 0; g->internal_error = 0;
    }
    if (c >= n_targets) return;
    DevContig z;
    z.n_primary = z.n_pass = z.n_nonsupp = z.sum_nm = z.sum_indel = 0;
    z.id_primary = z.id_nonsupp = 0.0;
    z.sum_d = z.sum_d2 = z.cov_win = z.cov_full = z.proc_win = 0;
    z.hist_off = 0;
    z.first_rec = 0xffffffffu; z.last_rec = 0;
    z.rec_start = 0xffffffffu; z.rec_end = 0;
    z.n_groups = 0; z.max_span = 0; z.flags = 0; z.hist_cap = 0;
    z.min_d = 0xffffffffu; z.max_d = 0; z.hist_len = 0; z.pad = 0; z.chist_off = 0;
    ctg[c] = z;
}

// ------------------------------------------------------------------------------------ k_prep
// Run word: x = start of the first merged M/=/X run (contig coordinate), y = its length with bit 31
// set when the record has further runs (a D or N gap), in which case k_pileup re-walks the CIGAR.
// Adjacent runs separated only by I/S/H/P are merged: their +1/-1 events cancel (contig.rs:178-183).
template <bool WANT_IDENTITY>
__global__ __launch_bounds__(256) void k_prep(Records r, const u32 *__restrict__ tlen, u32 n_targets,
                                              const uint8_t *__restrict__ mask, FilterCfg f, DevContig *ctg,
                                              DevGlobal *g, uint2 *__restrict__ runs, double *__restrict__ ident) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < r.n;
    const u32 flag = in ? r.flag[i] : 0x904u;
    const int tid = in ? r.tid[i] : -1;
    const int pos = in ? r.pos[i] : 0;
    const int lane = lane_id();

    // every record read counts towards num_detected_primary_alignments when !secondary && !supplementary
    {
        u64 b = __ballot(in && !(flag & 0x900u));
        if (lane == 0 && b) atomicAdd(&g->n_primary_all, (u64)__popcll(b));
    }
    const bool tid_ok = in && tid >= 0 && (u32)tid < n_targets;
    // span of records carrying this tid + grouping / position-order checks (all records, considered or not)
    if (tid_ok) {
        const int ptid = i > 0 ? r.tid[i - 1] : -2;
        if (ptid != tid) {
            atomicMin(&ctg[tid].rec_start, i);
            atomicAdd(&ctg[tid].n_groups, 1u);
        } else if (r.pos[i - 1] > pos) {
            atomicOr(&ctg[tid].flags, F_POS_UNSORTED);
        }
        const int ntid = (i + 1 < r.n) ? r.tid[i + 1] : -2;
        if (ntid != tid) atomicMax(&ctg[tid].rec_end, i + 1);
    }

    const bool unmapped = flag & 0x4u;
    const bool supp = flag & 0x800u, sec = flag & 0x100u;
    // reader stage: ReferenceSortedBamFilter::read single-read branch, filter_out = true (filter.rs:88-116)
    bool survives = in;
    bool need_filter_eval = false;
    if (f.filter_single) {
        survives = false;
        const bool p1 = in && !unmapped && (f.include_supplementary || !supp) && (f.include_secondary || !sec);
        if (p1) {
            const u32 mq = r.mapq[i];
            if (!(f.min_mapq != 255u && (mq < f.min_mapq || mq == 255u))) need_filter_eval = true;  // :250-254
        }
    }
    // scan stage gate: FlagFilter::passes (lib.rs:67-78) then !unmapped (contig.rs:125)
    const bool flags_ok = !(!f.include_secondary && sec) && !(!f.include_supplementary && supp) &&
                          !(!f.include_improper_pairs && !(flag & 0x2u));
    const bool scan_gate = in && flags_ok && !unmapped;

    u64 aligned = 0, indel = 0;
    u32 run_start = 0, run_len = 0, n_runs = 0, span = 0;
    bool oob = false, badcig = false;
    // with the reader-stage filter on, only records that reach single_read_passes_filter can survive
    const bool do_walk = f.filter_single ? need_filter_eval : scan_gate;
    if (do_walk) {
        const u32 L = tid_ok ? tlen[tid] : 0u;
        long long cursor = pos;
        long long cur_s = 0, cur_e = -1;  // open merged run [cur_s, cur_e)
        const u32 c0 = r.cigar_off[i], c1 = r.cigar_off[i + 1];
        for (u32 c = c0; c < c1; c++) {
            const u32 w = r.cigar[c];
            const u32 op = w & 15u, len = w >> 4;
            if (op == 0u || op == 7u || op == 8u) {          // M = X  (contig.rs:171-186)
                if (cursor < 0 || cursor >= (long long)L) oob = true;
                if (n_runs > 0 && cursor == cur_e) {
                    cur_e += len;
                    if (n_runs == 1) run_len += len;
                } else {
                    n_runs++;
                    cur_s = cursor; cur_e = cursor + len;
                    if (n_runs == 1) { run_start = (u32)cursor; run_len = len; }
                }
                cursor += len; aligned += len;
            } else if (op == 2u) { cursor += len; indel += len; aligned += len; }  // D  (:187-191)
            else if (op == 3u) { cursor += len; }                                  // N  (:192-195)
            else if (op == 1u) { indel += len; aligned += len; }                   // I  (:196-199)
            else if (op > 8u) badcig = true;                                       // S H P ignored (:200)
        }
        (void)cur_s;
        const long long sp = cursor - (long long)pos;
        span = sp > 0 ? (u32)min(sp, (long long)0xffffffffu) : 0u;
    }
    if (need_filter_eval) {  // single_read_passes_filter, filter.rs:256-278
        const u32 k = r.nm_kind[i];
        if (k != 1u) report_error(g, i, k == 0u ? 2u : 3u);
        else {
            const u32 al = (u32)aligned;  // u32 accumulation in the reference
            const float a = (float)al;
            survives = al >= f.min_aligned_length && a / (float)r.l_seq[i] >= f.min_aligned_percent &&
                       1.0f - (float)r.nm[i] / a >= f.min_percent_identity;
        }
    }
    const bool considered = survives && scan_gate;
    const bool masked_in = considered && tid_ok && (mask == nullptr || mask[tid]);
    u64 nmv = 0;
    double idv = 0.0;
    if (considered && !tid_ok) report_error(g, i, 7u);  // header.target_len(tid).expect("Corrupt BAM file?")
    if (masked_in) {
        if (badcig) report_error(g, i, 6u);
        else if (oob) report_error(g, i, 4u);
        const u32 k = r.nm_kind[i];
        if (k != 1u) report_error(g, i, k == 0u ? 2u : 3u);  // nm(&record), contig.rs:206
        else nmv = r.nm[i];
        if (WANT_IDENTITY && aligned > 0) idv = ((double)aligned - (double)nmv) / (double)aligned;
    }
    if (in) {
        uint2 rw;
        rw.x = masked_in ? run_start : 0u;
        rw.y = masked_in ? ((run_len & 0x7fffffffu) | (n_runs > 1 ? 0x80000000u : 0u)) : 0u;
        if (masked_in && n_runs > 1 && run_len == 0) rw.y = 0x80000000u;
        runs[i] = rw;
        if (WANT_IDENTITY) ident[i] = (masked_in && !supp) ? idv : 0.0;
    }

    // per-contig counters, one atomic per wave when every considered lane shares a tid (the common case)
    const bool cnt = considered && tid_ok;
    const u64 m = __ballot(cnt);
    if (m == 0) return;
    const int first = __ffsll((long long)m) - 1;
    const int ftid = __shfl(tid, first);
    const bool uni = __all(!cnt || tid == ftid);
    const u32 base = i - lane;
    if (uni) {
        const u32 c_prim = __popcll(__ballot(cnt && !supp && !sec));
        const u32 c_nons = __popcll(__ballot(cnt && !supp));
        const u32 c_pass = __popcll(m);
        const u64 s_nm = wave_sum_u64(masked_in ? nmv : 0ull);
        const u64 s_in = wave_sum_u64(masked_in ? indel : 0ull);
        const u32 mx = wave_max_u32(masked_in ? span : 0u);
        if (lane == first) {
            DevContig *C = &ctg[ftid];
            atomicAdd(&C->n_primary, (u64)c_prim);
            atomicAdd(&C->n_pass, (u64)c_pass);
            atomicAdd(&C->n_nonsupp, (u64)c_nons);
            if (s_nm) atomicAdd(&C->sum_nm, s_nm);
            if (s_in) atomicAdd(&C->sum_indel, s_in);
            atomicMax(&C->max_span, mx);
            atomicMin(&C->first_rec, base + (u32)first);
            atomicMax(&C->last_rec, base + (u32)(63 - __clzll((long long)m)));
            atomicAdd(&g->n_considered, (u64)c_pass);
        }
    } else if (cnt) {
        DevContig *C = &ctg[tid];
        if (!supp && !sec) atomicAdd(&C->n_primary, 1ull);
        atomicAdd(&C->n_pass, 1ull);
        if (!supp) atomicAdd(&C->n_nonsupp, 1ull);
        if (masked_in) {
            if (nmv) atomicAdd(&C->sum_nm, nmv);
            if (indel) atomicAdd(&C->sum_indel, indel);
            atomicMax(&C->max_span, span);
        }
        atomicMin(&C->first_rec, i);
        atomicMax(&C->last_rec, i);
        atomicAdd(&g->n_considered, 1ull);
    }
}

// ------------------------------------------------------------------------------------ k_ranges
__device__ __forceinline__ u32 lower_bound_pos(const int32_t *__restrict__ pos, u32 lo, u32 hi, long long key) {
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if ((long long)pos[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <bool WANT_HIST>
__global__ __launch_bounds__(256) void k_ranges(const u32 *__restrict__ tile_contig, const u32 *__restrict__ tile_start,
                                                u32 n_tiles, const int32_t *__restrict__ pos,
                                                const uint8_t *__restrict__ mask, DevContig *ctg,
                                                uint2 *__restrict__ cand) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const u32 c = tile_contig[t];
    DevContig *C = &ctg[c];
    uint2 out = make_uint2(0u, 0u);
    if (C->n_pass != 0 && (mask == nullptr || mask[c])) {
        const u32 rs = C->rec_start, re = C->rec_end;
        if (C->n_groups != 1u || (C->flags & F_POS_UNSORTED)) {
            out = make_uint2(rs, re);  // generic path: every tile of this contig scans the whole span
        } else {
            const long long lo = tile_start[t];
            // a run [s,e) of a record at `pos` (pos <= s, e <= pos + max_span) overlaps [lo, lo+TILE) only if
            // pos > lo - max_span and pos < lo + TILE
            out.x = lower_bound_pos(pos, rs, re, lo - (long long)C->max_span + 1);
            out.y = lower_bound_pos(pos, out.x, re, lo + TILE);
        }
        if (WANT_HIST && out.y > out.x) atomicMax(&C->hist_cap, out.y - out.x);
    }
    cand[t] = out;
}

// ------------------------------------------------------------------------------------ histogram layout
// Single workgroup: exclusive scan of per-contig bin counts -> offsets.  MODE 0: arena layout from
// hist_cap (an upper bound on depth); MODE 1: compact layout from the realised max depth.
template <int MODE>
__global__ __launch_bounds__(1024) void k_hist_layout(DevContig *ctg, u32 n_targets, const u32 *__restrict__ tlen,
                                                      const uint8_t *__restrict__ mask, u64 excl, DevGlobal *g) {
    __shared__ u64 wtot[16];
    __shared__ u64 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (u32 base = 0; base < n_targets; base += 1024) {
        const u32 c = base + threadIdx.x;
        u64 v = 0;
        if (c < n_targets) {
            DevContig *C = &ctg[c];
            const bool live = C->n_pass != 0 && (mask == nullptr || mask[c]);
            if (MODE == 0) v = live ? (u64)C->hist_cap + 1 : 0;
            else {
                const bool has_win = 2 * excl < (u64)tlen[c];
                v = (live && has_win) ? (u64)C->max_d + 1 : 0;
                C->hist_len = (u32)v;
            }
        }
        u64 inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            u64 t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        u64 wbase = 0;
        for (int k = 0; k < w; k++) wbase += wtot[k];
        const u64 carry = carry_s;
        if (c < n_targets) {
            if (MODE == 0) ctg[c].hist_off = carry + wbase + inc - v;
            else ctg[c].chist_off = carry + wbase + inc - v;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wbase + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (MODE == 0) g->hist_cap_total = carry_s; else g->chist_total = carry_s;
    }
}

__global__ void k_zero_u32(u32 *__restrict__ p, const u64 *__restrict__ n_ptr) {
    const u64 n = *n_ptr;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = 0u;
}

// hist[chist_off + d] = arena[hist_off + d]; bin 0 also receives the window positions of tiles that had no
// candidate record (depth 0 everywhere, never visited by k_pileup).
__global__ __launch_bounds__(256) void k_hist_compact(const DevContig *__restrict__ ctg, u32 n_targets,
                                                      const u32 *__restrict__ tlen, u64 excl,
                                                      const u32 *__restrict__ arena, u64 *__restrict__ out) {
    const u32 c = blockIdx.x;
    const DevContig *C = &ctg[c];
    const u32 n = C->hist_len;
    if (n == 0) return;
    const u64 win_len = (u64)tlen[c] - 2 * excl;
    for (u32 d = threadIdx.x; d < n; d += blockDim.x) {
        u64 v = arena[C->hist_off + d];
        if (d == 0) v += win_len - C->proc_win;
        out[C->chist_off + d] = v;
    }
}

// ------------------------------------------------------------------------------------ k_identity
// File-order f64 sums, bit-identical to the reference's sequential `+=` (contig.rs:208-211): every lane
// of the wave walks the same serial chain, 64 records per coalesced load.
__global__ __launch_bounds__(64) void k_identity(DevContig *ctg, u32 n_targets, const double *__restrict__ ident,
                                                 const uint16_t *__restrict__ flag, const int32_t *__restrict__ tidv) {
    const u32 c = blockIdx.x;
    if (c >= n_targets) return;
    DevContig *C = &ctg[c];
    if (C->n_pass == 0) return;
    const u32 rs = C->rec_start, re = C->rec_end;
    const bool generic = C->n_groups != 1u;
    double accp = 0.0, accn = 0.0;
    const int lane = lane_id();
    for (u32 b = rs; b < re; b += 64) {
        const u32 i = b + lane;
        double x = 0.0;
        u32 fl = 0x100u;
        if (i < re && (!generic || tidv[i] == (int)c)) { x = ident[i]; fl = flag[i]; }
        const double xp = (fl & 0x100u) ? 0.0 : x;  // primary variant additionally excludes secondary
#pragma unroll 8
        for (int k = 0; k < 64; k++) {
            accn += __shfl(x, k);
            accp += __shfl(xp, k);
        }
    }
    if (lane == 0) { C->id_primary = accp; C->id_nonsupp = accn; }
}

// ------------------------------------------------------------------------------------ k_pileup
struct PileupArgs {
    const u32 *tile_contig, *tile_start;
    const uint2 *cand;
    const uint2 *runs;
    Records r;
    const u32 *tlen;
    DevContig *ctg;
    DevGlobal *g;
    u32 *hist_arena;
    u64 excl;
    int32_t *depth_out;   // WRITE_DEPTH: depth of one contig
    u32 tile_base;        // first tile index handled by blockIdx 0
};

template <int NT, bool WANT_HIST, bool WRITE_DEPTH>
__global__ __launch_bounds__(NT) void k_pileup(PileupArgs a) {
    constexpr int NW = NT / 64;
    constexpr int ROWS = TILE / (4 * NT);   // quads per thread
    static_assert(NW * ROWS == 64, "cross-wave prefix is resolved by one 64-lane scan");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *tile = reinterpret_cast<int *>(smem);                       // TILE i32
    int *wtot = reinterpret_cast<int *>(smem + TILE * 4);            // 64 wave totals
    u64 *red64 = reinterpret_cast<u64 *>(smem + TILE * 4 + 256);     // NW * 2
    u32 *red32 = reinterpret_cast<u32 *>(smem + TILE * 4 + 256 + NW * 16);  // NW * 4 (+2 broadcast)
    u32 *lhist = reinterpret_cast<u32 *>(smem + TILE * 4 + 256 + NW * 16 + NW * 16 + 16);

    const u32 t = a.tile_base + blockIdx.x;
    const uint2 cr = a.cand[t];
    if (cr.x >= cr.y) return;  // no record can touch this tile: depth 0 everywhere, accounted on the host side
    const u32 c = a.tile_contig[t];
    const u32 lo = a.tile_start[t];
    const u32 L = a.tlen[c];
    const u32 tlen_t = min((u32)TILE, L - lo);
    const bool generic = (a.ctg[c].n_groups != 1u);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rows_used = (int)((tlen_t + 4 * NT - 1) / (4 * NT));

    // ---- zero the tile (and the LDS histogram)
    {
        int4 z = make_int4(0, 0, 0, 0);
        int4 *t4 = reinterpret_cast<int4 *>(tile);
        for (int rr = 0; rr < rows_used; rr++) t4[rr * NT + tid] = z;
        if (WANT_HIST)
            for (int b = tid; b < HB; b += NT) lhist[b] = 0u;
    }
    __syncthreads();

    // ---- events: +1 at the (clipped) start, -1 at the end of every M/=/X run that overlaps the tile
    const u32 hi = lo + TILE;
    auto add_run = [&](u32 s, u32 e) {
        if (s < hi && e > lo) {
            const u32 s0 = s > lo ? s - lo : 0u;
            atomicAdd(&tile[s0], 1);
            if (e < hi) atomicAdd(&tile[e - lo], -1);
        }
    };
    for (u32 i = cr.x + tid; i < cr.y; i += NT) {
        const uint2 rw = a.runs[i];
        if (rw.y == 0u) continue;
        if (generic && a.r.tid[i] != (int)c) continue;
        if (!(rw.y & 0x80000000u)) {
            add_run(rw.x, rw.x + rw.y);
        } else {  // D / N gaps: re-walk the CIGAR (contig.rs:166-202)
            u32 cursor = (u32)a.r.pos[i];
            const u32 c0 = a.r.cigar_off[i], c1 = a.r.cigar_off[i + 1];
            for (u32 k = c0; k < c1; k++) {
                const u32 wd = a.r.cigar[k];
                const u32 op = wd & 15u, len = wd >> 4;
                if (op == 0u || op == 7u || op == 8u) { add_run(cursor, cursor + len); cursor += len; }
                else if (op == 2u || op == 3u) cursor += len;
            }
        }
    }
    __syncthreads();

    // ---- block-wide prefix sum.  Thread `tid` owns quad q = row * NT + tid of every row (striped, so the
    // ds_read_b128 is bank-conflict free); rows are resolved with one wave scan each plus one 64-entry scan.
    int4 v[ROWS];
    int ex[ROWS];
    {
        const int4 *t4 = reinterpret_cast<const int4 *>(tile);
#pragma unroll
        for (int rr = 0; rr < ROWS; rr++) {
            if (rr < rows_used) v[rr] = t4[rr * NT + tid]; else v[rr] = make_int4(0, 0, 0, 0);
            const int s3 = v[rr].x + v[rr].y + v[rr].z + v[rr].w;
            const int inc = wave_incl_scan(s3);
            ex[rr] = inc - s3;
            if (lane == 63) wtot[rr * NW + w] = inc;
        }
    }
    __syncthreads();
    int pre;  // exclusive prefix of (row, wave) totals in row-major order, one entry per lane
    {
        const int tot = wtot[lane];
        pre = wave_incl_scan(tot) - tot;
    }

    // ---- statistics
    const u64 excl = a.excl;
    const bool has_win = 2 * excl < (u64)L;
    const u32 ws = has_win ? (u32)excl : 0u, we = has_win ? (u32)(L - excl) : 0u;  // window [ws, we)
    const u32 wst = max(ws, lo), wet = min(we, lo + tlen_t);                       // window ∩ tile
    const bool win_any = has_win && wst < wet;
    const bool interior = has_win && lo >= ws && lo + TILE <= we;                   // implies tlen_t == TILE
    u64 sum_d = 0, sum_d2 = 0;
    u32 cov_w = 0, cov_f = 0, mn = 0xffffffffu, mx = 0;
    DevContig *C = &a.ctg[c];
    const u64 hoff = WANT_HIST ? C->hist_off : 0;
    const u32 hcap = WANT_HIST ? C->hist_cap : 0;
    auto hist_add = [&](u32 d, u32 x) {
        if (d < (u32)HB) atomicAdd(&lhist[d], x);
        else if (d <= hcap) atomicAdd(&a.hist_arena[hoff + d], x);
        else atomicOr(&a.g->internal_error, 1u);
    };
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++) {
        if (rr >= rows_used) break;
        const int base = __shfl(pre, rr * NW + w) + ex[rr];
        const u32 p0 = lo + 4u * (u32)(rr * NT + tid);
        const int dl[4] = {v[rr].x, v[rr].y, v[rr].z, v[rr].w};
        int d = base;
        if (interior) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int prev = d;
                d += dl[j];
                const u32 du = (u32)d;
                sum_d += du;
                sum_d2 += (u64)du * du;
                cov_w += d > 0;
                mn = min(mn, du); mx = max(mx, du);
                if (WANT_HIST && dl[j] != 0) {
                    const u32 rel = p0 + j - wst;
                    if (rel) { hist_add((u32)prev, rel); hist_add(du, 0u - rel); }
                }
                if (WRITE_DEPTH) a.depth_out[p0 + j] = d;
            }
            cov_f = cov_w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int prev = d;
                d += dl[j];
                const u32 p = p0 + j;
                const u32 du = (u32)d;
                if (p < L) {
                    cov_f += d > 0;
                    if (WRITE_DEPTH) a.depth_out[p] = d;
                    if (win_any && p >= wst && p < wet) {
                        sum_d += du;
                        sum_d2 += (u64)du * du;
                        cov_w += d > 0;
                        mn = min(mn, du); mx = max(mx, du);
                        if (WANT_HIST && dl[j] != 0) {
                            const u32 rel = p - wst;
                            if (rel) { hist_add((u32)prev, rel); hist_add(du, 0u - rel); }
                        }
                    }
                }
            }
        }
        // closing term of the change-point histogram: the last window position of this tile
        if (WANT_HIST && win_any) {
            const u32 last = wet - 1;
            if (last >= p0 && last < p0 + 4) {
                int dd = base;
                for (u32 j = 0; j <= last - p0; j++) dd += dl[j];
                hist_add((u32)dd, wet - wst);
            }
        }
    }

    // ---- workgroup reduction, then one set of global atomics per tile
    sum_d = wave_sum_u64(sum_d);
    sum_d2 = wave_sum_u64(sum_d2);
    cov_w = wave_sum_u32(cov_w);
    cov_f = wave_sum_u32(cov_f);
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    if (lane == 0) {
        red64[w * 2 + 0] = sum_d; red64[w * 2 + 1] = sum_d2;
        red32[w * 4 + 0] = cov_w; red32[w * 4 + 1] = cov_f; red32[w * 4 + 2] = mn; red32[w * 4 + 3] = mx;
    }
    __syncthreads();
    if (w == 0) {
        u64 a0 = lane < NW ? red64[lane * 2 + 0] : 0, a1 = lane < NW ? red64[lane * 2 + 1] : 0;
        u32 b0 = lane < NW ? red32[lane * 4 + 0] : 0, b1 = lane < NW ? red32[lane * 4 + 1] : 0;
        u32 b2 = lane < NW ? red32[lane * 4 + 2] : 0xffffffffu, b3 = lane < NW ? red32[lane * 4 + 3] : 0;
        a0 = wave_sum_u64(a0); a1 = wave_sum_u64(a1);
        b0 = wave_sum_u32(b0); b1 = wave_sum_u32(b1);
        b2 = wave_min_u32(b2); b3 = wave_max_u32(b3);
        if (lane == 0) {
            if (a0) atomicAdd(&C->sum_d, a0);
            if (a1) atomicAdd(&C->sum_d2, a1);
            if (b0) atomicAdd(&C->cov_win, (u64)b0);
            if (b1) atomicAdd(&C->cov_full, (u64)b1);
            if (win_any) {
                atomicAdd(&C->proc_win, (u64)(wet - wst));
                atomicMin(&C->min_d, b2);
                atomicMax(&C->max_d, b3);
            }
            red32[NW * 4 + 0] = b2; red32[NW * 4 + 1] = b3;
        }
    }
    if (WANT_HIST) {
        __syncthreads();
        if (win_any) {
            const u32 dmin = red32[NW * 4 + 0], dmax = min(red32[NW * 4 + 1], (u32)HB - 1);
            for (u32 b = dmin + tid; b <= dmax; b += NT) {
                const u32 x = lhist[b];
                if (x) atomicAdd(&a.hist_arena[hoff + b], x);
            }
        }
    }
}

constexpr size_t pileup_smem_bytes(int nt, bool hist) {
    return (size_t)TILE * 4 + 256 + (size_t)(nt / 64) * 32 + 16 + (hist ? (size_t)HB * 4 : 0);
}

}  // namespace covk
