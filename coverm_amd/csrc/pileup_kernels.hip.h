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

// Tile = bases handled by one workgroup, resident in LDS as i32 (TILE * 4 bytes).  Smaller tiles put more
// workgroups on a CU (160 KiB LDS), which is what hides the dependent global-load latency of the event phase.
constexpr u32 F_POS_UNSORTED = 1u;
// run-word types (top two bits of .y)
constexpr u32 RW_SINGLE = 0u, RW_DOUBLE = 1u, RW_COMPLEX = 2u, RW_BUCKET = 3u;
constexpr u32 FAST_MAX_CAND = 8191u;     // tiles with more candidate runs than this go to k_pileup_stream (k_pileup_fast's table holds 16-bit fields of 4 x the delta)
constexpr u32 TILE_F_GENERIC = 1u, TILE_F_SLOW = 2u;   // tile descriptor flags (desc[2t].w)
constexpr u32 CX_MIN_OPS = 16;   // RW_COMPLEX records with more CIGAR operations than this go through the per-tile buckets

// Per-contig device accumulators (128 B).
struct DevContig {
    u64 n_primary, n_pass, n_nonsupp, sum_nm, sum_indel;   // k_prep
    double id_primary, id_nonsupp;                          // k_identity
    u64 sum_d, sum_d2, cov_win, cov_full, proc_win;         // k_pileup
    u64 hist_off;                                           // k_hist_off<0>
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

constexpr u32 COUNTER_SLOTS = 64;   // device-wide counters are spread over this many 64-byte lines
struct DevGlobal {
    u64 first_error;     // min over erroring records of (record_index << 8 | code); ~0 = none
    u64 hist_cap_total;  // arena bins in use
    u64 chist_total;     // compact histogram bins
    u32 internal_error;  // depth exceeded its proven bound (would indicate a bug), etc.
    u32 n_cx;            // RW_BUCKET records appended to CxIdx::list
    u64 cx_total;        // (operation, tile) pairs they expand to = entries needed in CxIdx::runs
    u32 n_slow;          // tiles k_ranges left to k_pileup_stream (TILE_F_SLOW), listed in slow_list
    u32 n_gen;           // steps k_prep_lean left to k_prep_generic, listed in PrepCold::gen_list
    u64 pad2[2];
    u64 prim_slots[COUNTER_SLOTS * 8];  // sum = num_detected_primary_alignments (bam_generator.rs:114-118)
    u64 cons_slots[COUNTER_SLOTS * 8];  // sum = number of considered records
    u32 pad_q[8 * 16];                  // (k_pileup_stream's work-queue counters until round 6; kept so that the block's size does not change)
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
    u32 cigar_end;   // one past the largest index cigar_off can address (for clamped speculative loads)
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Inclusive wave64 prefix sum with DPP lane moves (VALU latency, no LDS crossbar round trips):
// row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2-3.
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}
// Inclusive wave64 running maximum (same DPP schedule as wave_incl_scan; lanes without a source keep INT_MIN).
__device__ __forceinline__ int wave_incl_max(int v) {
    const int NEG = (int)0x80000000;
    v = max(v, __builtin_amdgcn_update_dpp(NEG, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(NEG, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(NEG, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(NEG, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(NEG, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(NEG, v, 0x143, 0xc, 0xf, false));
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

// Wave totals by DPP prefix sums (VALU only; the __shfl_xor versions above go through the LDS crossbar, twelve round trips for a u64).  Every lane
// returns the total.  wave_sum_u56: the values must be below 2^56 (two 24-bit-apart halves, each summed in 32 bits).
__device__ __forceinline__ u32 wave_sum_u32_dpp(u32 v) { return (u32)__builtin_amdgcn_readlane(wave_incl_scan((int)v), 63); }
__device__ __forceinline__ u64 wave_sum_u56(u64 v) {
    const u32 lo = (u32)v & 0xffffffu, hi = (u32)(v >> 24);
    return (u64)wave_sum_u32_dpp(lo) + ((u64)wave_sum_u32_dpp(hi) << 24);
}
__device__ __forceinline__ u32 wave_max_u32_dpp(u32 v) {      // (wave_incl_max compares as signed: flip the top bit)
    return (u32)__builtin_amdgcn_readlane(wave_incl_max((int)(v ^ 0x80000000u)), 63) ^ 0x80000000u;
}
__device__ __forceinline__ u32 wave_min_u32_dpp(u32 v) { return ~wave_max_u32_dpp(~v); }

// broadcast of lane q's value through the scalar unit (v_readlane), far cheaper than a ds_bpermute shuffle
__device__ __forceinline__ double bcast_f64(double v, int q) {
    const long long b = __double_as_longlong(v);
    const u32 lo = __builtin_amdgcn_readlane((u32)b, q), hi = __builtin_amdgcn_readlane((u32)(b >> 32), q);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ u64 bcast_u64(u64 v, int q) {
    const u32 lo = __builtin_amdgcn_readlane((u32)v, q), hi = __builtin_amdgcn_readlane((u32)(v >> 32), q);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ void report_error(DevGlobal *g, u32 rec, u32 code) {
    atomicMin(&g->first_error, ((u64)rec << 8) | code);
}

// ------------------------------------------------------------------------------------ k_init
// Per-tile record bookkeeping filled by k_prep and turned into candidate ranges by k_tile_scan* + k_ranges:
//   tcnt[t]  records of the tile's contig whose (clamped) start position lies in tile t
//   fov[t]   smallest index of a record that starts LEFT of tile t and reaches into it (0xffffffff = none)
// For a position-sorted contig, F(X) = first record with pos >= X is rec_start + (prefix sum of tcnt), so the
// records that can touch tile t are [min(F(lo_t), fov[t]), F(lo_t) + tcnt[t]) -- no binary search, and one long
// read widens only the tiles it really spans.
struct TileIdx {
    const u32 *tile_first;   // n_targets + 1: first tile of each contig
    u32 *tcnt, *fov;
    u32 shift;               // log2(tile width)
    u32 n_tiles;
};

// Per-tile buckets of expanded M/=/X operations of RW_BUCKET records (k_cx_expand): cnt[t] entries at
// runs[off(t) ..), off = exclusive prefix sum of cnt (cscan + ctop, two-level like tcnt).
struct CxIdx {
    u32 *list; u32 list_cap;     // record indices of RW_BUCKET records, any order
    u32 *cnt, *cur;              // per tile: entries, fill cursor
    const u32 *cscan, *ctop;     // exclusive prefix sums of cnt (valid after k_tile_scan1/2)
    uint2 *runs; u64 runs_cap;   // (start, end) in contig coordinates
};

// Arguments of k_prep_lean (prep_lean.hip.h).  PrepHot is what its loop reads every step (passed by value: scalar registers); PrepCold what
// only rare branches read.  k_init, which runs in front of every k_prep, writes both into one block in device memory (PrepArgs): the
// out-of-line generic step and the rare branches fetch their arguments from there instead of holding registers for them.
struct PrepPartial;
struct PrepCold {
    DevContig *ctg; DevGlobal *g; PrepPartial *part; u32 *cx_list;
    const uint8_t *mask; const u32 *tlen; const u32 *tile_first;
    u32 *gen_list;        // first record of every step k_prep_lean leaves to k_prep_generic (capacity: one entry per 64 records)
    u32 cx_list_cap, pad;
};
struct PrepHot {
    const int32_t *tid, *pos;
    const uint16_t *flag;
    const uint8_t *nmk;
    const u32 *nm, *coff, *cigar;
    const uint8_t *mapq;
    const u32 *lseq;
    uint2 *runs;
    u32 *tcnt, *fov;
    double *identp, *identn;
    u32 n, cigar_end, n_targets, shift, steps;
    u32 gate_mask;        // scan-stage gate: a record passes iff ((flag ^ 2) & gate_mask) == 0  (FlagFilter::passes, lib.rs:67-78, + !unmapped, contig.rs:125)
    u32 p1_mask;          // reader stage, filter.rs:100-102: passes_filter1 iff (flag & p1_mask) == 0
    u32 min_mapq, min_aligned_length;
    float min_percent_identity, min_aligned_percent;
};
struct PrepArgs { PrepHot hot; PrepCold cold; };

__global__ void k_init(DevContig *ctg, u32 n_targets, DevGlobal *g, TileIdx ti, CxIdx cx, PrepArgs *pa_out, PrepArgs pa) {
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && pa_out != nullptr) *pa_out = pa;
    if (c < ti.n_tiles) { ti.tcnt[c] = 0u; ti.fov[c] = 0xffffffffu; cx.cnt[c] = 0u; cx.cur[c] = 0u; }
    if (c == 0) {
        g->first_error = ~0ull; g->hist_cap_total = 0; g->chist_total = 0; g->internal_error = 0;
        g->n_cx = 0; g->cx_total = 0; g->n_slow = 0; g->n_gen = 0;
    }
    if (c < COUNTER_SLOTS * 8) { g->prim_slots[c] = 0; g->cons_slots[c] = 0; }
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
// Run word (8 bytes per record, all k_pileup needs for almost every read): x = start of the first merged
// M/=/X run (contig coordinate); y = 0 for "no events", else top two bits give the type:
//   RW_SINGLE  y = length of the only run
//   RW_DOUBLE  two runs split by one short D/N gap: len1 (10 bits) | gap (8 bits) << 10 | len2 (10 bits) << 18
//   RW_COMPLEX anything else with a short CIGAR (several gaps): k_pileup re-walks the CIGAR
//   RW_BUCKET  long CIGARs (long-read mappings): k_cx_expand writes every M/=/X operation into the buckets of the tiles
//              it overlaps; k_pileup reads its tile's bucket (coalesced) and ignores the record itself
// Adjacent runs separated only by I/S/H/P are merged: their +1/-1 events cancel (contig.rs:178-183).
//
// One workgroup walks PREP_CHUNK consecutive records (PREP_ITEMS coalesced passes).  Each wave keeps
// per-lane running sums for the contig it is currently in and flushes them with one set of atomics
// only when the contig changes: a single hot address costs ~12 ns per atomic on MI355X, so per-wave
// (let alone per-record) atomics on the per-contig or global counters would dominate the kernel.
constexpr int PREP_B = 2;        // records per thread per pass: their loads are issued together
constexpr int PREP_PASSES = 8;
constexpr int PREP_CHUNK = 256 * PREP_B * PREP_PASSES;

// Literal 64-bit CIGAR walk (the path every record took before the branch-free fast walk existed); k_prep keeps it
// for records the fast walk cannot take: more than CIG_FAST_OPS operations or an operation of >= 2^24 bases.
constexpr u32 CIG_FAST_OPS = 128;
__device__ __forceinline__ void cigar_walk_slow(const u32 *__restrict__ cigar, u32 co0, u32 nops, int pos, u32 L, u64 &aligned,
                                             u64 &indel, u32 &run_start, u32 &run_len, u32 &run2_start, u32 &run2_len,
                                             u32 &n_runs, u32 &span, bool &oob, bool &badcig) {
    aligned = 0; indel = 0; run_start = run_len = run2_start = run2_len = n_runs = span = 0; oob = false; badcig = false;
    long long cursor = pos;
    long long cur_e = -1;  // end of the open merged run
    for (u32 c = 0; c < nops; c++) {
        const u32 wd = cigar[co0 + c];
        const u32 op = wd & 15u, len = wd >> 4;
        if (op == 0u || op == 7u || op == 8u) {          // M = X  (contig.rs:171-186)
            if (cursor < 0 || cursor >= (long long)L) oob = true;
            if (n_runs > 0 && cursor == cur_e) {
                cur_e += len;
                if (n_runs == 1) run_len += len; else if (n_runs == 2) run2_len += len;
            } else {
                n_runs++;
                cur_e = cursor + len;
                if (n_runs == 1) { run_start = (u32)cursor; run_len = len; }
                else if (n_runs == 2) { run2_start = (u32)cursor; run2_len = len; }
            }
            cursor += len; aligned += len;
        } else if (op == 2u) { cursor += len; indel += len; aligned += len; }  // D  (:187-191)
        else if (op == 3u) { cursor += len; }                                  // N  (:192-195)
        else if (op == 1u) { indel += len; aligned += len; }                   // I  (:196-199)
        else if (op > 8u) badcig = true;                                       // S H P ignored (:200)
    }
    const long long sp = cursor - (long long)pos;
    span = sp > 0 ? (u32)min(sp, (long long)0xffffffffu) : 0u;
}

// Whole-wave walk of ONE long CIGAR (all arguments wave-uniform; every lane returns the same result): 64 operations
// per step, loaded coalesced; reference positions come from a wave prefix sum of the reference-consuming lengths and
// run boundaries from a running maximum of the M/=/X ends (an M op continues the open run iff it starts where the
// previous M op ended, contig.rs:178-183).  A long-read record costs nops/64 steps instead of nops serial iterations
// of one lane.  `absurd` (an operation of >= 2^24 bases) sends the record to the literal serial walk.
struct WalkOut {
    u64 aligned, indel;
    u32 run_start, run_len, run2_start, run2_len, n_runs, span;
    bool oob, badcig, absurd;
};
__device__ __forceinline__ void cigar_walk_wave(const u32 *__restrict__ cigar, u32 co0, u32 nops, int pos, u32 L, WalkOut &o) {
    const int lane = lane_id();
    const int NEG = (int)0x80000000;
    long long cur0 = pos, last_e = 0, rs1 = 0, rs2 = 0, r1end = 0;
    bool have_prev = false, oobl = false, badl = false;
    u64 al = 0, in_ = 0;
    u32 n_runs = 0;
    o.absurd = false;
    u32 wd_next = (u32)lane < nops ? cigar[co0 + (u32)lane] : 4u;  // padding = 0S: no flag set
    for (u32 base = 0; base < nops; base += 64) {
        const u32 wd = wd_next;
        const u32 idn = base + 64u + (u32)lane;                    // next step's words are in flight during this one
        wd_next = idn < nops ? cigar[co0 + idn] : 4u;
        const u32 len = wd >> 4, bit = 1u << (wd & 15u);
        if (__any(len >= (1u << 24))) { o.absurd = true; return; }
        const u32 fr = (bit & 0x18du) ? len : 0u;
        const u32 incl = (u32)wave_incl_scan((int)fr), excl = incl - fr;
        const bool m = (bit & 0x181u) != 0u;
        badl |= (bit & 0xfe00u) != 0u;
        al += (bit & 0x187u) ? len : 0u;
        in_ += (bit & 0x006u) ? len : 0u;
        const long long sj = cur0 + (long long)excl;
        oobl |= m && (sj < 0 || sj >= (long long)L);
        const int erel = m ? (int)(excl + len) : NEG;
        int pm = wave_incl_max(erel);
        pm = __builtin_amdgcn_update_dpp(NEG, pm, 0x138, 0xf, 0xf, false);      // exclusive: wave_shr:1
        const int carry_rel = have_prev ? (int)(last_e - cur0) : NEG;            // <= 0
        pm = max(pm, carry_rel);
        const bool nr = m && (pm == NEG || (int)excl != pm);
        u64 mm = __ballot(nr);
        while (mm != 0 && n_runs < 2u) {
            const int l = __ffsll((long long)mm) - 1;
            const long long sl = cur0 + (long long)__builtin_amdgcn_readlane(excl, l);
            if (n_runs == 0u) rs1 = sl;
            else { rs2 = sl; r1end = cur0 + (long long)(int)__builtin_amdgcn_readlane((u32)pm, l); }
            n_runs++;
            mm &= mm - 1;
        }
        n_runs += (u32)__popcll(mm);
        const u64 mb = __ballot(m);
        if (mb != 0) {
            const int lm = 63 - __builtin_clzll(mb);
            last_e = cur0 + (long long)__builtin_amdgcn_readlane(excl + len, lm);
            have_prev = true;
        }
        cur0 += (long long)__builtin_amdgcn_readlane(incl, 63);
    }
    o.aligned = wave_sum_u64(al); o.indel = wave_sum_u64(in_);
    o.oob = __any(oobl); o.badcig = __any(badl);
    const long long sp = cur0 - (long long)pos;
    o.span = sp > 0 ? (u32)min(sp, (long long)0xffffffffu) : 0u;
    o.n_runs = n_runs;
    o.run_start = (u32)rs1; o.run2_start = (u32)rs2;
    o.run_len = n_runs == 1u ? (u32)(last_e - rs1) : (u32)(r1end - rs1);
    o.run2_len = n_runs >= 2u ? (u32)(last_e - rs2) : 0u;      // only meaningful when n_runs == 2
}

// Counters of one workgroup that lies entirely inside one contig (tid >= 0), reduced later by k_prep_reduce:
// per-wave atomics on a hot contig's accumulator line serialise at ~12 ns each and dominated k_prep.
struct PrepPartial {
    int tid; u32 prim, pass, nons, span, first, last, pad;
    u64 nm, indel;
};

// (prim / pass / nons are WAVE totals — sums of ballot popcounts, so they live in scalar registers —, the others per lane)
struct PrepAcc {
    u32 prim, pass, nons, span, first, last;
    u64 nm, indel;
    __device__ __forceinline__ void reset() { prim = pass = nons = span = 0; first = 0xffffffffu; last = 0; nm = indel = 0; }
};

__device__ __forceinline__ void prep_flush(DevContig *ctg, int cur, PrepAcc &a) {
    if (cur >= 0) {
        const u32 prim = a.prim, pass = a.pass, nons = a.nons;
        const u32 span = wave_max_u32(a.span), first = wave_min_u32(a.first), last = wave_max_u32(a.last);
        const u64 nm = wave_sum_u64(a.nm), indel = wave_sum_u64(a.indel);
        if (lane_id() == 0 && pass) {
            DevContig *C = &ctg[cur];
            if (prim) atomicAdd(&C->n_primary, (u64)prim);
            atomicAdd(&C->n_pass, (u64)pass);
            if (nons) atomicAdd(&C->n_nonsupp, (u64)nons);
            if (nm) atomicAdd(&C->sum_nm, nm);
            if (indel) atomicAdd(&C->sum_indel, indel);
            if (span) atomicMax(&C->max_span, span);
            atomicMin(&C->first_rec, first);
            atomicMax(&C->last_rec, last);
        }
    }
    a.reset();
}

// FILTER: the reader-stage single-read filter is on (filter.rs:88-116); MASKED: a target mask is set (genome.rs:170-171).
// Both are compile-time so that the common `coverm contig` shape carries neither the loads (mapq, l_seq, mask) nor the code.
// (k_prep2 — a lane taking two ADJACENT records: 8-byte loads of tid / pos / nm / cigar_off, one dword of flags, neighbours in file order by
// DPP, 12-byte CIGAR loads, one 16-byte store of both run words: 17 memory instructions per pair instead of 36 — was built and measured in
// round 5, alternating with this kernel on one box: 0.76-0.78 ms against 0.61, profiles/r05_prep_pileup_ab.log.  Its counters say why: 43 %
// fewer memory instructions and 6 % fewer VALU instructions, but the SIMDs are busy ISSUING — 29 % of a wave's time executing at 4.3 waves per
// SIMD — not waiting for bytes, and the 64-bit address arithmetic of the wide loads plus the wave-wide DPP moves cost more issue slots than
// the narrow loads they replaced.  Removed.)
template <bool WANT_IDENTITY, bool FILTER, bool MASKED, bool PREFETCH, int PB>
__device__ __forceinline__ void prep_body(const Records &r, const u32 *__restrict__ tlen, u32 n_targets,
                                          const uint8_t *__restrict__ mask, const FilterCfg &f, DevContig *ctg,
                                          DevGlobal *g, uint2 *__restrict__ runs, double *__restrict__ identp,
                                          double *__restrict__ identn, PrepPartial *__restrict__ part, const TileIdx &ti,
                                          int passes, int b_active, u32 *__restrict__ cx_list, u32 cx_list_cap) {
    __shared__ u32 blk_cnt[2][4];
    __shared__ PrepPartial wpart[4];
    bool flushed_early = false;   // this wave already sent sums for an earlier contig through atomics
    const int lane = lane_id(), w = threadIdx.x >> 6;
    // records per workgroup: the host picks one pass of one record per thread for long CIGARs (few records, each a lot
    // of work: more, smaller workgroups), 8 passes of PB records for short reads
    const u32 chunk_recs = 256u * (u32)b_active * (u32)passes;
    const u32 chunk = blockIdx.x * chunk_recs;
    PrepAcc acc; acc.reset();
    int cur = -1;            // wave-uniform: contig the running sums belong to
    u32 g_prim = 0, g_cons = 0;
    const u32 nlast = r.n - 1u;                                  // r.n > 0 (host never launches on an empty store)
    const u32 cig_last = r.cigar_end ? r.cigar_end - 1u : 0u;

    // Chunk-relative addressing: uniform (SGPR) bases + a small per-thread offset, so every load is
    // `global_load v, voff, s[base]` instead of a 64-bit per-lane address computation.
    const u32 lmax = min(r.n - chunk, chunk_recs) - 1u;     // last valid chunk-relative index (r.n > chunk)
    const uint16_t *flag_c = r.flag + chunk; const int *tid_c = r.tid + chunk; const int *pos_c = r.pos + chunk;
    const uint8_t *mapq_c = r.mapq + chunk; const uint8_t *nmk_c = r.nm_kind + chunk; const u32 *nm_c = r.nm + chunk;
    const u32 *lseq_c = r.l_seq + chunk; const u32 *coff_c = r.cigar_off + chunk;
    const int *tid_m = r.tid + chunk - 1; const int *pos_m = r.pos + chunk - 1;   // [x + 1] = record x (never read below 0)
    uint2 *runs_c = runs + chunk;

    // PREFETCH: the two fields the dependent loads of a pass hang on (tid -> contig length, first tile, mask; cigar_off -> the CIGAR words)
    // are loaded ONE PASS AHEAD, behind the current pass's loads, so that a pass has one load phase instead of two dependent ones; two
    // registers per record stay live across the per-record logic for it (4-5 % at equal waves per SIMD).  (The same through LDS —
    // global_load_lds_dword, no register held — measured slower at five and at six waves: the compiler drains every load before the first
    // use of an ordinary one while an LDS-bound load is in flight, and the pass begins with a wait for the previous pass's stores.  EVERY
    // independent field one pass ahead, 20-24 more registers, lost at four and at five waves.  profiles/r05_prep_prefetch_ab.log.)
    // PB: records per thread and pass, their loads issued together.
    struct PhaseA { u32 fl, mq, nmk, nmv32, lsq, co0, co1; int td, ps_, ptid, ppos, ntid; };
    auto load_a = [&](u32 l0x, int k, bool roots, bool rest, PhaseA &A) {   // clamped to the chunk's last record: always in bounds
        const u32 lc = min(l0x + (u32)k * 256u, lmax) & (u32)(PREP_CHUNK - 1);   // the mask is a no-op that bounds lc for the compiler
        const u32 ic = chunk + lc;
        if (roots) { A.td = tid_c[lc]; A.co0 = coff_c[lc]; }
        if (rest) {
            A.fl = flag_c[lc]; A.ps_ = pos_c[lc]; A.mq = FILTER ? mapq_c[lc] : 0u; A.nmk = nmk_c[lc];
            A.nmv32 = nm_c[lc]; A.lsq = FILTER ? lseq_c[lc] : 0u; A.co1 = coff_c[lc + 1u];
            const u32 lp = (lc + (ic > 0u ? 0u : 1u)) & (u32)(2 * PREP_CHUNK - 1), ln = (lc + (ic < nlast ? 1u : 0u)) & (u32)(2 * PREP_CHUNK - 1);
            A.ptid = tid_m[lp]; A.ppos = pos_m[lp]; A.ntid = tid_c[ln];
        }
    };
    PhaseA nx[PB];
    if (PREFETCH) {
#pragma unroll
        for (int k = 0; k < PB; k++) load_a(threadIdx.x, k, true, false, nx[k]);
    }

    for (int ps = 0; ps < passes; ps++) {
        const u32 l0 = (u32)(ps * b_active) * 256u + threadIdx.x;
        const u32 i0 = chunk + l0;
        if (!__any(i0 < r.n)) break;
        // ---- phase A: every independent field of PB records, issued back to back (clamped, branch-free)
        u32 fl[PB], mq[PB], nmk[PB], nmv32[PB], lsq[PB], co0[PB], co1[PB];
        int td[PB], ps_[PB], ptid[PB], ppos[PB], ntid[PB];
#pragma unroll
        for (int k = 0; k < PB; k++) {
            PhaseA A = nx[k];
            load_a(l0, k, !PREFETCH, true, A);
            fl[k] = A.fl; td[k] = A.td; ps_[k] = A.ps_; mq[k] = A.mq; nmk[k] = A.nmk; nmv32[k] = A.nmv32; lsq[k] = A.lsq; co0[k] = A.co0; co1[k] = A.co1;
            ptid[k] = A.ptid; ppos[k] = A.ppos; ntid[k] = A.ntid;
        }
        // ---- phase B: loads that depend on phase A (first three CIGAR words, contig length, mask)
        u32 cw[PB][3], Lc[PB], mk[PB], t0[PB];
#pragma unroll
        for (int k = 0; k < PB; k++) {
            const bool tok = td[k] >= 0 && (u32)td[k] < n_targets;
            Lc[k] = tok ? tlen[td[k]] : 0u;
            t0[k] = tok ? ti.tile_first[td[k]] : 0u;
            mk[k] = (MASKED && tok) ? mask[td[k]] : 1u;
#pragma unroll
            for (int c = 0; c < 3; c++) cw[k][c] = r.cigar_end ? r.cigar[min(co0[k] + (u32)c, cig_last)] : 0u;
        }
        if (PREFETCH) {
#pragma unroll
            for (int k = 0; k < PB; k++) load_a(l0 + (u32)b_active * 256u, k, true, false, nx[k]);
        }
        // ---- phase C: per-record logic
#pragma unroll
        for (int k = 0; k < PB; k++) {
            const u32 i = i0 + (u32)k * 256u;
            const bool in = i < r.n && k < b_active;
            const u32 flag = in ? fl[k] : 0x904u;
            const int tid = in ? td[k] : -1;
            const int pos = ps_[k];
            // every record read counts towards num_detected_primary_alignments when !secondary && !supplementary
            g_prim += (u32)__popcll(__ballot(in && !(flag & 0x900u)));      // (wave totals: scalar registers)
            const bool tid_ok = in && tid >= 0 && (u32)tid < n_targets;
            // span of records carrying this tid + grouping / position-order checks (all records, considered or not)
            if (tid_ok) {
                const int pt = i > 0 ? ptid[k] : -2;
                if (pt != tid) {
                    atomicMin(&ctg[tid].rec_start, i);
                    atomicAdd(&ctg[tid].n_groups, 1u);
                } else if (ppos[k] > pos) {
                    atomicOr(&ctg[tid].flags, F_POS_UNSORTED);
                }
                const int nt = (i + 1 < r.n) ? ntid[k] : -2;
                if (nt != tid) atomicMax(&ctg[tid].rec_end, i + 1);
            }
            const bool unmapped = flag & 0x4u;
            const bool supp = flag & 0x800u, sec = flag & 0x100u;
            // reader stage: ReferenceSortedBamFilter::read single-read branch, filter_out = true (filter.rs:88-116)
            bool survives = in;
            bool need_filter_eval = false;
            if (FILTER) {
                survives = false;
                const bool p1 = in && !unmapped && (f.include_supplementary || !supp) && (f.include_secondary || !sec);
                if (p1 && !(f.min_mapq != 255u && (mq[k] < f.min_mapq || mq[k] == 255u))) need_filter_eval = true;  // :250-254
            }
            // scan stage gate: FlagFilter::passes (lib.rs:67-78) then !unmapped (contig.rs:125)
            const bool flags_ok = !(!f.include_secondary && sec) && !(!f.include_supplementary && supp) &&
                                  !(!f.include_improper_pairs && !(flag & 0x2u));
            const bool scan_gate = in && flags_ok && !unmapped;

            u64 aligned = 0, indel = 0;
            u32 run_start = 0, run_len = 0, run2_start = 0, run2_len = 0, n_runs = 0, span = 0;
            bool oob = false, badcig = false;
            // with the reader-stage filter on, only records that reach single_read_passes_filter can survive
            const bool do_walk = FILTER ? need_filter_eval : scan_gate;
            const u32 nops_all = do_walk ? co1[k] - co0[k] : 0u;
            // Fast walk: branch-free, 32-bit, trip count uniform over the wave.  Valid while nothing can overflow
            // 32 bits: at most CIG_FAST_OPS ops of < 2^24 each (sum < 2^31); anything else (long reads, absurd
            // lengths) is redone by the literal 64-bit walk below.
            bool hard = nops_all > CIG_FAST_OPS;
            bool big = false;   // an operation of >= 2^24 bases: such a record never takes the bucket path
            if (__all(nops_all <= 3u)) {
                // Every record of this wave step has at most three operations (all but a few per mille of a short-read sample: kM, kM iI jM,
                // kM dD jM, sS kM): the state machine below in closed form — a third fewer instructions than three of its steps.  With
                // c_j the cursor in front of operation j and an absent operation read as 0S:  consecutive M/=/X operations always merge
                // (an M consumes what it aligns), so the only second run is M, <D or N of positive length>, M; everything else is one run
                // that starts at the first M/=/X operation.
                const u32 w0 = 0u < nops_all ? cw[k][0] : 4u, w1 = 1u < nops_all ? cw[k][1] : 4u, w2 = 2u < nops_all ? cw[k][2] : 4u;
                const u32 len0 = w0 >> 4, len1 = w1 >> 4, len2 = w2 >> 4;
                const u32 bit0 = 1u << (w0 & 15u), bit1 = 1u << (w1 & 15u), bit2 = 1u << (w2 & 15u);
                const bool m0 = (bit0 & 0x181u) != 0u, m1 = (bit1 & 0x181u) != 0u, m2 = (bit2 & 0x181u) != 0u;       // M = X   (contig.rs:171-186)
                const u32 ref0 = (bit0 & 0x18du) ? len0 : 0u, ref1 = (bit1 & 0x18du) ? len1 : 0u, ref2 = (bit2 & 0x18du) ? len2 : 0u;
                const u32 c0 = (u32)pos, c1 = c0 + ref0, c2 = c1 + ref1;
                const u32 L = Lc[k];
                span = ref0 + ref1 + ref2;
                aligned = ((bit0 & 0x187u) ? len0 : 0u) + ((bit1 & 0x187u) ? len1 : 0u) + ((bit2 & 0x187u) ? len2 : 0u);     // M I D = X   (:187-199)
                indel = ((bit0 & 0x006u) ? len0 : 0u) + ((bit1 & 0x006u) ? len1 : 0u) + ((bit2 & 0x006u) ? len2 : 0u);       // I D
                oob = (m0 && c0 >= L) || (m1 && c1 >= L) || (m2 && c2 >= L);       // negative cursors wrap to >= 2^31 > L
                badcig = ((bit0 | bit1 | bit2) & 0xfe00u) != 0u;
                big = ((w0 | w1 | w2) >> 28) != 0u;                                // a length of >= 2^24
                const bool second = m0 && !m1 && m2 && ref1 != 0u;
                n_runs = ((m0 || m1 || m2) ? 1u : 0u) + (second ? 1u : 0u);
                run_start = m0 ? c0 : (m1 ? c1 : c2);
                run_len = (m0 ? len0 : 0u) + (m1 ? len1 : 0u) + ((m2 && !second) ? len2 : 0u);
                run2_start = second ? c2 : 0u; run2_len = second ? len2 : 0u;
                hard = big;
            } else {
                const u32 nops = hard ? 0u : nops_all;
                const u32 L = Lc[k];
                u32 cursor = (u32)pos, cur_e = 0, al32 = 0, in32 = 0;
                // (an operation the record does not have comes in as 0S — no flag set, no length — so no step tests whether it is active)
                auto step = [&](u32 wd) {
                    const u32 len = wd >> 4, bit = 1u << (wd & 15u);
                    big |= len >= (1u << 24);
                    const bool m = (bit & 0x181u) != 0u;                       // M = X   (contig.rs:171-186)
                    badcig |= (bit & 0xfe00u) != 0u;
                    oob |= m && cursor >= L;                                   // negative cursors wrap to >= 2^31 > L
                    const bool ext = m && n_runs > 0u && cursor == cur_e;      // continues the open merged run
                    n_runs += (m && !ext) ? 1u : 0u;
                    const bool r1 = m && n_runs == 1u, r2 = m && n_runs == 2u;
                    run_start = (r1 && !ext) ? cursor : run_start;
                    run2_start = (r2 && !ext) ? cursor : run2_start;
                    run_len = r1 ? (ext ? run_len : 0u) + len : run_len;
                    run2_len = r2 ? (ext ? run2_len : 0u) + len : run2_len;
                    cur_e = m ? cursor + len : cur_e;
                    cursor += (bit & 0x18du) ? len : 0u;                       // M D N = X consume the reference
                    al32 += (bit & 0x187u) ? len : 0u;                         // M I D = X   (:187-199)
                    in32 += (bit & 0x006u) ? len : 0u;                         // I D
                };
                step(0u < nops ? cw[k][0] : 4u);
                step(1u < nops ? cw[k][1] : 4u);
                step(2u < nops ? cw[k][2] : 4u);
                for (u32 c = 3; __any(c < nops); c++) step(c < nops ? r.cigar[co0[k] + c] : 4u);
                aligned = al32; indel = in32;
                span = cursor - (u32)pos;
                hard |= big;
            }
            for (u64 hm = __ballot(hard); hm != 0; hm &= hm - 1) {      // long CIGARs: one at a time, whole wave on each
                const int l = __ffsll((long long)hm) - 1;
                WalkOut o;
                cigar_walk_wave(r.cigar, __builtin_amdgcn_readlane(co0[k], l), __builtin_amdgcn_readlane(nops_all, l),
                                (int)__builtin_amdgcn_readlane((u32)pos, l), __builtin_amdgcn_readlane(Lc[k], l), o);
                if (lane == l) {
                    big |= o.absurd;
                    if (o.absurd) cigar_walk_slow(r.cigar, co0[k], nops_all, pos, Lc[k], aligned, indel, run_start, run_len,
                                                  run2_start, run2_len, n_runs, span, oob, badcig);
                    else {
                        aligned = o.aligned; indel = o.indel; run_start = o.run_start; run_len = o.run_len;
                        run2_start = o.run2_start; run2_len = o.run2_len; n_runs = o.n_runs; span = o.span;
                        oob = o.oob; badcig = o.badcig;
                    }
                }
            }
            if (FILTER && need_filter_eval) {  // single_read_passes_filter, filter.rs:256-278
                if (nmk[k] != 1u) report_error(g, i, nmk[k] == 0u ? 2u : 3u);
                else {
                    const u32 al = (u32)aligned;  // u32 accumulation in the reference
                    const float a = (float)al;
                    survives = al >= f.min_aligned_length && a / (float)lsq[k] >= f.min_aligned_percent &&
                               1.0f - (float)nmv32[k] / a >= f.min_percent_identity;
                }
            }
            const bool considered = survives && scan_gate;
            const bool masked_in = considered && tid_ok && mk[k];
            u64 nmv = 0;
            double idv = 0.0;
            if (considered && !tid_ok) report_error(g, i, 7u);  // header.target_len(tid).expect("Corrupt BAM file?")
            if (masked_in) {
                if (badcig) report_error(g, i, 6u);
                else if (oob) report_error(g, i, 4u);
                if (nmk[k] != 1u) report_error(g, i, nmk[k] == 0u ? 2u : 3u);  // nm(&record), contig.rs:206
                else nmv = nmv32[k];
                if (WANT_IDENTITY && aligned > 0) idv = ((double)aligned - (double)nmv) / (double)aligned;
            }
            bool is_bucket = false;
            if (in) {
                uint2 rw;
                rw.x = 0u; rw.y = 0u;
                if (masked_in && n_runs > 0) {
                    rw.x = run_start;
                    const u32 gap = run2_start - (run_start + run_len);
                    if (n_runs == 1 && run_len < (1u << 30)) rw.y = run_len;   // RW_SINGLE; 0 = nothing to add
                    else if (n_runs == 2 && run_len - 1u < 1023u && run2_len - 1u < 1023u && gap - 1u < 255u)
                        rw.y = (RW_DOUBLE << 30) | run_len | (gap << 10) | (run2_len << 18);
                    else {
                        is_bucket = nops_all > CX_MIN_OPS && !big;
                        rw.y = (is_bucket ? RW_BUCKET : RW_COMPLEX) << 30;
                    }
                }
                runs_c[i - chunk] = rw;
                if (WANT_IDENTITY) {   // a NULL stream is one the caller does not need (COV_WANT_IDENTITY_*_ONLY)
                    if (identn != nullptr) identn[i] = (masked_in && !supp) ? idv : 0.0;
                    if (identp != nullptr) identp[i] = (masked_in && !supp && !sec) ? idv : 0.0;
                }
            }

            {   // RW_BUCKET records: one aggregated append per wave
                const u64 qm = __ballot(is_bucket);
                if (qm != 0) {
                    const int l = __ffsll((long long)qm) - 1;
                    u32 b = 0;
                    if (lane == l) b = atomicAdd(&g->n_cx, (u32)__popcll(qm));
                    b = __builtin_amdgcn_readlane(b, l);
                    const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(qm >> 32), __builtin_amdgcn_mbcnt_lo((u32)qm, 0u));
                    if (is_bucket && b + rank < cx_list_cap) cx_list[b + rank] = i;
                }
            }
            // ---- tile bookkeeping for k_ranges (every record of a real contig counts towards F, considered or not)
            {
                const u32 L = Lc[k];
                const bool tv = tid_ok && L > 0u;
                const u32 pc = pos < 0 ? 0u : min((u32)pos, L - 1u);
                const u32 tl = pc >> ti.shift;
                const u32 key = tv ? t0[k] + tl : 0xffffffffu;
                const u32 pk = (u32)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)key, 0x138, 0xf, 0xf, false);   // wave_shr:1
                const bool lead = tv && (lane == 0 || key != pk);
                const u64 bm = __ballot(lead) | ~__ballot(tv);              // lanes where a run of equal keys ends
                if (lead) {
                    const u64 rest = (bm >> lane) >> 1;
                    const u32 follow = rest ? (u32)__builtin_ctzll(rest) : (u32)(63 - lane);
                    atomicAdd(&ti.tcnt[key], 1u + follow);
                }
                // records that reach beyond their own tile announce themselves to the tiles they enter
                const u32 e = span > L - pc ? L : pc + span;                 // end of the record's reference extent, clipped
                const u32 te = (tv && span > 0u) ? (e - 1u) >> ti.shift : 0u;
                const bool cross = tv && masked_in && n_runs > 0u && span > 0u && te > tl && !is_bucket;   // buckets deliver those
                const u64 cm = __ballot(cross);
                if (cm != 0) {
                    const int fl0 = __ffsll((long long)cm) - 1;
                    const u32 tgt = key + 1u, tgt0 = __builtin_amdgcn_readlane(tgt, fl0);
                    // usual case: all of them enter the same single tile, and the lowest lane has the smallest index
                    if (__all(!cross || (te == tl + 1u && tgt == tgt0))) { if (lane == fl0) atomicMin(&ti.fov[tgt0], i); }
                    else if (cross) for (u32 t2 = tl + 1u; t2 <= te; t2++) atomicMin(&ti.fov[t0[k] + t2], i);
                }
            }

            // per-contig counters: accumulate per lane while the wave stays inside one contig
            const bool cnt = considered && tid_ok;
            const u64 m = __ballot(cnt);
            g_cons += (u32)__popcll(m);
            if (m != 0) {
                const int ftid = __shfl(tid, __ffsll((long long)m) - 1);
                const bool uni = __all(!cnt || tid == ftid);
                if (uni) {
                    if (ftid != cur) { if (cur >= 0) flushed_early = true; prep_flush(ctg, cur, acc); cur = ftid; }
                    acc.prim += (u32)__popcll(__ballot(cnt && !supp && !sec));
                    acc.pass += (u32)__popcll(m);
                    acc.nons += (u32)__popcll(__ballot(cnt && !supp));
                    if (cnt) {
                        acc.first = min(acc.first, i); acc.last = max(acc.last, i);
                        if (masked_in) { acc.nm += nmv; acc.indel += indel; acc.span = max(acc.span, span); }
                    }
                } else {  // contig boundary inside this wave pass: rare, resolve with per-lane atomics
                    prep_flush(ctg, cur, acc); cur = -1; flushed_early = true;
                    if (cnt) {
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
                    }
                }
            }
        }
    }
    // End of chunk: if all four waves stayed inside the same single contig, publish ONE partial record for the
    // workgroup (reduced by k_prep_reduce); otherwise fall back to atomics.
    {
        PrepPartial pw;
        pw.tid = flushed_early ? -2 : cur;
        pw.prim = acc.prim; pw.pass = acc.pass; pw.nons = acc.nons;
        pw.span = wave_max_u32(acc.span); pw.first = wave_min_u32(acc.first); pw.last = wave_max_u32(acc.last);
        pw.nm = wave_sum_u64(acc.nm); pw.indel = wave_sum_u64(acc.indel); pw.pad = 0;
        if (lane == 0) wpart[w] = pw;
    }
    if (lane == 0) { blk_cnt[0][w] = g_prim; blk_cnt[1][w] = g_cons; }
    __syncthreads();
    bool uniform_wg = true;
    {
        int t0 = -1;
        for (int k = 0; k < 4; k++) {
            const int t = wpart[k].tid;
            if (t == -2) uniform_wg = false;
            else if (t >= 0) { if (t0 < 0) t0 = t; else if (t != t0) uniform_wg = false; }
        }
        if (threadIdx.x == 0) {
            PrepPartial o; o.tid = -1; o.prim = o.pass = o.nons = o.span = 0; o.first = 0xffffffffu; o.last = 0; o.pad = 0; o.nm = o.indel = 0;
            if (uniform_wg && t0 >= 0) {
                o.tid = t0;
                for (int k = 0; k < 4; k++) {
                    const PrepPartial &q = wpart[k];
                    if (q.tid < 0) continue;
                    o.prim += q.prim; o.pass += q.pass; o.nons += q.nons; o.span = max(o.span, q.span);
                    o.first = min(o.first, q.first); o.last = max(o.last, q.last); o.nm += q.nm; o.indel += q.indel;
                }
            }
            part[blockIdx.x] = o;
        }
    }
    if (!uniform_wg) prep_flush(ctg, cur, acc);
    // device-wide counters: one pair of atomics per workgroup, spread over COUNTER_SLOTS cache lines
    if (threadIdx.x == 0) {
        const u32 p = blk_cnt[0][0] + blk_cnt[0][1] + blk_cnt[0][2] + blk_cnt[0][3];
        const u32 c = blk_cnt[1][0] + blk_cnt[1][1] + blk_cnt[1][2] + blk_cnt[1][3];
        const u32 slot = (blockIdx.x % COUNTER_SLOTS) * 8u;
        if (p) atomicAdd(&g->prim_slots[slot], (u64)p);
        if (c) atomicAdd(&g->cons_slots[slot], (u64)c);
    }
}

#define COV_PREP_KERNEL(NAME, ATTR, PF, PB)                                                                                                           \
    template <bool WANT_IDENTITY, bool FILTER, bool MASKED>                                                                                      \
    __global__ __launch_bounds__(256) ATTR void NAME(Records r, const u32 *__restrict__ tlen, u32 n_targets, const uint8_t *__restrict__ mask,   \
                                                     FilterCfg f, DevContig *ctg, DevGlobal *g, uint2 *__restrict__ runs,                        \
                                                     double *__restrict__ identp, double *__restrict__ identn, PrepPartial *__restrict__ part,   \
                                                     TileIdx ti, int passes, int b_active, u32 *__restrict__ cx_list, u32 cx_list_cap) {         \
        prep_body<WANT_IDENTITY, FILTER, MASKED, PF, PB>(r, tlen, n_targets, mask, f, ctg, g, runs, identp, identn, part, ti, passes, b_active, cx_list, \
                                                 cx_list_cap);                                                                                   \
    }
// ONE compilation of this body ships, as the second implementation the tests force with COVERM_PREP_KERNEL=7 (the default is k_prep_lean,
// prep_lean.hip.h): one record per thread and pass, the roots of the dependent loads one pass ahead, seven waves per SIMD (65-72 registers, no
// scratch in any shape).  Round 5 measured four compilations of it (k_prep8s 0.535 ms, k_prep7s 0.541, k_prep6 0.544, k_prep5p 0.553 at BASELINE
// config 2, profiles/r05_prep_prefetch_ab.log): 2 % apart, all issue-bound on the same ~500 instructions per 64 records.
COV_PREP_KERNEL(k_prep7s, __attribute__((amdgpu_waves_per_eu(7, 7))), true, 1)
#undef COV_PREP_KERNEL

// One wave per contig: adds the partial records of the workgroups that lay entirely inside it.
__device__ __forceinline__ void prep_reduce_body(u32 c, DevContig *ctg, u32 n_targets, const PrepPartial *__restrict__ part, u32 n_parts, u32 chunk_recs) {
    if (c >= n_targets) return;
    DevContig *C = &ctg[c];
    const u32 rs = C->rec_start, re = C->rec_end;
    if (rs >= re) return;
    const u32 b0 = rs / chunk_recs, b1 = min((re - 1) / chunk_recs, n_parts - 1);
    u32 prim = 0, pass = 0, nons = 0, span = 0, first = 0xffffffffu, last = 0;
    u64 nm = 0, indel = 0;
    for (u32 b = b0 + (threadIdx.x & 63); b <= b1; b += 64) {
        const PrepPartial q = part[b];
        if (q.tid != (int)c) continue;
        prim += q.prim; pass += q.pass; nons += q.nons; span = max(span, q.span);
        first = min(first, q.first); last = max(last, q.last); nm += q.nm; indel += q.indel;
    }
    prim = wave_sum_u32(prim); pass = wave_sum_u32(pass); nons = wave_sum_u32(nons); span = wave_max_u32(span);
    first = wave_min_u32(first); last = wave_max_u32(last); nm = wave_sum_u64(nm); indel = wave_sum_u64(indel);
    if ((threadIdx.x & 63) == 0 && pass) {
        C->n_primary += prim; C->n_pass += pass; C->n_nonsupp += nons; C->sum_nm += nm; C->sum_indel += indel;
        C->max_span = max(C->max_span, span); C->first_rec = min(C->first_rec, first); C->last_rec = max(C->last_rec, last);
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

// Tile descriptor, two uint4 per tile: [2t] = (first candidate record, last, contig length, flags — bit 0: records
// of other contigs may be interleaved in the range, check tid); [2t+1] = (contig, tile start, 0, 0).
// Exclusive prefix sums of tcnt over all tiles: per 1024-tile block (k_tile_scan1) + over the block totals (k_tile_scan2).
// Two arrays per launch (the tile counts and the long-CIGAR bucket counts: both are ready once k_post_prep has run): blocks
// [0, n_blocks) scan the first, [n_blocks, 2 n_blocks) the second.
struct ScanPair { const u32 *cnt[2]; u32 *scan[2]; u32 *top[2]; u64 *total[2]; };
__global__ __launch_bounds__(1024) void k_tile_scan1(ScanPair sp, u32 n_tiles, u32 n_blocks) {
    __shared__ u32 wtot[16];
    const u32 which = blockIdx.x >= n_blocks ? 1u : 0u, blk = blockIdx.x - which * n_blocks;
    const u32 *__restrict__ tcnt = sp.cnt[which];
    const u32 t = blk * 1024u + threadIdx.x;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const u32 v = t < n_tiles ? tcnt[t] : 0u;
    const u32 inc = (u32)wave_incl_scan((int)v);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    u32 wbase = 0, tot = 0;
    for (int k = 0; k < 16; k++) { const u32 x = wtot[k]; if (k < w) wbase += x; tot += x; }
    if (t < n_tiles) sp.scan[which][t] = wbase + inc - v;
    if (threadIdx.x == 0) sp.top[which][blk] = tot;
}

__global__ __launch_bounds__(1024) void k_tile_scan2(ScanPair sp, u32 n_blocks) {
    __shared__ u32 wtot[16];
    __shared__ u32 carry_s;
    u32 *ttop = sp.top[blockIdx.x];
    u64 *total = sp.total[blockIdx.x];
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (u32 base = 0; base < n_blocks; base += 1024u) {
        const u32 b = base + threadIdx.x;
        const u32 v = b < n_blocks ? ttop[b] : 0u;
        const u32 inc = (u32)wave_incl_scan((int)v);
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        u32 wbase = 0, tot = 0;
        for (int k = 0; k < 16; k++) { const u32 x = wtot[k]; if (k < w) wbase += x; tot += x; }
        const u32 carry = carry_s;
        if (b < n_blocks) ttop[b] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (total != nullptr && threadIdx.x == 0) *total = carry_s;
}

// Expansion of RW_BUCKET records (long CIGARs): one wave per record walks the CIGAR 64 operations per step (as
// cigar_walk_wave does) and, for every M/=/X operation, visits the tiles it overlaps: FILL = false counts entries per
// tile, FILL = true writes (start, end) at the tile's cursor.  Lanes that hit the same tile share one atomic.
template <bool FILL>
__device__ __forceinline__ void cx_expand_body(const Records &r, const u32 *__restrict__ tlen, DevGlobal *g, const CxIdx &cx, const TileIdx &ti, u32 wave, u32 n_waves) {
    const int lane = lane_id();
    const u32 n = min(g->n_cx, cx.list_cap);
    if (FILL && g->cx_total > cx.runs_cap) return;     // the host grows the buffer and runs the pipeline again
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (u32 j = wave; j < n; j += n_waves) {
        const u32 i = cx.list[j];
        const int tid = r.tid[i];
        const u32 co0 = r.cigar_off[i], nops = r.cigar_off[i + 1] - co0;
        const u32 L = tlen[tid], t0 = ti.tile_first[tid];
        long long cur0 = r.pos[i];
        u32 wd_next = (u32)lane < nops ? r.cigar[co0 + (u32)lane] : 4u;
        for (u32 base = 0; base < nops; base += 64) {
            const u32 wd = wd_next;
            const u32 idn = base + 64u + (u32)lane;
            wd_next = idn < nops ? r.cigar[co0 + idn] : 4u;
            const u32 len = wd >> 4, bit = 1u << (wd & 15u);             // every len < 2^24 (k_prep's `big` test)
            const u32 fr = (bit & 0x18du) ? len : 0u;
            const u32 incl = (u32)wave_incl_scan((int)fr), excl = incl - fr;
            const long long sj = cur0 + (long long)excl;
            const bool m = (bit & 0x181u) != 0u && len > 0u && sj >= 0 && sj < (long long)L;
            const u32 sv = (u32)sj, ev = sv + len;
            const u32 ts = sv >> ti.shift, te = m ? (min(ev, L) - 1u) >> ti.shift : 0u;
            for (u32 k = 0; __any(m && ts + k <= te); k++) {              // k-th tile of every operation
                const bool v = m && ts + k <= te;
                const u32 key = t0 + ts + k;
                u32 slot = 0;
                for (u64 todo = __ballot(v); todo != 0;) {
                    const int l = __ffsll((long long)todo) - 1;
                    const u32 k0 = __builtin_amdgcn_readlane(key, l);
                    const u64 grp = __ballot(v && key == k0);
                    u32 b = 0;
                    if (lane == l) b = atomicAdd(FILL ? &cx.cur[k0] : &cx.cnt[k0], (u32)__popcll(grp));
                    if (FILL) {
                        b = __builtin_amdgcn_readlane(b, l);
                        if (v && key == k0) slot = b + (u32)__popcll(grp & lt);
                    }
                    todo &= ~grp;
                }
                if (FILL && v) cx.runs[(u64)cx.cscan[key] + cx.ctop[key >> 10] + slot] = make_uint2(sv, ev);
            }
            cur0 += (long long)__builtin_amdgcn_readlane(incl, 63);
        }
    }
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_cx_expand(Records r, const u32 *__restrict__ tlen, DevGlobal *g, CxIdx cx, TileIdx ti) {
    cx_expand_body<FILL>(r, tlen, g, cx, ti, blockIdx.x * 4u + (threadIdx.x >> 6), gridDim.x * 4u);
}

// What depends on k_prep alone, in one launch: blocks [0, n_red) add the workgroups' partial counters to their contigs (one wave per
// contig), the blocks behind them count the long-CIGAR bucket entries per tile (k_cx_expand<false>; nothing to do for short reads).
__global__ __launch_bounds__(256) void k_post_prep(Records r, const u32 *__restrict__ tlen, DevGlobal *g, CxIdx cx, TileIdx ti, DevContig *ctg, u32 n_targets,
                                                   const PrepPartial *__restrict__ part, u32 n_parts, u32 chunk_recs, u32 n_red) {
    if (blockIdx.x < n_red) prep_reduce_body(blockIdx.x * 4u + (threadIdx.x >> 6), ctg, n_targets, part, n_parts, chunk_recs);
    else cx_expand_body<false>(r, tlen, g, cx, ti, (blockIdx.x - n_red) * 4u + (threadIdx.x >> 6), (gridDim.x - n_red) * 4u);
}

// One thread per tile: candidate record range [x, y) from the tile counts (see TileIdx), contig length, generic flag.
template <bool WANT_HIST>
__global__ __launch_bounds__(256) void k_ranges(const u32 *__restrict__ tile_contig, const u32 *__restrict__ tile_start,
                                                u32 n_tiles, const u32 *__restrict__ tlen, const uint8_t *__restrict__ mask,
                                                DevContig *ctg, uint4 *__restrict__ desc, TileIdx ti,
                                                const u32 *__restrict__ tscan, const u32 *__restrict__ ttop, CxIdx cx,
                                                DevGlobal *__restrict__ g, u32 *__restrict__ slow_list) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const u32 c = tile_contig[t];
    DevContig *C = &ctg[c];
    uint4 out = make_uint4(0u, 0u, tlen[c], 0u);
    u32 cxn = 0, cxo = 0;
    if (C->n_pass != 0 && (mask == nullptr || mask[c])) {
        const u32 rs = C->rec_start, re = C->rec_end;
        // buckets that did not fit were not filled: this pass is discarded and repeated by the host (cov_finish)
        cxn = g->cx_total <= cx.runs_cap ? cx.cnt[t] : 0u;
        cxo = cx.cscan[t] + cx.ctop[t >> 10];
        if (C->n_groups != 1u) out.w = 1u;
        if (C->n_groups != 1u || (C->flags & F_POS_UNSORTED)) {
            out.x = rs; out.y = re;  // generic path: every tile of this contig scans the whole span
        } else {
            const u32 tf = ti.tile_first[c];
            const u32 before = (tscan[t] + ttop[t >> 10]) - (tscan[tf] + ttop[tf >> 10]);   // records starting left of the tile
            const u32 f = rs + before;                    // F(lo_t)
            out.y = min(f + ti.tcnt[t], re);              // F(lo_t + tile)
            out.x = max(min(f, ti.fov[t]), rs);
            out.x = min(out.x, out.y);
        }
    }
    // Tiles the default kernel (k_pileup_fast) does not take: records of other contigs interleaved in the range (per-record
    // tid test), or more candidates than its u16 count tables can hold.
    if (out.x < out.y || cxn) {
        if (out.w || out.y - out.x + cxn > FAST_MAX_CAND) {
            out.w |= TILE_F_SLOW;
            if (slow_list != nullptr) slow_list[atomicAdd(&g->n_slow, 1u)] = t;
        }
    }
    if (WANT_HIST) {   // one atomic per wave when all its tiles belong to one contig (the usual case)
        // depth <= candidates + bucket entries, and <= considered records of the contig (the runs of one record are
        // disjoint); the second bound also keeps the arena within R + n_targets bins when contigs interleave
        const u32 npass = (u32)min(C->n_pass, (u64)0xffffffffu);
        const u32 cap = min(out.y - out.x + cxn, npass);
        if (__all(c == (u32)__builtin_amdgcn_readfirstlane((int)c))) {
            const u32 m = wave_max_u32(cap);
            if (lane_id() == 0 && m) atomicMax(&C->hist_cap, m);
        } else if (cap) atomicMax(&C->hist_cap, cap);
    }
    desc[2 * t] = out;
    desc[2 * t + 1] = make_uint4(c, tile_start[t], cxo, cxn);
}

// ------------------------------------------------------------------------------------ histogram layout
// Exclusive scan of per-contig bin counts -> offsets, in contig order.  MODE 0: arena layout from hist_cap (an upper bound on depth);
// MODE 1: compact layout from the realised max depth.  Two launches over blocks of 1024 contigs (one contig per thread): k_hist_sum leaves
// every block's total, k_hist_off adds the totals of the blocks in front of it (every block sums them itself: nT / 1024 values at most)
// to its own exclusive scan.  (Until round 6 ONE workgroup walked all contigs: 17 us at 5 000 contigs, 1.06 ms at 200 000, 19.3 ms at
// 2 000 000 — profiles/r06_contig_sweep_before.log.)
template <int MODE>
__device__ __forceinline__ u64 hist_bins_of(const DevContig *C, u32 c, const u32 *__restrict__ tlen, const uint8_t *__restrict__ mask, u64 excl) {
    const bool live = C->n_pass != 0 && (mask == nullptr || mask[c]);
    if (MODE == 0) return live ? (u64)C->hist_cap + 1 : 0;
    return (live && 2 * excl < (u64)tlen[c]) ? (u64)C->max_d + 1 : 0;
}
// inclusive scan over the 1024 threads of a block (16 waves); returns this thread's inclusive prefix, `total` = the block's sum
__device__ __forceinline__ u64 block_incl_scan_u64(u64 v, u64 *wtot, u64 &total) {
    const int lane = lane_id(), w = threadIdx.x >> 6;
    u64 inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    u64 wbase = 0; total = 0;
    for (int k = 0; k < 16; k++) { const u64 x = wtot[k]; if (k < w) wbase += x; total += x; }
    __syncthreads();
    return wbase + inc;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k_hist_sum(const DevContig *__restrict__ ctg, u32 n_targets, const u32 *__restrict__ tlen,
                                                   const uint8_t *__restrict__ mask, u64 excl, u64 *__restrict__ top) {
    __shared__ u64 wtot[16];
    const u32 c = blockIdx.x * 1024u + threadIdx.x;
    u64 total;
    (void)block_incl_scan_u64(c < n_targets ? hist_bins_of<MODE>(&ctg[c], c, tlen, mask, excl) : 0, wtot, total);
    if (threadIdx.x == 0) top[blockIdx.x] = total;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k_hist_off(DevContig *ctg, u32 n_targets, const u32 *__restrict__ tlen, const uint8_t *__restrict__ mask,
                                                   u64 excl, const u64 *__restrict__ top, DevGlobal *g) {
    __shared__ u64 wtot[16];
    u64 mine = 0, base;
    for (u32 b = threadIdx.x; b < blockIdx.x; b += 1024u) mine += top[b];
    (void)block_incl_scan_u64(mine, wtot, base);          // base = bins of all contigs in front of this block
    const u32 c = blockIdx.x * 1024u + threadIdx.x;
    const u64 v = c < n_targets ? hist_bins_of<MODE>(&ctg[c], c, tlen, mask, excl) : 0;
    u64 total;
    const u64 inc = block_incl_scan_u64(v, wtot, total);
    if (c < n_targets) {
        if (MODE == 0) ctg[c].hist_off = base + inc - v;
        else { ctg[c].chist_off = base + inc - v; ctg[c].hist_len = (u32)v; }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        if (MODE == 0) g->hist_cap_total = base + total; else g->chist_total = base + total;
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
    const u32 c = blockIdx.x * 4u + (threadIdx.x >> 6);      // one wave per contig (a workgroup per contig until round 6: 2 M workgroups for an assembly)
    if (c >= n_targets) return;
    const DevContig *C = &ctg[c];
    const u32 n = C->hist_len;
    if (n == 0) return;
    const u64 win_len = (u64)tlen[c] - 2 * excl;
    for (u32 d = threadIdx.x & 63u; d < n; d += 64u) {
        u64 v = arena[C->hist_off + d];
        if (d == 0) v += win_len - C->proc_win;
        out[C->chist_off + d] = v;
    }
}

// ------------------------------------------------------------------------------------ k_identity
// File-order f64 sums, bit-identical to the reference's sequential `+=` (contig.rs:208-211).  The additions of
// one contig form a strict dependency chain (f64 addition is not associative), so one wave walks each contig:
// 64 values per coalesced load, then every lane replays the same 64 dependent adds from lane broadcasts.
// k_prep wrote two value streams: identn (not supplementary) and identp (primary).  Runs on the session's side
// stream, beside k_ranges / k_pileup.  Cost ~10 cycles per record of the LONGEST contig; an exact parallel
// formulation (integer summation per binade of the running sum) is planned — see DESIGN.md.
__global__ __launch_bounds__(64) void k_identity(DevContig *ctg, u32 n_targets, const double *__restrict__ identp,
                                                 const double *__restrict__ identn, const int32_t *__restrict__ tidv) {
    const u32 c = blockIdx.x;
    if (c >= n_targets) return;
    DevContig *C = &ctg[c];
    if (C->n_pass == 0) return;
    const u32 rs = C->rec_start, re = C->rec_end;
    const bool generic = C->n_groups != 1u;
    double accp = 0.0, accn = 0.0;
    const int lane = lane_id();
    u32 i = rs + (u32)lane;
    double xp = 0.0, xn = 0.0;
    if (i < re && (!generic || tidv[i] == (int)c)) { if (identp) xp = identp[i]; if (identn) xn = identn[i]; }
    for (u32 b = rs; b < re; b += 64) {
        const double cp = xp, cn = xn;
        i += 64;   // prefetch the next 64 while the dependent adds of this batch run
        xp = 0.0; xn = 0.0;
        if (i < re && (!generic || tidv[i] == (int)c)) { if (identp) xp = identp[i]; if (identn) xn = identn[i]; }
#pragma unroll 16
        for (int k = 0; k < 64; k++) {
            accn += __shfl(cn, k);
            accp += __shfl(cp, k);
        }
    }
    if (lane == 0) { C->id_primary = accp; C->id_nonsupp = accn; }
}

// ------------------------------------------------------------------------------------ exact parallel identity sums
// The reference adds one f64 per read to a running sum (contig.rs:208-211); that is a serial chain of rounded
// additions.  It is nevertheless parallelisable EXACTLY: while the running sum S stays inside one binade
// [2^e, 2^(e+1)) its ulp u = 2^(e-52) is constant, S = m*u with m an integer, and for 0 <= x <= 1 <= S
//     fl(S + x) = (m + q + r) * u,   q = floor(x/u),  r = 1 if frac(x/u) > 1/2, 0 if < 1/2
// (x/u is an exact shift of x's mantissa).  Only an exact tie frac == 1/2 depends on m (round-half-even) and only a
// binade crossing changes u.  So a chunk of reads contributes the INTEGER sum of (q + r), computable in any order,
// provided it has no tie and S does not leave the binade — which the combine step verifies against the exact S,
// falling back to the serial chain for the few chunks that cross a binade, contain a tie or a value outside
// [0, 1], or straddle two contigs.  Four small kernels:
//   k_id_approx   per 1024-record chunk: approximate f64 sums (tree order) + "values outside [0,1]" flag
//   k_id_predict  per contig: approximate prefix -> predicted binade of S at every interior chunk
//   k_id_exact    per chunk : exact integer sum at the predicted binade + tie flag
//   k_id_combine  per contig: exact running sum, verified fast path or serial chain per chunk
constexpr u32 ID_CH = 1024;
struct IdChunk {
    double ap, an;      // approximate sums (primary / not-supplementary stream)
    u64 tp, tn;         // exact integer sums in units of 2^(e-52)
    int ep, en;         // predicted binade, or INT_MIN = do this chunk serially
    u32 flags;          // bit0 irregular value in p stream, bit1 in n stream, bit2 tie p, bit3 tie n
    u32 pad;
};
constexpr int ID_SERIAL = -2147483647 - 1;

__device__ __forceinline__ int f64_exp(double v) { return (int)((__double_as_longlong(v) >> 52) & 0x7ff) - 1023; }

__global__ __launch_bounds__(256) void k_id_approx(const double *__restrict__ identp, const double *__restrict__ identn,
                                                   u32 n, IdChunk *__restrict__ ch) {
    __shared__ double sp[4], sn[4];
    __shared__ u32 sf[4];
    const u32 k = blockIdx.x, base = k * ID_CH;
    double ap = 0.0, an = 0.0;
    u32 fl = 0;
#pragma unroll
    for (int j = 0; j < (int)(ID_CH / 256); j++) {
        const u32 i = base + (u32)j * 256u + threadIdx.x;
        if (i < n) {
            const double xp = identp ? identp[i] : 0.0, xn = identn ? identn[i] : 0.0;
            if (!(xp >= 0.0 && xp <= 1.0)) fl |= 1u;
            if (!(xn >= 0.0 && xn <= 1.0)) fl |= 2u;
            ap += xp; an += xn;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ap += __shfl_xor(ap, o); an += __shfl_xor(an, o); fl |= __shfl_xor(fl, o); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sp[w] = ap; sn[w] = an; sf[w] = fl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        IdChunk c;
        c.ap = sp[0] + sp[1] + sp[2] + sp[3]; c.an = sn[0] + sn[1] + sn[2] + sn[3];
        c.tp = c.tn = 0; c.ep = c.en = ID_SERIAL; c.flags = sf[0] | sf[1] | sf[2] | sf[3]; c.pad = 0;
        ch[k] = c;
    }
}

__global__ __launch_bounds__(64) void k_id_predict(const DevContig *__restrict__ ctg, u32 n_targets,
                                                   const double *__restrict__ identp, const double *__restrict__ identn,
                                                   IdChunk *__restrict__ ch) {
    const u32 c = blockIdx.x;
    if (c >= n_targets) return;
    const DevContig *C = &ctg[c];
    if (C->n_pass == 0 || C->n_groups != 1u) return;
    const u32 rs = C->rec_start, re = C->rec_end;
    const u32 k0 = (rs + ID_CH - 1) / ID_CH, k1 = re / ID_CH;   // interior chunks [k0, k1)
    if (k0 >= k1) return;
    // approximate sum of the head partial segment [rs, k0*ID_CH)
    double pp = 0.0, pn = 0.0;
    for (u32 i = rs + (threadIdx.x & 63); i < k0 * ID_CH; i += 64) { if (identp) pp += identp[i]; if (identn) pn += identn[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pp += __shfl_xor(pp, o); pn += __shfl_xor(pn, o); }
    const int lane = threadIdx.x & 63;
    const double lo_f = 1.0 - 1.0 / 1048576.0, hi_f = 1.0 + 1.0 / 1048576.0;
    // the prefix is only a prediction, so any summation order will do: 64 chunks per step, wave prefix scan
    for (u32 kb = k0; kb < k1; kb += 64) {
        const u32 k = kb + (u32)lane;
        const bool live = k < k1;
        const double ap = live ? ch[k].ap : 0.0, an = live ? ch[k].an : 0.0;
        const u32 fl = live ? ch[k].flags : 3u;
        double ip = ap, in_ = an;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double tp = __shfl_up(ip, o), tn = __shfl_up(in_, o);
            if (lane >= o) { ip += tp; in_ += tn; }
        }
        const double sp = pp + (ip - ap), sn = pn + (in_ - an);   // prefix before this chunk
        if (live) {
            int e = ID_SERIAL;
            if (identp && !(fl & 1u) && sp >= 2.0 && f64_exp(sp * lo_f) == f64_exp((sp + ap) * hi_f)) e = f64_exp(sp);
            ch[k].ep = e;
            e = ID_SERIAL;
            if (identn && !(fl & 2u) && sn >= 2.0 && f64_exp(sn * lo_f) == f64_exp((sn + an) * hi_f)) e = f64_exp(sn);
            ch[k].en = e;
        }
        pp += __shfl(ip, 63); pn += __shfl(in_, 63);
    }
}

// (q + r) of one value at binade e; sets tie when frac(x/u) == 1/2 exactly
__device__ __forceinline__ u64 id_units(double x, int e, bool &tie) {
    const u64 b = (u64)__double_as_longlong(x);
    if ((b << 1) == 0) return 0;                      // +-0
    const int ex = (int)((b >> 52) & 0x7ff);
    if (ex == 0) { tie = true; return 0; }            // subnormal: let the serial path handle it
    const u64 mant = (b & 0xfffffffffffffull) | (1ull << 52);
    const int sh = e - (ex - 1023);                   // x/u = mant >> sh
    if (sh <= 0) { if (sh < 0) tie = true; return sh == 0 ? mant : 0; }
    if (sh >= 54) return 0;
    const u64 q = mant >> sh, rem = mant & ((1ull << sh) - 1), half = 1ull << (sh - 1);
    if (rem == half) tie = true;
    return q + (rem > half ? 1ull : 0ull);
}

__global__ __launch_bounds__(256) void k_id_exact(const double *__restrict__ identp, const double *__restrict__ identn,
                                                  u32 n, IdChunk *__restrict__ ch) {
    __shared__ u64 sp[4], sn[4];
    __shared__ u32 sf[4];
    const u32 k = blockIdx.x, base = k * ID_CH;
    const int ep = ch[k].ep, en = ch[k].en;
    if (ep == ID_SERIAL && en == ID_SERIAL) return;
    u64 tp = 0, tn = 0;
    bool tiep = false, tien = false;
#pragma unroll
    for (int j = 0; j < (int)(ID_CH / 256); j++) {
        const u32 i = base + (u32)j * 256u + threadIdx.x;
        if (i < n) {
            if (ep != ID_SERIAL) tp += id_units(identp[i], ep, tiep);
            if (en != ID_SERIAL) tn += id_units(identn[i], en, tien);
        }
    }
    u32 fl = (tiep ? 4u : 0u) | (tien ? 8u : 0u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { tp += __shfl_xor(tp, o); tn += __shfl_xor(tn, o); fl |= __shfl_xor(fl, o); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sp[w] = tp; sn[w] = tn; sf[w] = fl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ch[k].tp = sp[0] + sp[1] + sp[2] + sp[3]; ch[k].tn = sn[0] + sn[1] + sn[2] + sn[3];
        ch[k].flags |= sf[0] | sf[1] | sf[2] | sf[3];
    }
}

// adds T units of 2^(e-52) to S if S is in binade e and stays inside it; returns false if not applicable
__device__ __forceinline__ bool id_fast_add(double &S, int e, u64 T) {
    const u64 b = (u64)__double_as_longlong(S);
    if (e == ID_SERIAL || (int)((b >> 52) & 0x7ff) - 1023 != e) return false;
    const u64 m = (b & 0xfffffffffffffull) | (1ull << 52);
    if (T >= (1ull << 53) || m + T >= (1ull << 53)) return false;
    S = __longlong_as_double((long long)((b & 0xfff0000000000000ull) | ((m + T) & 0xfffffffffffffull)));
    return true;
}

__global__ __launch_bounds__(64) void k_id_combine(DevContig *ctg, u32 n_targets, const double *__restrict__ identp,
                                                   const double *__restrict__ identn, const int32_t *__restrict__ tidv,
                                                   const IdChunk *__restrict__ ch) {
    const u32 c = blockIdx.x;
    if (c >= n_targets) return;
    DevContig *C = &ctg[c];
    if (C->n_pass == 0) return;
    const u32 rs = C->rec_start, re = C->rec_end;
    const bool generic = C->n_groups != 1u;
    const int lane = lane_id();
    double Sp = 0.0, Sn = 0.0;
    // Exact in-order addition of 64 values (lane order) to S.  While S stays in one binade every addition is an
    // integer addition of round(x/ulp) to the mantissa, so a wave prefix sum applies all values up to the first one
    // that leaves the binade, is an exact rounding tie, or is not in [0, 1]; that one value takes a true f64 add (the
    // reference's own operation) and the rest continue in the new binade.  Replaces a 64-step dependent add chain.
    auto block_add = [&](double &S, const double xv) {
        const bool irregular = !(xv >= 0.0 && xv <= 1.0);
        u32 q = 0;
        while (q < 64u) {
            const u64 bits = (u64)__double_as_longlong(S);
            const int e_cur = (int)((bits >> 52) & 0x7ff) - 1023;
            if (e_cur < 1 || (bits >> 63)) { S += bcast_f64(xv, (int)q); q++; continue; }   // |S| < 2 or S < 0 (negative identities: NM > aligned): plain adds
            const u64 m = (bits & 0xfffffffffffffull) | (1ull << 52);
            bool tie = false;
            const u64 t = irregular ? 0ull : id_units(xv, e_cur, tie);        // <= 2^52 for x <= 1, e >= 1
            const bool in_range = (u32)lane >= q;
            u64 P = in_range ? t : 0ull;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const u64 u = __shfl_up(P, o); if (lane >= o) P += u; }
            const bool fits = in_range && !tie && !irregular && m + P < (1ull << 53);
            const u64 failm = __ballot(in_range && !fits);
            const u32 jstar = failm ? (u32)(__ffsll((long long)failm) - 1) : 64u;
            if (jstar > q) {
                const u64 T = bcast_u64(P, (int)(jstar - 1));
                S = __longlong_as_double((long long)((bits & 0xfff0000000000000ull) | ((m + T) & 0xfffffffffffffull)));
            }
            if (jstar < 64u) S += bcast_f64(xv, (int)jstar);
            q = jstar + 1;
        }
    };
    auto serial = [&](u32 from, u32 to, bool need_p, bool need_n) {   // the reference's serial chain over [from, to)
        auto fetch = [&](u32 b, double &xp, double &xn) {
            const u32 i = b + (u32)lane;
            xp = 0.0; xn = 0.0;
            if (i < to && (!generic || tidv[i] == (int)c)) {
                if (need_p) xp = identp[i];
                if (need_n) xn = identn[i];
            }
        };
        need_p = need_p && identp != nullptr; need_n = need_n && identn != nullptr;
        double xp, xn;
        if (from < to) fetch(from, xp, xn);
        for (u32 b = from; b < to; b += 64) {
            const double cp = xp, cn = xn;
            if (b + 64 < to) fetch(b + 64, xp, xn);      // next block's values are in flight during this one
            if (need_n) block_add(Sn, cn);
            if (need_p) block_add(Sp, cp);
        }
    };
    const u32 k0 = generic ? 0u : (rs + ID_CH - 1) / ID_CH, k1 = generic ? 0u : re / ID_CH;   // interior chunks
    if (k0 >= k1) { serial(rs, re, true, true); }
    else {
        serial(rs, k0 * ID_CH, true, true);
        for (u32 kb = k0; kb < k1; kb += 64) {      // 64 chunk records per load, then walk them in order
            const u32 kk = kb + (u32)lane;
            IdChunk x;
            if (kk < k1) x = ch[kk]; else { x.tp = x.tn = 0; x.ep = x.en = ID_SERIAL; x.flags = 15u; }
            const u32 nb = min(64u, k1 - kb);
            // All chunks of the batch whose predicted binade is the binade S is in, and that keep S inside it, are
            // verified together: S only grows, so "the running mantissa after chunk j is still < 2^53" for the last
            // chunk implies it for every earlier one.  One wave prefix sum finds the first chunk that does not fit
            // (binade crossing, tie, irregular value, unpredicted); that one is replayed serially and the rest of the
            // batch continues from there.  A contig of 1.5 M reads costs ~25 steps + ~20 replays instead of 1500 steps.
            auto run_stream = [&](double &S, const int e_l, const u64 t_l, const bool bad_l, const bool is_p) {
                u32 q = 0;
                while (q < nb) {
                    const u64 bits = (u64)__double_as_longlong(S);
                    const int e_cur = (int)((bits >> 52) & 0x7ff) - 1023;
                    const u64 m = (bits & 0xfffffffffffffull) | (1ull << 52);
                    const bool in_range = (u32)lane >= q && (u32)lane < nb;
                    u64 P = in_range ? t_l : 0ull;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const u64 t = __shfl_up(P, o); if (lane >= o) P += t; }
                    const bool fits = in_range && !bad_l && !(bits >> 63) && e_l == e_cur && t_l < (1ull << 53) &&
                                      P < (1ull << 53) && m + P < (1ull << 53);
                    const u64 failm = __ballot(in_range && !fits);
                    const u32 jstar = failm ? (u32)(__ffsll((long long)failm) - 1) : nb;
                    if (jstar > q) {
                        const u64 T = bcast_u64(P, (int)(jstar - 1));
                        S = __longlong_as_double((long long)((bits & 0xfff0000000000000ull) | ((m + T) & 0xfffffffffffffull)));
                    }
                    if (jstar < nb) serial((kb + jstar) * ID_CH, (kb + jstar + 1) * ID_CH, is_p, !is_p);
                    q = jstar + 1;
                }
            };
            if (identp != nullptr) run_stream(Sp, x.ep, x.tp, (x.flags & (1u | 4u)) != 0u, true);
            if (identn != nullptr) run_stream(Sn, x.en, x.tn, (x.flags & (2u | 8u)) != 0u, false);
        }
        serial(k1 * ID_CH, re, true, true);
    }
    if (lane == 0) { C->id_primary = Sp; C->id_nonsupp = Sn; }
}

// ------------------------------------------------------------------------------------ k_pileup
// Histogram bins beyond the LDS window go straight to the contig's slice of the global arena.  Rare (depth
// >= 512/1024/2048), so it is kept out of line: inlined at every call site it costs dozens of branches.
__device__ __attribute__((noinline)) void hist_add_overflow(u32 *arena, u64 hoff, u32 hcap, DevGlobal *g, u32 d, u32 x) {
    if (d <= hcap) atomicAdd(&arena[hoff + d], x);
    else atomicOr(&g->internal_error, 1u);
}

struct PileupArgs {
    const u32 *tile_contig, *tile_start;
    const uint4 *desc;
    const uint2 *runs;
    const uint2 *cx_runs;   // per-tile buckets of RW_BUCKET records (desc[2t+1].z = offset, .w = count)
    Records r;
    DevContig *ctg;
    DevGlobal *g;
    u32 *hist_arena;
    u64 excl;
    int32_t *depth_out;   // WRITE_DEPTH: depth of one contig, or of every contig when depth_off is set
    const u64 *depth_off; // WRITE_DEPTH over all tiles: offset of each contig in depth_out (NULL = depth_out is one contig)
    u32 tile_base;        // first tile index handled by blockIdx 0
};

// (k_pileup<TILE, NT> — round 1's kernel, one workgroup per 4096 / 8192 / 16384-base tile with the tile's i32 depth array in LDS; the tests'
// third pileup implementation until round 6 — is gone: k_pileup_fast2t and k_pileup_stream (COVERM_PILEUP=stream) are the cross-checks.)

// ------------------------------------------------------------------------------------ k_pileup_stream
// Barrier-free variant: every WAVE owns a private LDS tile of TW = 1024 bases (+ a private 512-bin histogram)
// and walks a chunk of consecutive tiles, keeping its running sums in registers across tiles of one contig and
// flushing them (one set of global atomics + histogram sweep) only when the contig changes or the chunk ends.
// No __syncthreads anywhere: the resident waves of a CU interleave their load latencies independently.
//
// Layout inside a tile is TRANSPOSED: base p lives in LDS dword p and is owned by lane p & 63 in row p >> 6.
// That makes every per-base step a whole-wave instruction over 64 consecutive bases, and turns stream compaction
// into ballot + popcount (scalar unit) + mbcnt.  Statistics take one of two exact paths per tile:
//   sparse  (#changed positions <= 256): the non-zero deltas are compacted in position order into (pos, delta)
//           entries, depth comes from one wave scan per 64 ENTRIES, and the sums are formed per constant-depth
//           segment (d*len, d^2*len, one histogram add per segment) — work ~ alignment ends, not bases;
//   dense   otherwise (deep piles), or when depth is written out: one wave scan per 64 bases, per-base sums,
//           change-point histogram.
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int STREAM_TW = 1024, STREAM_HB = 512, STREAM_CAP = 512;

template <bool WANT_HIST, bool WRITE_DEPTH>
__global__ __launch_bounds__(256) void k_pileup_stream(PileupArgs a, u32 n_tiles, u32 chunk_tiles,
                                                       const u32 *__restrict__ tile_list, const u32 *__restrict__ n_list) {
    if (tile_list != nullptr) n_tiles = *n_list;      // slow-tile mode: only the tiles k_ranges listed (any order)
    constexpr int TW = STREAM_TW, HBW = STREAM_HB, CAP = STREAM_CAP, ROWS = TW / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int *tile = reinterpret_cast<int *>(smem + (size_t)w * (TW * 4 + HBW * 4));
    u32 *lhist = reinterpret_cast<u32 *>(tile + TW);
    int4 *t4 = reinterpret_cast<int4 *>(tile);
    int2 *ent = reinterpret_cast<int2 *>(tile);   // sparse path: CAP entries reuse dwords [0, 2*CAP)
    const u32 wave_id = blockIdx.x * 4u + (u32)w, n_waves = gridDim.x * 4u;
    if (WANT_HIST) {
#pragma unroll
        for (int b = lane; b < HBW; b += 64) lhist[b] = 0u;
    }
    u64 sum_d = 0, sum_d2 = 0, proc_win = 0;
    u32 cov_w = 0, cov_f = 0, mn = 0xffffffffu, mx = 0;
    int cur_c = -1;
    u64 hoff = 0; u32 hcap = 0;
    const u64 excl = a.excl;

    auto hist_add = [&](u32 d, u32 x) {
        if (__builtin_expect(d < (u32)HBW, 1)) atomicAdd(&lhist[d], x);
        else hist_add_overflow(a.hist_arena, hoff, hcap, a.g, d, x);
    };
    auto flush = [&]() {
        if (cur_c >= 0) {
            const u64 s1 = wave_sum_u64(sum_d), s2 = wave_sum_u64(sum_d2);
            const u32 c1 = wave_sum_u32(cov_w), c2 = wave_sum_u32(cov_f);
            const u32 m1 = wave_min_u32(mn), m2 = wave_max_u32(mx);
            DevContig *C = &a.ctg[cur_c];
            if (lane == 0) {
                if (s1) atomicAdd(&C->sum_d, s1);
                if (s2) atomicAdd(&C->sum_d2, s2);
                if (c1) atomicAdd(&C->cov_win, (u64)c1);
                if (c2) atomicAdd(&C->cov_full, (u64)c2);
                if (proc_win) {
                    atomicAdd(&C->proc_win, proc_win);
                    atomicMin(&C->min_d, m1);
                    atomicMax(&C->max_d, m2);
                }
            }
            if (WANT_HIST && proc_win) {
                lds_fence();
                const u32 hi_b = min(m2, (u32)HBW - 1u);
                for (u32 b = m1 + (u32)lane; b <= hi_b; b += 64) {
                    const u32 x = lhist[b];
                    if (x) { atomicAdd(&a.hist_arena[hoff + b], x); lhist[b] = 0u; }
                }
                lds_fence();
            }
        }
        sum_d = sum_d2 = 0; proc_win = 0; cov_w = cov_f = 0; mn = 0xffffffffu; mx = 0;
    };

    // Tile descriptors are wave-uniform: scalar loads straight into SGPRs.  (An earlier version prefetched descriptors
    // two tiles and run words one tile ahead through the vector path; it cost 24 VGPRs = one wave per SIMD of occupancy
    // and measured ~5 % slower than loading at the start of the tile.)
    auto load_runs = [&](const uint4 &d, uint2 &r0, uint2 &r1) {
        const u32 i0 = d.x + (u32)lane, i1 = i0 + 64u;
        r0 = i0 < d.y ? a.runs[i0] : make_uint2(0u, 0u);
        r1 = i1 < d.y ? a.runs[i1] : make_uint2(0u, 0u);
    };
    // Static round-robin over chunks.  (Chunks handed out from eight sharded counters — a wave draining its home shard, then stealing — were
    // built in round 2 and measured ~10 % slower on the benchmark workload: dequeue latency, hot counters.  Removed in round 6.)
    const u32 n_chunks = (n_tiles + chunk_tiles - 1) / chunk_tiles;
    u32 ch = wave_id < n_chunks ? wave_id : 0xffffffffu;
    while (ch != 0xffffffffu) {
        const u32 ch_next = ch + n_waves < n_chunks ? ch + n_waves : 0xffffffffu;
        const u32 t0 = a.tile_base + ch * chunk_tiles, t1 = a.tile_base + min((ch + 1) * chunk_tiles, n_tiles);
        for (u32 tq = t0; tq < t1; tq++) {
            const u32 t = tile_list != nullptr ? tile_list[tq] : tq;
            const uint4 ds = a.desc[2 * (size_t)t], dC1 = a.desc[2 * (size_t)t + 1];
            uint2 rw0, rw1;
            load_runs(ds, rw0, rw1);
            const u32 c = dC1.x, lo = dC1.y, L = ds.z;
            const u64 dbase = (WRITE_DEPTH && a.depth_off != nullptr) ? a.depth_off[c] : 0ull;
            const u32 cxo = dC1.z, cxn = dC1.w;
            if (ds.x >= ds.y && cxn == 0u) continue;   // depth 0 everywhere: accounted on the host side
            const bool generic = ds.w & 1u;
            if ((int)c != cur_c) {
                flush();
                cur_c = (int)c;
                if (WANT_HIST) { hoff = a.ctg[c].hist_off; hcap = a.ctg[c].hist_cap; }
            }
            const u32 tlen_t = min((u32)TW, L - lo);
            // ---- zero, scatter events
#pragma unroll
            for (int k = 0; k < TW / 256; k++) t4[k * 64 + lane] = make_int4(0, 0, 0, 0);
            lds_fence();
            const u32 hi = lo + TW;
            auto add_run = [&](u32 s, u32 e) {
                if (s < hi && e > lo) {
                    const u32 s0 = s > lo ? s - lo : 0u;
                    atomicAdd(&tile[s0], 1);
                    if (e < hi) atomicAdd(&tile[e - lo], -1);
                }
            };
            auto apply = [&](const uint2 rw, u32 i) {
                if (rw.y == 0u) return;
                if (generic && a.r.tid[i] != (int)c) return;
                const u32 type = rw.y >> 30;
                if (type <= RW_DOUBLE) {
                    // one shared first run for both encodings (a wave almost always holds both kinds, so separate
                    // branches would issue the run's two LDS atomics twice), second run only for RW_DOUBLE lanes
                    const bool dbl = type == RW_DOUBLE;
                    const u32 l1 = dbl ? (rw.y & 1023u) : rw.y;
                    add_run(rw.x, rw.x + l1);
                    if (dbl) {
                        const u32 s2 = rw.x + l1 + ((rw.y >> 10) & 255u);
                        add_run(s2, s2 + ((rw.y >> 18) & 1023u));
                    }
                } else if (type == RW_COMPLEX) {
                    u32 cursor = (u32)a.r.pos[i];
                    const u32 c0 = a.r.cigar_off[i], c1 = a.r.cigar_off[i + 1];
                    for (u32 k = c0; k < c1; k++) {
                        const u32 wd = a.r.cigar[k];
                        const u32 op = wd & 15u, len = wd >> 4;
                        if (op == 0u || op == 7u || op == 8u) { add_run(cursor, cursor + len); cursor += len; }
                        else if (op == 2u || op == 3u) cursor += len;
                    }
                }   // RW_BUCKET: delivered through the tile's bucket
            };
            apply(rw0, ds.x + (u32)lane);
            apply(rw1, ds.x + 64u + (u32)lane);
            for (u32 i = ds.x + 128u + (u32)lane; i < ds.y; i += 64) apply(a.runs[i], i);   // deep tiles only
            for (u32 j = (u32)lane; j < cxn; j += 64) { const uint2 q = a.cx_runs[(u64)cxo + j]; add_run(q.x, q.y); }   // long reads
            lds_fence();
            // ---- read the rows back (lane = base mod 64) and count changed positions on the scalar unit
            int x[ROWS];
            u32 E = 0;
#pragma unroll
            for (int j = 0; j < ROWS; j++) x[j] = tile[j * 64 + lane];
#pragma unroll
            for (int j = 0; j < ROWS; j++) E += (u32)__popcll(__ballot(x[j] != 0));
            const bool has_win = 2 * excl < (u64)L;
            const u32 ws = has_win ? (u32)excl : 0u, we = has_win ? (u32)(L - excl) : 0u;
            const u32 wst = max(ws, lo), wet = min(we, lo + tlen_t);
            const bool win_any = has_win && wst < wet;
            if (win_any) proc_win += (u64)(wet - wst);

            if (!WRITE_DEPTH && E <= (u32)CAP) {
                // ---- sparse path.  LDS requests of one wave are served in order, so the entry writes below cannot
                // overtake the row reads above.
                u32 base = 0;
#pragma unroll
                for (int j = 0; j < ROWS; j++) {
                    const u64 m = __ballot(x[j] != 0);
                    if (x[j] != 0) {
                        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                        ent[base + rank] = make_int2(j * 64 + lane, x[j]);
                    }
                    base += (u32)__popcll(m);
                }
                lds_fence();
                const u32 wl0 = win_any ? wst - lo : 0u, wl1 = win_any ? wet - lo : 0u;
                // leading zero-depth segment [0, first entry)
                if (lane == 0) {
                    const u32 en = min(E ? (u32)ent[0].x : tlen_t, tlen_t);
                    const u32 a1 = min(en, wl1);
                    if (a1 > wl0) { mn = 0u; if (WANT_HIST) hist_add(0u, a1 - wl0); }
                }
                int carry = 0;
                // Per-tile 32-bit partial sums: while depth < 1024 the products fit 24-bit multiplies (full rate; a
                // 64-bit multiply is several quarter-rate instructions) and, the segments of a tile being disjoint,
                // sum(d*len) <= 2^20 and sum(d^2*len) <= 2^30 per lane.
                u32 s1t = 0, s2t = 0;
                for (u32 e0 = 0; e0 < E; e0 += 64) {
                    const u32 e = e0 + (u32)lane;
                    const bool live = e < E;
                    const int2 en = live ? ent[e] : make_int2(0, 0);
                    const u32 nxt = (e + 1 < E) ? (u32)ent[e + 1].x : tlen_t;
                    const int inc = wave_incl_scan(en.y);
                    const int d = carry + inc;
                    carry += __builtin_amdgcn_readlane(inc, 63);
                    const u32 du = (u32)d;
                    const bool small = !__any(live && du >= 1024u);
                    if (live) {
                        const u32 s = min((u32)en.x, tlen_t), t_ = min(nxt, tlen_t);
                        if (d > 0) cov_f += t_ - s;
                        const u32 a0 = max(s, wl0), a1 = min(t_, wl1);
                        if (a1 > a0) {
                            const u32 len = a1 - a0;
                            if (small) { s1t += __umul24(du, len); s2t += __umul24(__umul24(du, du), len); }
                            else { sum_d += (u64)du * len; sum_d2 += (u64)du * du * len; }
                            if (d > 0) cov_w += len;
                            mn = min(mn, du); mx = max(mx, du);
                            if (WANT_HIST) hist_add(du, len);
                        }
                    }
                }
                sum_d += s1t; sum_d2 += s2t;
            } else {
                // ---- dense path: one wave scan per row of 64 bases
                int carry = 0;
#pragma unroll
                for (int j = 0; j < ROWS; j++) {
                    const int inc = wave_incl_scan(x[j]);
                    const int d = carry + inc;
                    carry += __builtin_amdgcn_readlane(inc, 63);
                    const u32 p = lo + (u32)(j * 64 + lane);
                    const u32 du = (u32)d;
                    if (p < L) {
                        cov_f += d > 0;
                        if (WRITE_DEPTH) a.depth_out[dbase + p] = d;
                        if (win_any && p >= wst && p < wet) {
                            sum_d += du;
                            sum_d2 += (u64)du * du;
                            cov_w += d > 0;
                            mn = min(mn, du); mx = max(mx, du);
                            if (WANT_HIST) {
                                const u32 rel = p - wst;
                                if (x[j] != 0 && rel) { hist_add((u32)(d - x[j]), rel); hist_add(du, 0u - rel); }
                                if (p == wet - 1) hist_add(du, wet - wst);
                            }
                        }
                    }
                }
            }
        }
        flush();
        cur_c = -1;
        ch = ch_next;
    }
}

constexpr size_t pileup_stream_smem_bytes() { return (size_t)4 * ((size_t)STREAM_TW * 4 + STREAM_HB * 4); }

// ------------------------------------------------------------------------------------ k_pileup_fast
// Default pileup.  Same tile / candidate-range / run-word contract as k_pileup_stream (one wave per 1024-base tile,
// chunks of consecutive tiles, sums kept in registers across the tiles of a contig), different arithmetic:
//
//   * two u16 count tables per wave in LDS, S[p] = runs starting at p (clipped to the tile start: the carry-in),
//     E[p] = runs ending at p.  2 x 2 KiB; one ds_add_u32 per event; a tile with >= 32768 candidates (a u16 could
//     wrap) is left to k_pileup_stream through the slow-tile list.
//   * BLOCKED ownership: lane l owns positions [16 l, 16 l + 16) = 32 contiguous bytes of each table, read with two
//     ds_read_b128.  Depth is then a lane-serial running sum over 16 packed i16 deltas (v_pk_sub_i16 of the two
//     tables) started from ONE wave prefix sum of the per-lane net deltas: one cross-lane scan per tile instead of
//     one per 64 bases, no ballot / mbcnt compaction, no second LDS round trip.  Per tile the only LDS dependency is
//     zero -> scatter -> read (LDS serves one wave's requests in order, so no s_waitcnt between them).
//   * the next tile's descriptor (scalar loads) and run words (two 8-byte vector loads) are requested before the
//     statistics of the current tile, so HBM latency overlaps the VALU work of the same wave.
//   * interior tiles (wholly inside the contig and its end-exclusion window) with < 512 candidates take a loop with
//     32-bit per-tile partial sums and no bounds tests; everything else takes the general loop (window / contig-end
//     masks, 64-bit sums, histogram overflow to the arena).
//   * histogram: one LDS atomic per constant-depth segment of a lane's 16 positions (depth changes only where a delta
//     is non-zero).  (Copies of every bin in adjacent words, lane l adding to copy l mod 2 | 4, were measured: 0.62 -> 0.72 / 0.85 ms,
//     profiles/r04_pileup_hrep.log — what the conflicts cost is less than what the lost occupancy does.)
//   * (Packed 16-bit arithmetic — a lane's dword k holding positions k and k + 8, v_pk_add / v_dot2_u32_u16 / v_pk_min / v_pk_max walking both
//     halves at once, 16 + 8 + 8 + 32 instructions instead of ~130 — was built and measured: 0.683 ms against 0.623.  A VOP3P instruction
//     issues in 4.2 cycles per wave where the plain two-operand forms it replaces take 2.4 (profiles/r03_valu_rate.json), so two values per
//     instruction buy nothing here, and the permuted table layout costs two more address operations per event.  profiles/r04_pileup_packed.log.)
constexpr int FAST_TW = 1024, FAST_HB = 512;
constexpr int FAST_HB7 = 384;   // k_pileup_fast2t: seven workgroups per CU (22 KiB of LDS each, 72 registers)
constexpr size_t pileup_fast_smem_bytes(bool hist, int hb = FAST_HB, int tables = 2) { return (size_t)4 * ((size_t)FAST_TW * 2 * tables + (hist ? (size_t)hb * 4 : 0)); }

typedef short v2i16 __attribute__((ext_vector_type(2)));
// LDS by byte address (the 32-bit offset inside the workgroup's allocation): the histogram loops keep "address of the bin of the running depth"
// as their running value — base + 4 x depth, advanced by 4 x delta — so that an atomic needs no address arithmetic of its own
typedef __attribute__((address_space(3))) u32 lds_u32_t;
__device__ __forceinline__ u32 lds_addr_of(const u32 *p) { return (u32)(uintptr_t)(lds_u32_t *)p; }
__device__ __forceinline__ void lds_atomic_add(u32 addr, u32 x) { (void)__hip_atomic_fetch_add((lds_u32_t *)(uintptr_t)addr, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// TABLES = 2: the two count tables described above (rounds 3-5).  TABLES = 1 (round 6, the default): ONE table of 16-bit DELTAS, every
// field biased by 0x8000 — a start adds 1 to its field, an end subtracts 1 (ds_add_u32 / ds_sub_u32 of 1 or 0x10000: a biased field never
// reaches 0 or 0xffff with fewer than 32768 candidates, so nothing carries into or borrows from the neighbouring field) —: half the bytes
// to preset and to read back per tile (two ds_write_b128 + two ds_read_b128 instead of four each), an xor with 0x80008000 instead of the
// packed subtraction of the two tables, the same events.
template <bool WANT_HIST, int HB, int TABLES>
__device__ __forceinline__ void pileup_fast_body(const PileupArgs &a, u32 n_tiles, u32 chunk_tiles) {
    constexpr int TW = FAST_TW, HBW = HB;
    constexpr int DS = TABLES == 1 ? 2 : 0;      // the one-table kernel counts in units of 1/4: an event adds or subtracts 4, so the running depth IS the byte offset of its bin
    constexpr size_t WB = (size_t)TW * 2 * TABLES + (WANT_HIST ? (size_t)HBW * 4 : 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 *S = reinterpret_cast<u32 *>(smem + (size_t)w * WB);   // 512 dwords = 1024 u16
    u32 *E = TABLES == 2 ? S + TW / 2 : S;
    u32 *lhist = S + (TW / 2) * TABLES;
    uint4 *S4 = reinterpret_cast<uint4 *>(S), *E4 = reinterpret_cast<uint4 *>(E);
    const u32 wave_id = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (u32)w)), n_waves = gridDim.x * 4u;
    if (WANT_HIST) {
#pragma unroll
        for (int b = lane; b < HBW; b += 64) lhist[b] = 0u;
    }
    u64 sum_d = 0, sum_d2 = 0, proc_win = 0;
    u32 cov_w = 0, cov_f = 0, mn = 0xffffffffu, mx = 0;
    int cur_c = -1;
    u64 hoff = 0; u32 hcap = 0;
    const u64 excl = a.excl;
    // With the histogram wanted, interior tiles do NOT accumulate sum d, sum d^2, covered, min and max per position: every position's
    // depth goes into the wave's LDS histogram anyway, and the five are read off its bins when they are moved to the arena (`drain`).
    // The general loop (contig ends, tiles deeper than the bins) counts per position as before, so the bins hold segments of ONE kind at a
    // time — lh_explicit says which — and are drained when the kind changes.  All wave-uniform.
    bool lh_explicit = false;
    u32 hbound = 0;            // no LDS bin above this was touched since the last drain (depth <= candidate runs of the tile)
    auto drain = [&](bool derive) {
        lds_fence();
        for (u32 b = (u32)lane; b <= hbound; b += 64) {
            const u32 x = lhist[b];
            if (x) {
                atomicAdd(&a.hist_arena[hoff + b], x); lhist[b] = 0u;
                if (derive) {
                    const u32 cv = b ? x : 0u;
                    sum_d += (u64)b * x; sum_d2 += (u64)(b * b) * x; cov_w += cv; cov_f += cv;
                    mn = min(mn, b); mx = max(mx, b);
                }
            }
        }
        lds_fence();
        hbound = 0;
    };

    auto hist_add = [&](u32 d, u32 x) {
        if (__builtin_expect(d < (u32)HBW, 1)) atomicAdd(&lhist[d], x);
        else hist_add_overflow(a.hist_arena, hoff, hcap, a.g, d, x);
    };
    auto flush = [&]() {
        if (cur_c >= 0) {
            if (WANT_HIST && proc_win) drain(!lh_explicit);
            // (DPP sums: a contig of an assembly is a tile or two, so this runs as often as the tile loop does)
            const bool narrow = !__any((sum_d | sum_d2) >> 56);
            const u64 s1 = narrow ? wave_sum_u56(sum_d) : wave_sum_u64(sum_d), s2 = narrow ? wave_sum_u56(sum_d2) : wave_sum_u64(sum_d2);
            const u32 c1 = wave_sum_u32_dpp(cov_w), c2 = wave_sum_u32_dpp(cov_f);
            const u32 m1 = wave_min_u32_dpp(mn), m2 = wave_max_u32_dpp(mx);
            DevContig *C = &a.ctg[cur_c];
            if (lane == 0) {
                if (s1) atomicAdd(&C->sum_d, s1);
                if (s2) atomicAdd(&C->sum_d2, s2);
                if (c1) atomicAdd(&C->cov_win, (u64)c1);
                if (c2) atomicAdd(&C->cov_full, (u64)c2);
                if (proc_win) {
                    atomicAdd(&C->proc_win, proc_win);
                    atomicMin(&C->min_d, m1);
                    atomicMax(&C->max_d, m2);
                }
            }
        }
        sum_d = sum_d2 = 0; proc_win = 0; cov_w = cov_f = 0; mn = 0xffffffffu; mx = 0; hbound = 0; lh_explicit = false;
    };
    auto load_runs = [&](const uint4 &d, uint2 &r0, uint2 &r1) {
        const u32 i0 = d.x + (u32)lane, i1 = i0 + 64u;
        r0 = i0 < d.y ? a.runs[i0] : make_uint2(0u, 0u);
        r1 = i1 < d.y ? a.runs[i1] : make_uint2(0u, 0u);
    };
    // tile sequence of this wave: chunks wave_id, wave_id + n_waves, ... of chunk_tiles consecutive tiles each
    const u32 n_chunks = (n_tiles + chunk_tiles - 1) / chunk_tiles;
    if (wave_id >= n_chunks) return;
    u32 ch = wave_id;
    u32 t = ch * chunk_tiles, t_end = min(t + chunk_tiles, n_tiles);
    uint4 ds = a.desc[2 * (size_t)(a.tile_base + t)], dC1 = a.desc[2 * (size_t)(a.tile_base + t) + 1];
    uint2 rw0, rw1;
    load_runs(ds, rw0, rw1);
    for (;;) {
        // ---- next tile of the sequence (wave-uniform), its descriptor requested now
        u32 tn = t + 1, tn_end = t_end, chn = ch;
        bool chunk_end = false;
        if (tn >= t_end) {
            chunk_end = true;
            chn = ch + n_waves;
            tn = chn < n_chunks ? chn * chunk_tiles : 0xffffffffu;
            tn_end = chn < n_chunks ? min(tn + chunk_tiles, n_tiles) : 0u;
        }
        const bool have_next = tn != 0xffffffffu;
        uint4 nds = make_uint4(0u, 0u, 0u, 0u), ndC1 = make_uint4(0u, 0u, 0u, 0u);
        if (have_next) { nds = a.desc[2 * (size_t)(a.tile_base + tn)]; ndC1 = a.desc[2 * (size_t)(a.tile_base + tn) + 1]; }

        const u32 c = dC1.x, lo = dC1.y, L = ds.z, cxo = dC1.z, cxn = dC1.w;
        const bool live = (ds.x < ds.y || cxn != 0u) && !(ds.w & TILE_F_SLOW);
        uint4 sv0, sv1, ev0, ev1;
        if (live) {
            if ((int)c != cur_c) {
                flush();
                cur_c = (int)c;
                if (WANT_HIST) { hoff = a.ctg[c].hist_off; hcap = a.ctg[c].hist_cap; }
            }
            // ---- zero, scatter, read back: three LDS phases of one wave, served in order
            const u32 zb = TABLES == 2 ? 0u : 0x80008000u;
            const uint4 z4 = make_uint4(zb, zb, zb, zb);
            S4[2 * lane] = z4; S4[2 * lane + 1] = z4;
            if (TABLES == 2) { E4[2 * lane] = z4; E4[2 * lane + 1] = z4; }
            asm volatile("" ::: "memory");
            const u32 hi = lo + TW;
            auto add_run = [&](u32 s, u32 e) {
                if (s < hi && e > lo) {
                    const u32 s0 = s > lo ? s - lo : 0u;
                    atomicAdd(&S[s0 >> 1], (1u << DS) << ((s0 & 1u) << 4));
                    if (e < hi) {
                        const u32 e0 = e - lo;
                        if (TABLES == 2) atomicAdd(&E[e0 >> 1], 1u << ((e0 & 1u) << 4)); else atomicSub(&E[e0 >> 1], (1u << DS) << ((e0 & 1u) << 4));
                    }
                }
            };
            auto apply = [&](const uint2 rw, u32 i) {
                if (rw.y == 0u) return;
                const u32 type = rw.y >> 30;
                if (type <= RW_DOUBLE) {
                    const bool dbl = type == RW_DOUBLE;
                    const u32 l1 = dbl ? (rw.y & 1023u) : rw.y;
                    add_run(rw.x, rw.x + l1);
                    if (dbl) {
                        const u32 s2 = rw.x + l1 + ((rw.y >> 10) & 255u);
                        add_run(s2, s2 + ((rw.y >> 18) & 1023u));
                    }
                } else if (type == RW_COMPLEX) {
                    u32 cursor = (u32)a.r.pos[i];
                    const u32 c0 = a.r.cigar_off[i], c1 = a.r.cigar_off[i + 1];
                    for (u32 k = c0; k < c1; k++) {
                        const u32 wd = a.r.cigar[k];
                        const u32 op = wd & 15u, len = wd >> 4;
                        if (op == 0u || op == 7u || op == 8u) { add_run(cursor, cursor + len); cursor += len; }
                        else if (op == 2u || op == 3u) cursor += len;
                    }
                }   // RW_BUCKET: delivered through the tile's bucket
            };
            apply(rw0, ds.x + (u32)lane);
            if (ds.y - ds.x > 64u) {     // wave-uniform: more than half of the tiles of a 7x-deep sample stop here
                apply(rw1, ds.x + 64u + (u32)lane);
                for (u32 i = ds.x + 128u + (u32)lane; i < ds.y; i += 64) apply(a.runs[i], i);   // deep tiles only
            }
            for (u32 j = (u32)lane; j < cxn; j += 64) { const uint2 q = a.cx_runs[(u64)cxo + j]; add_run(q.x, q.y); }   // long reads
            asm volatile("" ::: "memory");
            sv0 = S4[2 * lane]; sv1 = S4[2 * lane + 1];
            if (TABLES == 2) { ev0 = E4[2 * lane]; ev1 = E4[2 * lane + 1]; }
        }
        // ---- next tile's run words: in flight during this tile's statistics
        uint2 nrw0 = make_uint2(0u, 0u), nrw1 = make_uint2(0u, 0u);
        if (have_next) load_runs(nds, nrw0, nrw1);

        if (live) {
            // packed i16 deltas of the 16 positions this lane owns
            const u32 sw[8] = {sv0.x, sv0.y, sv0.z, sv0.w, sv1.x, sv1.y, sv1.z, sv1.w};
            u32 ew[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            if (TABLES == 2) { ew[0] = ev0.x; ew[1] = ev0.y; ew[2] = ev0.z; ew[3] = ev0.w; ew[4] = ev1.x; ew[5] = ev1.y; ew[6] = ev1.z; ew[7] = ev1.w; }
            v2i16 dl[8];
#pragma unroll
            for (int k = 0; k < 8; k++) dl[k] = TABLES == 2 ? __builtin_bit_cast(v2i16, sw[k]) - __builtin_bit_cast(v2i16, ew[k]) : __builtin_bit_cast(v2i16, sw[k] ^ 0x80008000u);
            int net = 0;
            const v2i16 one2 = {1, 1};
#pragma unroll
            for (int k = 0; k < 8; k++) net = __builtin_amdgcn_sdot2(dl[k], one2, net, false);   // sum of 16 i16 in 32 bits
            int d = wave_incl_scan(net) - net;       // depth (<< DS) just left of this lane's first position

            const bool has_win = 2 * excl < (u64)L;
            const u32 tlen_t = min((u32)TW, L - lo);
            const u32 ws = has_win ? (u32)excl : 0u, we = has_win ? (u32)(L - excl) : 0u;
            const u32 wst = max(ws, lo), wet = min(we, lo + tlen_t);
            const bool win_any = has_win && wst < wet;
            if (win_any) proc_win += (u64)(wet - wst);
            const bool interior = has_win && lo >= ws && lo + (u32)TW <= we;
            const u32 cand = ds.y - ds.x + cxn;
            const bool fast_tile = interior && cand < (u32)HBW;
            const bool derive_tile = fast_tile || cand < (u32)HBW;      // (with the histogram wanted) the tile's statistics come off the bins
            if (WANT_HIST) {
                if (lh_explicit == derive_tile) { if (proc_win) drain(!lh_explicit); lh_explicit = !derive_tile; }   // the bins change kind
                hbound = max(hbound, min(cand, (u32)HBW - 1u));
            }
            if (WANT_HIST && fast_tile) {
                // ---- fast loop, histogram wanted: every position is inside the window and depth < 512 = the LDS bins: one atomic per
                // constant-depth segment of the lane's 16 positions and nothing else
                u32 seg0 = 0;
                u32 da = lds_addr_of(lhist) + ((u32)d << (2 - DS));      // address of bin d
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int dj = (j & 1) ? (int)dl[j >> 1].y : (int)dl[j >> 1].x;
                    if (j > 0 && dj != 0) { lds_atomic_add(da, (u32)j - seg0); seg0 = (u32)j; }
                    da += (u32)(dj << (2 - DS));
                }
                lds_atomic_add(da, 16u - seg0);
            } else if (WANT_HIST && cand < (u32)HBW) {
                // ---- a tile at an end of its contig (or of the end-exclusion window), shallow enough for the LDS bins: the same loop with every
                // segment clipped to the lane's part [wa, wb) of the window, so that the bins still hold window positions only and the window
                // statistics still come off them (`drain`); the covered positions of the contig OUTSIDE the window — the full-length covered
                // count has no end exclusion (estimators.rs:467-502) — are counted from a 16-bit mask of the lane's covered positions.
                // (Until round 6 such tiles took the general loop below: ~20 instructions per position where this takes ~8; an assembly
                // of short contigs has no other tiles.)
                const u32 p0 = lo + 16u * (u32)lane;
                const u32 wa = win_any ? min(max(wst, p0) - p0, 16u) : 0u, wb = win_any ? min(max(wet, p0) - p0, 16u) : 0u;
                const u32 le = min(max(L, p0) - p0, 16u);          // this lane's positions inside the contig
                u32 seg0 = 0, covm = 0;
                const u32 da0 = lds_addr_of(lhist);
                u32 da = da0 + ((u32)d << (2 - DS));      // address of bin d
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int dj = (j & 1) ? (int)dl[j >> 1].y : (int)dl[j >> 1].x;
                    if (j > 0 && dj != 0) {
                        const u32 s_ = max(seg0, wa), e_ = min((u32)j, wb);
                        if (e_ > s_) lds_atomic_add(da, e_ - s_);
                        seg0 = (u32)j;
                    }
                    da += (u32)(dj << (2 - DS));
                    covm = covm + covm + (da != da0 ? 1u : 0u);       // bit 15 - j: position j is covered
                }
                {
                    const u32 s_ = max(seg0, wa);
                    if (wb > s_) lds_atomic_add(da, wb - s_);
                }
                // positions [0, le) without [wa, wb), as bits 15 - j
                const u32 in_ctg = 0xffffu & ~(0xffffu >> le), in_win = (0xffffu >> wa) & ~(0xffffu >> wb);
                cov_f += (u32)__popc(covm & in_ctg & ~in_win);
            } else if (fast_tile) {
                // ---- fast loop, no histogram: every position is inside the window; depth < 512 so 24-bit multiplies and 32-bit
                // per-tile sums are exact (16 positions x 2^18 per lane)
                u32 s1t = 0, s2t = 0, cv = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int dj = (j & 1) ? (int)dl[j >> 1].y : (int)dl[j >> 1].x;
                    d += dj;
                    const u32 du = (u32)d >> DS;
                    s1t += du;
                    s2t = __umul24(du, du) + s2t;
                    cv += du != 0u ? 1u : 0u;
                    mn = min(mn, du); mx = max(mx, du);
                }
                sum_d += s1t; sum_d2 += s2t; cov_w += cv; cov_f += cv;
            } else {
                // ---- general loop: window and contig-end tests per position, 64-bit sums, histogram overflow
                const u32 p0 = lo + 16u * (u32)lane;
                u32 seg0 = 0;          // first window position (lane-relative) of the open constant-depth segment
                bool seg_open = false;
#pragma unroll 4
                for (int j = 0; j < 16; j++) {
                    const int dj = (j & 1) ? (int)dl[j >> 1].y : (int)dl[j >> 1].x;
                    const u32 p = p0 + (u32)j;
                    const bool in_w = win_any && p >= wst && p < wet;
                    if (WANT_HIST && seg_open && (dj != 0 || !in_w)) { hist_add((u32)d >> DS, (u32)j - seg0); seg_open = false; }
                    d += dj;
                    const u32 du = (u32)d >> DS;
                    if (p < L) cov_f += du != 0u ? 1u : 0u;
                    if (in_w) {
                        sum_d += du; sum_d2 += (u64)du * du;
                        cov_w += du != 0u ? 1u : 0u;
                        mn = min(mn, du); mx = max(mx, du);
                        if (WANT_HIST && !seg_open) { seg_open = true; seg0 = (u32)j; }
                    }
                }
                if (WANT_HIST && seg_open) hist_add((u32)d >> DS, 16u - seg0);
            }
        }
        if (chunk_end) { flush(); cur_c = -1; }
        if (!have_next) break;
        t = tn; t_end = tn_end; ch = chn; ds = nds; dC1 = ndC1; rw0 = nrw0; rw1 = nrw1;
    }
}

// The default (round 6): ONE table of biased deltas, 512 LDS bins (16 KiB of LDS per workgroup), seven waves per SIMD: 0.475 ms at BASELINE
// config 2 against 0.531 for the two-table kernel below, alternating runs on one box (profiles/r06_lean_onetable_ab_5k.log; 384 bins 0.481,
// eight waves with 28 bytes of scratch 0.521).
template <bool WANT_HIST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_pileup_fast(PileupArgs a, u32 n_tiles, u32 chunk_tiles) {
    pileup_fast_body<WANT_HIST, FAST_HB, 1>(a, n_tiles, chunk_tiles);
}
// The second implementation (COVERM_FAST_TABLES=2; rounds 4-5's default): two u16 count tables, 384 bins; six waves per SIMD (with the clipped loop for contig ends the body no longer fits 72 registers without scratch).
template <bool WANT_HIST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_pileup_fast2t(PileupArgs a, u32 n_tiles, u32 chunk_tiles) {
    pileup_fast_body<WANT_HIST, FAST_HB7, 2>(a, n_tiles, chunk_tiles);
}

// (The same step once more — 256 bins, 64 registers with 12 dwords of scratch, eight waves per SIMD — was measured in round 5: 0.816 ms
// against 0.535, alternating runs on one box, profiles/r05_prep_pileup_ab.log: the stripped loop then only takes tiles with fewer than 256
// candidate runs and the scratch traffic sits in the per-position loop.  Removed.)

// ------------------------------------------------------------------------------------ k_estimate
// CoverageEstimator::calculate_coverage (estimators.rs:530-839) on the device, for `coverm contig`: one entry = one contig, the
// unobserved lengths are [0] (contig.rs:62-66), so an estimator sees exactly one add_contig (estimators.rs:366-528).  One WAVE per
// contig.  Everything but the trimmed mean is a handful of f32 operations of lane 0 — the reference's own expressions in the reference's
// order, every operation rounded once (no contraction into fused multiply-adds: the host evaluates them with separate operations), so
// the floats are the host path's bit for bit (tests/test_gpu_estimates.py).  The trimmed mean walks the depth histogram; its loop
// (estimators.rs:596-640) is restated over the inclusive prefix sums of the bins, which the wave builds 64 bins at a time:
//     s = first bin whose prefix reaches min_index:  total  = (min(prefix_s, max_index) - min_index + 1) * s   [the reference's two branches]
//     the bins behind s while their prefix stays <= max_index:  total += count_i * i
//     e = first bin behind s whose prefix exceeds max_index:    total += (max_index >= prefix_{e-1} ? max_index - prefix_{e-1} + 1 : 0) * e
// — integers throughout, so the order of the additions does not matter.  TPM (f64 exp / ln of the host's libm) and the coverage
// histogram (it prints the histogram itself) stay on the host: cov_set_estimators refuses them.
struct DevEstimator { int kind; float min_frac; u64 excl; int exclude_mismatches; float trim_min, trim_max; u32 pad; };
static_assert(sizeof(DevEstimator) == 32, "DevEstimator mirrors cov_estimator");
constexpr u32 EST_MAX = 16;
struct EstParams { DevEstimator e[EST_MAX]; u32 n; u32 pad; };
enum { EST_MEAN = 0, EST_TRIMMED_MEAN = 1, EST_PILEUP_COUNTS = 2, EST_COVERED_FRACTION = 3, EST_COVERED_BASES = 4, EST_RPKM = 5, EST_TPM = 6,
       EST_VARIANCE = 7, EST_LENGTH = 8, EST_READ_COUNT = 9, EST_READS_PER_BASE = 10, EST_ANIR = 11 };

__device__ __forceinline__ float est_f32(u64 x) { return __ull2float_rn(x); }          // Rust `x as f32`: round to nearest even
__device__ __forceinline__ u64 est_f32_to_usize(float x) {                             // Rust `x as usize`: saturating, NaN -> 0
    if (!(x == x) || x <= 0.0f) return 0ull;
    if (x >= 18446744073709551616.0f) return ~0ull;
    return (u64)x;
}
__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = __shfl_up(v, o);
        if (lane_id() >= o) v += t;
    }
    return v;
}

// The trimmed mean's total (estimators.rs:596-640, restated over prefix sums as described above) by the WHOLE wave for one contig: 64 bins per
// step.  Every argument is wave-uniform; every lane returns the total.
__device__ __forceinline__ u64 trimmed_total_wave(const u32 *__restrict__ bins, u32 nh, u64 bin0_extra, u64 min_index, u64 max_index) {
    const int lane = lane_id();
    u64 total = 0, carry = 0;        // carry = prefix of the bins in front of this batch (wave-uniform)
    int state = 0;                   // 0: before s, 1: between s and e, 2: done
    for (u32 b = 0; b < nh && state < 2; b += 64) {
        const u32 i = b + (u32)lane;
        u64 n = i < nh ? (u64)bins[i] : 0ull;
        if (i == 0) n += bin0_extra;
        const u64 acc = carry + wave_incl_scan_u64(n);
        const bool valid = i < nh;
        int first = 0;                                   // first lane of this batch that still adds n * i
        if (state == 0) {
            const u64 sm = __ballot(valid && acc >= min_index);
            if (sm == 0) { carry = bcast_u64(acc, 63); continue; }
            const int ls = __ffsll((long long)sm) - 1;
            const u64 acc_s = bcast_u64(acc, ls);
            total = (acc_s > max_index ? max_index - min_index + 1 : acc_s - min_index + 1) * (u64)(b + (u32)ls);
            state = 1; first = ls + 1;
        }
        // state 1: the bins from `first` on add n * i until the first one whose prefix exceeds max_index
        const u64 em = __ballot(valid && lane >= first && acc > max_index);
        const int le = em ? __ffsll((long long)em) - 1 : 64;
        u64 part = (valid && lane >= first && lane < le) ? n * (u64)i : 0ull;
        if (em && lane == le) {
            const u64 excess = acc - n;
            part = (max_index >= excess ? max_index - excess + 1 : 0ull) * (u64)i;
        }
        total += wave_sum_u64(part);
        if (em) state = 2;
        carry = bcast_u64(acc, 63);
    }
    return total;
}

// LANES = false: one WAVE per contig (lane 0 evaluates, the wave walks the histogram); LANES = true: one LANE per contig — for assemblies
// (10^5 - 10^7 contigs, a few dozen bins each: a wave per contig is 2 M nearly idle waves, 1.46 ms at 2 M contigs) — where a lane walks
// its own contig's bins one by one (the same integers in the same order of bins; integer sums, so the floats are the same bits) and the few
// contigs with more than EST_SERIAL_BINS bins are walked by the whole wave, one after the other.
constexpr u32 EST_SERIAL_BINS = 96;
template <bool LANES>
__device__ __forceinline__ void estimate_body(const DevContig *__restrict__ ctg, u32 n_targets, const u32 *__restrict__ tlen, u64 excl,
                                              const u32 *__restrict__ arena, const EstParams &P, float *__restrict__ out) {
#pragma clang fp contract(off)
    const int lane = lane_id();
    const u32 c_raw = LANES ? blockIdx.x * 256u + threadIdx.x : blockIdx.x * 4u + (threadIdx.x >> 6);
    if (!LANES && c_raw >= n_targets) return;
    const bool in = c_raw < n_targets;
    const u32 c = in ? c_raw : n_targets - 1u;        // (LANES: lanes behind the last contig walk along, write nothing)
    const DevContig *C = &ctg[c];
    float *o = out + (size_t)c * P.n;
    const bool touched = C->n_pass != 0;
    if (!LANES && !touched) { if ((u32)lane < P.n) o[lane] = 0.0f; return; }     // (the host prints such a contig through print_zero_coverage)
    // the integer statistics as convert_results + EntryAcc::add_contig would hand them to calculate (host_coverage.cpp)
    const u64 L = tlen[c];
    const bool has_win = 2 * excl < L;
    const u64 win_len = has_win ? L - 2 * excl : 0;
    const u64 win_sum_d = has_win ? C->sum_d : 0, win_sum_d2 = has_win ? C->sum_d2 : 0, win_covered = has_win ? C->cov_win : 0;
    const u64 win_min_d = has_win ? ((C->proc_win < win_len || C->min_d == 0xffffffffu) ? 0u : C->min_d) : 0xffffffffu;
    const u64 full_len = L, full_covered = C->cov_full, n_reads = C->n_primary, mismatches = C->sum_nm - C->sum_indel;
    const u32 nh = has_win ? C->max_d + 1u : 0u;          // = the compact histogram's length (k_hist_off<1>; no target mask here)
    const u64 bin0_extra = win_len - C->proc_win;       // window positions of tiles no record touched: depth 0 (k_hist_compact adds the same)
    const u32 *bins = arena + C->hist_off;
    for (u32 k = 0; k < P.n; k++) {
        const DevEstimator e = P.e[k];
        float r = 0.0f;
        switch (e.kind) {
        case EST_MEAN: {
            const u64 T = win_len;
            if (T == 0 || (est_f32(win_covered) / est_f32(T)) < e.min_frac) break;
            const float num = e.exclude_mismatches ? est_f32(win_sum_d - mismatches) : est_f32(win_sum_d);
            r = num / est_f32(T);
            break;
        }
        case EST_TRIMMED_MEAN: {
            const u64 T = win_len;
            const bool walk = (!LANES || (in && touched)) && T != 0 && !((est_f32(win_covered) / est_f32(T)) < e.min_frac) && win_covered != 0;
            const u64 min_index = est_f32_to_usize(__builtin_floorf(e.trim_min * est_f32(T)));
            const u64 max_index = est_f32_to_usize(__builtin_ceilf(e.trim_max * est_f32(T)));
            u64 total = 0;
            if (!LANES) {
                if (!walk) break;
                total = trimmed_total_wave(bins, nh, bin0_extra, min_index, max_index);
            } else {
                // a lane's own walk, bin by bin (the reference's loop over the prefix sums, as in trimmed_total_wave)
                const bool serial = walk && nh <= EST_SERIAL_BINS;
                u64 acc = 0;
                int state = serial ? 0 : 2;
                for (u32 i = 0; __any(state < 2 && i < nh); i++) {
                    if (state < 2 && i < nh) {
                        u64 n = bins[i];
                        if (i == 0) n += bin0_extra;
                        acc += n;
                        if (state == 0) {
                            if (acc >= min_index) { total = (acc > max_index ? max_index - min_index + 1 : acc - min_index + 1) * (u64)i; state = 1; }
                        } else if (acc > max_index) {
                            const u64 excess = acc - n;
                            total += (max_index >= excess ? max_index - excess + 1 : 0ull) * (u64)i;
                            state = 2;
                        } else total += n * (u64)i;
                    }
                }
                for (u64 bm = __ballot(walk && !serial); bm != 0; bm &= bm - 1) {      // deep contigs: the whole wave on each
                    const int l = __builtin_ctzll(bm);
                    const u64 bp = bcast_u64((u64)(uintptr_t)bins, l);
                    const u64 t = trimmed_total_wave(reinterpret_cast<const u32 *>((uintptr_t)bp), __builtin_amdgcn_readlane(nh, l), bcast_u64(bin0_extra, l),
                                                     bcast_u64(min_index, l), bcast_u64(max_index, l));
                    if (lane == l) total = t;
                }
                if (!walk) break;
            }
            r = est_f32(total) / est_f32(max_index - min_index);
            break;
        }
        case EST_COVERED_FRACTION: {
            const u64 T = full_len;
            if (T == 0 || (est_f32(full_covered) / est_f32(T)) < e.min_frac) break;
            r = est_f32(full_covered) / est_f32(T);
            break;
        }
        case EST_COVERED_BASES: {
            const u64 T = full_len;
            if (T == 0 || (est_f32(full_covered) / est_f32(T)) < e.min_frac) break;
            r = est_f32(full_covered);
            break;
        }
        case EST_RPKM: {
            const u64 T = full_len;
            if (T == 0 || (est_f32(full_covered) / est_f32(T)) < e.min_frac) break;
            r = est_f32(n_reads * 1000000000ull) / est_f32(T);
            break;
        }
        case EST_VARIANCE: {
            const u64 T = win_len;
            if (T == 0) break;
            if ((est_f32(win_covered) / est_f32(T)) < e.min_frac || T < 3 || win_len == 0) break;
            const u64 kk = win_min_d, N = win_len;
            const u64 ex = win_sum_d - kk * N;
            const u64 ex2 = win_sum_d2 - 2 * kk * win_sum_d + kk * kk * N;
            r = (est_f32(ex2) - est_f32(ex * ex) / est_f32(T)) / est_f32(T - 1);
            break;
        }
        case EST_LENGTH: r = est_f32(full_len); break;
        case EST_READ_COUNT: r = est_f32(n_reads); break;
        case EST_READS_PER_BASE: r = est_f32(n_reads) / est_f32(full_len); break;
        case EST_ANIR: r = n_reads == 0 ? 0.0f : (float)(C->id_primary / (double)n_reads); break;
        default: break;
        }
        if (LANES) { if (in) o[k] = touched ? r : 0.0f; }
        else if (lane == 0) o[k] = r;
    }
}
__global__ __launch_bounds__(256) void k_estimate(const DevContig *__restrict__ ctg, u32 n_targets, const u32 *__restrict__ tlen, u64 excl,
                                                  const u32 *__restrict__ arena, EstParams P, float *__restrict__ out) {
    estimate_body<false>(ctg, n_targets, tlen, excl, arena, P, out);
}
__global__ __launch_bounds__(256) void k_estimate_lanes(const DevContig *__restrict__ ctg, u32 n_targets, const u32 *__restrict__ tlen, u64 excl,
                                                        const u32 *__restrict__ arena, EstParams P, float *__restrict__ out) {
    estimate_body<true>(ctg, n_targets, tlen, excl, arena, P, out);
}

// ------------------------------------------------------------------------------------ interval statistics
// Per-interval (per-gene, genes.rs:508-535) statistics over a materialised depth arena: one wave per interval.
// The window is the interval shrunk by `excl` at both of ITS ends (the reference hands the gene's own delta array to
// add_contig); covered bases without exclusion.  k_interval_hist fills each interval's histogram slice afterwards.
struct DevInterval { u32 tid, pad; u64 start, end; };
struct DevIntervalStats { u64 win_sum_d, win_sum_d2, win_covered, full_covered; u32 win_min_d, win_max_d, hist_len, pad; u64 hist_off; };

__global__ __launch_bounds__(256) void k_interval_stats(const int32_t *__restrict__ depth, const u64 *__restrict__ depth_off,
                                                        const DevInterval *__restrict__ iv, u64 n, u64 excl,
                                                        DevIntervalStats *__restrict__ out) {
    const int lane = lane_id();
    const u64 wave = (u64)blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = (u64)gridDim.x * 4u;
    for (u64 g = wave; g < n; g += n_waves) {
        const DevInterval I = iv[g];
        const int32_t *d = depth + depth_off[I.tid];
        const u64 len = I.end - I.start;
        const bool has_win = 2 * excl < len;
        const u64 ws = I.start + excl, we = has_win ? I.end - excl : ws;
        u64 s1 = 0, s2 = 0;
        u32 cw = 0, cf = 0, mn = 0xffffffffu, mx = 0;
        for (u64 p = I.start + (u64)lane; p < I.end; p += 64) {
            const u32 v = (u32)d[p];
            cf += v > 0u;
            if (has_win && p >= ws && p < we) { s1 += v; s2 += (u64)v * v; cw += v > 0u; mn = min(mn, v); mx = max(mx, v); }
        }
        s1 = wave_sum_u64(s1); s2 = wave_sum_u64(s2);
        cw = wave_sum_u32(cw); cf = wave_sum_u32(cf); mn = wave_min_u32(mn); mx = wave_max_u32(mx);
        if (lane == 0) {
            DevIntervalStats o;
            o.win_sum_d = s1; o.win_sum_d2 = s2; o.win_covered = cw; o.full_covered = cf;
            o.win_min_d = has_win ? mn : 0u; o.win_max_d = mx; o.hist_len = has_win ? mx + 1u : 0u; o.pad = 0; o.hist_off = 0;
            out[g] = o;
        }
    }
}

constexpr int IV_HB = 512;
__global__ __launch_bounds__(256) void k_interval_hist(const int32_t *__restrict__ depth, const u64 *__restrict__ depth_off,
                                                       const DevInterval *__restrict__ iv, u64 n, u64 excl,
                                                       const DevIntervalStats *__restrict__ st, unsigned long long *__restrict__ hist) {
    __shared__ u32 lh[4][IV_HB];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const u64 wave = (u64)blockIdx.x * 4u + (u64)w, n_waves = (u64)gridDim.x * 4u;
    for (int b = lane; b < IV_HB; b += 64) lh[w][b] = 0u;
    for (u64 g = wave; g < n; g += n_waves) {
        const DevInterval I = iv[g];
        const DevIntervalStats S = st[g];
        if (S.hist_len == 0u) continue;
        const int32_t *d = depth + depth_off[I.tid];
        unsigned long long *h = hist + S.hist_off;
        for (u64 p = I.start + excl + (u64)lane; p < I.end - excl; p += 64) {
            const u32 v = (u32)d[p];
            if (v < (u32)IV_HB) atomicAdd(&lh[w][v], 1u); else atomicAdd(&h[v], 1ull);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const u32 hb = min(S.hist_len, (u32)IV_HB);
        for (u32 b = (u32)lane; b < hb; b += 64) { const u32 x = lh[w][b]; if (x) { atomicAdd(&h[b], (unsigned long long)x); lh[w][b] = 0u; } }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

}  // namespace covk
