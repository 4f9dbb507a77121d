// k_prep_lean — the shipped k_prep (round 6).  Same contract as prep_body in pileup_kernels.hip.h (which stays, as k_prep7s, the second
// implementation the tests force with COVERM_PREP_KERNEL=7): reader-stage filter (filter.rs:88-116, 243-279), FlagFilter (lib.rs:67-78),
// CIGAR summary (contig.rs:166-202), per-contig record counters, the 8-byte run word per record, the tile index for k_ranges.
//
// What round 5's counters said about prep_body: ~500 instructions per 64 records (320 VALU), issue-stalled 48 % at eight waves per SIMD, and its
// listing says where they go: 61 SGPRs of kernel arguments + a dozen 64-bit lane masks do not fit the 80 scalar registers a wave has at eight
// waves per SIMD, so 73 of them live in the lanes of two vector registers and every use is a v_readlane / v_writelane (VALU) with its
// hazard no-ops; every record loads its neighbours' tid / pos / cigar_off a second time (three more loads with their own clamped addresses)
// and its contig's length and first tile through two dependent vector loads; the tile index and the group bookkeeping run per lane with
// 64-bit shifts on every step.  This kernel is the same arithmetic on an instruction diet:
//
//   * a WAVE owns 64 x `steps` CONSECUTIVE records, so a record's neighbours are the neighbouring lanes: pos of the record in front and
//     cigar_off of the one behind come by DPP wave shifts (2 VALU each), the wave's edges from the step before (v_readlane) and from the
//     step after — whose tid and cigar_off are loaded one step ahead anyway (the two roots of the dependent loads, as in k_prep8s);
//     6 + 1 loads per record instead of 15: flag, pos, nm_kind, nm, next tid, next cigar_off and ONE 12-byte load of the first three
//     CIGAR words;
//   * the loop body is the COMMON step only — 64 records that carry the tid of the record in front of them, no record with more than
//     three CIGAR operations (all but ~2 % of the steps of a 5 000-contig short-read sample): the contig's length, first tile and mask bit
//     are scalar registers, no record starts or ends a group, every per-contig counter is a wave total in a scalar register (ballot +
//     popcount, first / last record by s_ff1 / s_flbit).  Any other step — a contig border, the store's first or last records, a longer
//     CIGAR — is flagged and left WHOLE to k_prep_generic, a second small launch that does such steps per lane with atomics (the old
//     body's logic for one step); its registers and its arguments cost the loop nothing (an out-of-line call inside the loop was built
//     first: the call's register convention put the loop's own values into scratch);
//   * the tile index no longer counts records per tile with a per-lane "how many lanes follow me" (64-bit shifts per lane): a record whose
//     tile differs from its predecessor's (a LEAD) adds -i to its own tile and +i to the predecessor's, the last record of a group adds
//     i + 1: the sums telescope to the records per tile (mod 2^32) for every contig k_ranges uses them for (one group, sorted), two
//     atomics for ~2 leads per step.
#pragma once
#include "pileup_kernels.hip.h"

namespace covk {

struct __attribute__((packed, aligned(4))) CigTriple { u32 a, b, c; };

__device__ __forceinline__ int dpp_prev(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }   // lane l <- lane l - 1; lane 0 <- edge
__device__ __forceinline__ int dpp_next(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); }   // lane l <- lane l + 1; lane 63 <- edge

// What a record's CIGAR comes to (contig.rs:166-202): the first two merged M/=/X runs, how many there are, reference span, aligned and indel bases.
struct CigSum {
    u64 aligned, indel;
    u32 run_start, run_len, run2_start, run2_len, n_runs, span;
    bool oob, badcig, big;
};

// At most three operations (an absent one reads as 0S), in closed form: consecutive M/=/X operations always merge (an M consumes what it
// aligns), so the only second run is M, <D or N of positive length>, M; everything else is one run that starts at the first M/=/X operation.
__device__ __forceinline__ void cigar_sum3(u32 w0, u32 w1, u32 w2, int pos, u32 L, CigSum &o) {
    const u32 len0 = w0 >> 4, len1 = w1 >> 4, len2 = w2 >> 4;
    const u32 bit0 = 1u << (w0 & 15u), bit1 = 1u << (w1 & 15u), bit2 = 1u << (w2 & 15u);
    const bool m0 = (bit0 & 0x181u) != 0u, m1 = (bit1 & 0x181u) != 0u, m2 = (bit2 & 0x181u) != 0u;       // M = X   (contig.rs:171-186)
    const u32 ref0 = (bit0 & 0x18du) ? len0 : 0u, ref1 = (bit1 & 0x18du) ? len1 : 0u, ref2 = (bit2 & 0x18du) ? len2 : 0u;
    const u32 c0 = (u32)pos, c1 = c0 + ref0, c2 = c1 + ref1;
    o.span = ref0 + ref1 + ref2;
    o.aligned = ((bit0 & 0x187u) ? len0 : 0u) + ((bit1 & 0x187u) ? len1 : 0u) + ((bit2 & 0x187u) ? len2 : 0u);     // M I D = X   (:187-199)
    o.indel = ((bit0 & 0x006u) ? len0 : 0u) + ((bit1 & 0x006u) ? len1 : 0u) + ((bit2 & 0x006u) ? len2 : 0u);       // I D
    o.oob = (m0 && c0 >= L) || (m1 && c1 >= L) || (m2 && c2 >= L);       // negative cursors wrap to >= 2^31 > L
    o.badcig = ((bit0 | bit1 | bit2) & 0xfe00u) != 0u;
    o.big = ((w0 | w1 | w2) >> 28) != 0u;                                // a length of >= 2^24
    const bool second = m0 && !m1 && m2 && ref1 != 0u;
    o.n_runs = ((m0 || m1 || m2) ? 1u : 0u) + (second ? 1u : 0u);
    o.run_start = m0 ? c0 : (m1 ? c1 : c2);
    o.run_len = (m0 ? len0 : 0u) + (m1 ? len1 : 0u) + ((m2 && !second) ? len2 : 0u);
    o.run2_start = second ? c2 : 0u; o.run2_len = second ? len2 : 0u;
}

// The run word of a considered record (see "Run word" in pileup_kernels.hip.h); is_bucket: a long CIGAR that goes through the per-tile buckets.
__device__ __forceinline__ uint2 run_word(const CigSum &c, u32 nops_all, bool &is_bucket) {
    uint2 rw; rw.x = c.run_start; rw.y = 0u;
    const u32 gap = c.run2_start - (c.run_start + c.run_len);
    if (c.n_runs == 1 && c.run_len < (1u << 30)) rw.y = c.run_len;   // RW_SINGLE; 0 = nothing to add
    else if (c.n_runs == 2 && c.run_len - 1u < 1023u && c.run2_len - 1u < 1023u && gap - 1u < 255u)
        rw.y = (RW_DOUBLE << 30) | c.run_len | (gap << 10) | (c.run2_len << 18);
    else {
        is_bucket = nops_all > CX_MIN_OPS && !c.big;
        rw.y = (is_bucket ? RW_BUCKET : RW_COMPLEX) << 30;
    }
    return rw;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// One step of 64 records starting at i0, per lane, every counter through atomics: what the loop of k_prep_lean leaves on the list (see above).
// the step's independent loads
struct GenFields { u32 fl_raw, nk, nmv32, mq, lsq, co0, co1; int td, pos, pt, ppos, nt; };
template <bool FILTER>
__device__ __forceinline__ void gen_load(const PrepHot &h, u32 i0, GenFields &f) {
    const u32 n = h.n, nlast = n - 1u;
    const u32 i = i0 + (u32)lane_id();
    const bool in = i < n;
    const u32 ic = min(i, nlast);
    f.fl_raw = h.flag[ic];
    f.td = h.tid[ic]; f.pos = h.pos[ic];
    f.nk = h.nmk[ic]; f.nmv32 = h.nm[ic];
    f.mq = FILTER ? (u32)h.mapq[ic] : 0u; f.lsq = FILTER ? h.lseq[ic] : 0u;
    f.co0 = h.coff[ic]; f.co1 = h.coff[ic + 1u];
    f.pt = (in && i > 0u) ? h.tid[ic - 1u] : -2;
    f.ppos = (in && i > 0u) ? h.pos[ic - 1u] : 0;
    f.nt = (in && i + 1u < n) ? h.tid[ic + 1u] : -2;
}
template <bool WANT_IDENTITY, bool FILTER, bool MASKED>
__device__ __forceinline__ void prep_step_generic(const PrepArgs *__restrict__ pa, u32 i0, const GenFields &f) {
    const PrepHot &h = pa->hot;
    const PrepCold &cd = pa->cold;
    const int lane = lane_id();
    const u32 n = h.n;
    const u32 i = i0 + (u32)lane;
    const bool in = i < n;
    const u32 fl_raw = f.fl_raw, nk = f.nk, nmv32 = f.nmv32, mq = f.mq, lsq = f.lsq, co0 = f.co0, co1 = f.co1;
    const int td = f.td, pos = f.pos, pt = f.pt, ppos = f.ppos, nt = f.nt;
    const u32 flag = in ? fl_raw : 0x904u;
    const int tid = in ? td : -1;
    const bool tid_ok = tid >= 0 && (u32)tid < h.n_targets;
    const u32 Lv = tid_ok ? cd.tlen[tid] : 0u, t0v = tid_ok ? cd.tile_first[tid] : 0u;
    const u32 mkv = (MASKED && tid_ok) ? (u32)cd.mask[tid] : 1u;
    DevGlobal *g = cd.g;

    const u32 n_prim = (u32)__popcll(__ballot(!(flag & 0x900u)));
    const bool supp = flag & 0x800u, sec = flag & 0x100u;
    // span of records carrying this tid + grouping / position-order checks (all records, considered or not)
    if (tid_ok) {
        DevContig *C = &cd.ctg[tid];
        if (pt != tid) { atomicMin(&C->rec_start, i); atomicAdd(&C->n_groups, 1u); }
        else if (ppos > pos) atomicOr(&C->flags, F_POS_UNSORTED);
        if (nt != tid) atomicMax(&C->rec_end, i + 1u);
    }
    // reader stage: ReferenceSortedBamFilter::read single-read branch, filter_out = true (filter.rs:88-116)
    bool survives = in, need_filter_eval = false;
    if (FILTER) {
        survives = false;
        const bool p1 = in && !(flag & h.p1_mask);
        if (p1 && !(h.min_mapq != 255u && (mq < h.min_mapq || mq == 255u))) need_filter_eval = true;  // :250-254
    }
    const bool scan_gate = ((flag ^ 2u) & h.gate_mask) == 0u;       // FlagFilter::passes (lib.rs:67-78) then !unmapped (contig.rs:125)
    const bool do_walk = FILTER ? need_filter_eval : scan_gate;
    const u32 nops_all = do_walk ? co1 - co0 : 0u;
    CigSum cs;
    bool hard = nops_all > CIG_FAST_OPS;
    const u32 cl = h.cigar_end ? h.cigar_end - 1u : 0u;
    const u32 cw0 = h.cigar_end ? h.cigar[min(co0, cl)] : 0u, cw1 = h.cigar_end ? h.cigar[min(co0 + 1u, cl)] : 0u, cw2 = h.cigar_end ? h.cigar[min(co0 + 2u, cl)] : 0u;
    if (__all(nops_all <= 3u)) {
        cigar_sum3(0u < nops_all ? cw0 : 4u, 1u < nops_all ? cw1 : 4u, 2u < nops_all ? cw2 : 4u, pos, Lv, cs);
        hard = cs.big;
    } else {
        // branch-free 32-bit state machine, trip count uniform over the wave: valid while nothing can overflow (at most CIG_FAST_OPS
        // operations of < 2^24 each); anything else is redone by the whole-wave walk below
        const u32 nops = hard ? 0u : nops_all;
        u32 cursor = (u32)pos, cur_e = 0, al32 = 0, in32 = 0;
        cs.run_start = cs.run_len = cs.run2_start = cs.run2_len = cs.n_runs = 0; cs.oob = cs.badcig = cs.big = false;
        auto step = [&](u32 wd) {
            const u32 len = wd >> 4, bit = 1u << (wd & 15u);
            cs.big |= len >= (1u << 24);
            const bool m = (bit & 0x181u) != 0u;                       // M = X   (contig.rs:171-186)
            cs.badcig |= (bit & 0xfe00u) != 0u;
            cs.oob |= m && cursor >= Lv;                               // negative cursors wrap to >= 2^31 > L
            const bool ext = m && cs.n_runs > 0u && cursor == cur_e;   // continues the open merged run
            cs.n_runs += (m && !ext) ? 1u : 0u;
            const bool r1 = m && cs.n_runs == 1u, r2 = m && cs.n_runs == 2u;
            cs.run_start = (r1 && !ext) ? cursor : cs.run_start;
            cs.run2_start = (r2 && !ext) ? cursor : cs.run2_start;
            cs.run_len = r1 ? (ext ? cs.run_len : 0u) + len : cs.run_len;
            cs.run2_len = r2 ? (ext ? cs.run2_len : 0u) + len : cs.run2_len;
            cur_e = m ? cursor + len : cur_e;
            cursor += (bit & 0x18du) ? len : 0u;                       // M D N = X consume the reference
            al32 += (bit & 0x187u) ? len : 0u;                         // M I D = X   (:187-199)
            in32 += (bit & 0x006u) ? len : 0u;                         // I D
        };
        step(0u < nops ? cw0 : 4u);
        step(1u < nops ? cw1 : 4u);
        step(2u < nops ? cw2 : 4u);
        for (u32 c = 3; __any(c < nops); c++) step(c < nops ? h.cigar[co0 + c] : 4u);
        cs.aligned = al32; cs.indel = in32;
        cs.span = cursor - (u32)pos;
        hard |= cs.big;
    }
    for (u64 hm = __ballot(hard); hm != 0; hm &= hm - 1) {      // long CIGARs: one at a time, whole wave on each
        const int hl = __ffsll((long long)hm) - 1;
        WalkOut o;
        cigar_walk_wave(h.cigar, __builtin_amdgcn_readlane(co0, hl), __builtin_amdgcn_readlane(nops_all, hl),
                        (int)__builtin_amdgcn_readlane((u32)pos, hl), __builtin_amdgcn_readlane(Lv, hl), o);
        if (lane == hl) {
            cs.big |= o.absurd;
            if (o.absurd) cigar_walk_slow(h.cigar, co0, nops_all, pos, Lv, cs.aligned, cs.indel, cs.run_start, cs.run_len,
                                          cs.run2_start, cs.run2_len, cs.n_runs, cs.span, cs.oob, cs.badcig);
            else {
                cs.aligned = o.aligned; cs.indel = o.indel; cs.run_start = o.run_start; cs.run_len = o.run_len;
                cs.run2_start = o.run2_start; cs.run2_len = o.run2_len; cs.n_runs = o.n_runs; cs.span = o.span;
                cs.oob = o.oob; cs.badcig = o.badcig;
            }
        }
    }
    if (FILTER && need_filter_eval) {  // single_read_passes_filter, filter.rs:256-278
        if (nk != 1u) report_error(g, i, nk == 0u ? 2u : 3u);
        else {
            const u32 al = (u32)cs.aligned;  // u32 accumulation in the reference
            const float a = (float)al;
            survives = al >= h.min_aligned_length && a / (float)lsq >= h.min_aligned_percent &&
                       1.0f - (float)nmv32 / a >= h.min_percent_identity;
        }
    }
    const bool considered = survives && scan_gate;
    const bool masked_in = considered && tid_ok && mkv;
    u64 nmv = 0;
    double idv = 0.0;
    if (considered && !tid_ok) report_error(g, i, 7u);  // header.target_len(tid).expect("Corrupt BAM file?")
    if (masked_in) {
        if (cs.badcig) report_error(g, i, 6u);
        else if (cs.oob) report_error(g, i, 4u);
        if (nk != 1u) report_error(g, i, nk == 0u ? 2u : 3u);  // nm(&record), contig.rs:206
        else nmv = nmv32;
        if (WANT_IDENTITY && cs.aligned > 0) idv = ((double)cs.aligned - (double)nmv) / (double)cs.aligned;
    }
    bool is_bucket = false;
    if (in) {
        uint2 rw = make_uint2(0u, 0u);
        if (masked_in && cs.n_runs > 0) rw = run_word(cs, nops_all, is_bucket);
        h.runs[i] = rw;
        if (WANT_IDENTITY) {   // a NULL stream is one the caller does not need (COV_WANT_IDENTITY_*_ONLY)
            if (h.identn != nullptr) h.identn[i] = (masked_in && !supp) ? idv : 0.0;
            if (h.identp != nullptr) h.identp[i] = (masked_in && !supp && !sec) ? idv : 0.0;
        }
    }
    {   // RW_BUCKET records: one aggregated append per wave
        const u64 qm = __ballot(is_bucket);
        if (qm != 0) {
            const int ql = __ffsll((long long)qm) - 1;
            u32 b = 0;
            if (lane == ql) b = atomicAdd(&g->n_cx, (u32)__popcll(qm));
            b = __builtin_amdgcn_readlane(b, ql);
            const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(qm >> 32), __builtin_amdgcn_mbcnt_lo((u32)qm, 0u));
            if (is_bucket && b + rank < cd.cx_list_cap) cd.cx_list[b + rank] = i;
        }
    }
    // ---- tile index (every record of a real contig counts, considered or not): telescoping sums, see the header
    const bool tv = tid_ok && Lv > 0u;
    const u32 pc = pos < 0 ? 0u : min((u32)pos, Lv - 1u);
    const u32 tl = pc >> h.shift;
    const u32 key = t0v + tl;
    if (tv) {
        const bool same = pt == tid;                                                          // the record in front continues this group
        const u32 pkey = t0v + ((ppos < 0 ? 0u : min((u32)ppos, Lv - 1u)) >> h.shift);      // (its tile, if so)
        if (!same || pkey != key) { atomicAdd(&h.tcnt[key], 0u - i); if (same) atomicAdd(&h.tcnt[pkey], i); }
        if (nt != tid) atomicAdd(&h.tcnt[key], i + 1u);
        // records that reach beyond their own tile announce themselves to the tiles they enter
        const u32 e = cs.span > Lv - pc ? Lv : pc + cs.span;
        const u32 te = cs.span > 0u ? (e - 1u) >> h.shift : 0u;
        if (masked_in && cs.n_runs > 0u && cs.span > 0u && te > tl && !is_bucket)           // buckets deliver those
            for (u32 t2 = tl + 1u; t2 <= te; t2++) atomicMin(&h.fov[t0v + t2], i);
    }
    // ---- per-contig counters: one set of atomics per RUN of records of one contig, not per record (a hot accumulator line takes ~12 ns per
    // atomic, and a step at a contig border — or every step of an assembly with more contigs than records per step — has a few runs of
    // many records each).  Wave-uniform loop over the distinct contigs of the step's considered records; counts are popcounts of ballots,
    // the three sums one DPP prefix sum each.
    const bool cnt = considered && tid_ok;
    const u64 cm = __ballot(cnt);
    {
        const u64 pm = __ballot(cnt && !(flag & 0x900u)), nsm = __ballot(cnt && !supp);
        const u32 nm_c = masked_in ? (u32)nmv : 0u, in_c = masked_in ? (u32)cs.indel : 0u, sp_c = masked_in ? cs.span : 0u;
        const bool wide = __any(masked_in && (nmv32 >= (1u << 25) || cs.indel >= (1ull << 25)));      // 64 such values could pass 2^31: 64-bit sums then
        // the loop leaves every run's totals in the registers of the run's FIRST lane; the atomics follow it, all first lanes at once: eight
        // memory instructions per step whatever the number of runs (one per run and counter before: ~50 memory instructions per step of an
        // assembly, and the CU's one address path, not the L2's atomic units, set k_prep_generic's time — profiles/r06_c2M_pmc.json)
        u32 r_np = 0, r_n = 0, r_nn = 0, r_sp = 0, r_first = 0, r_last = 0;
        u64 r_nm = 0, r_in = 0;
        u64 heads = 0;
        for (u64 todo = cm; todo != 0ull;) {
            const int fl = __builtin_ctzll(todo);
            const int t = __builtin_amdgcn_readlane(tid, fl);
            const bool mine = cnt && tid == t;
            const u64 m = __ballot(mine);
            todo &= ~m;
            heads |= 1ull << fl;
            u64 s_nm, s_in;
            if (!wide) {
                s_nm = (u32)__builtin_amdgcn_readlane(wave_incl_scan((int)(mine ? nm_c : 0u)), 63);
                s_in = (u32)__builtin_amdgcn_readlane(wave_incl_scan((int)(mine ? in_c : 0u)), 63);
            } else {
                s_nm = wave_sum_u64(mine && masked_in ? nmv : 0ull); s_in = wave_sum_u64(mine && masked_in ? cs.indel : 0ull);
            }
            const u32 s_sp = wave_max_u32_dpp(mine ? sp_c : 0u);
            if (lane == fl) {
                r_np = (u32)__popcll(m & pm); r_n = (u32)__popcll(m); r_nn = (u32)__popcll(m & nsm);
                r_nm = s_nm; r_in = s_in; r_sp = s_sp;
                r_first = i0 + (u32)__builtin_ctzll(m); r_last = i0 + 63u - (u32)__builtin_clzll(m);
            }
        }
        if ((heads >> lane) & 1ull) {
            DevContig *C = &cd.ctg[tid];
            if (r_np) atomicAdd(&C->n_primary, (u64)r_np);
            atomicAdd(&C->n_pass, (u64)r_n);
            if (r_nn) atomicAdd(&C->n_nonsupp, (u64)r_nn);
            if (r_nm) atomicAdd(&C->sum_nm, r_nm);
            if (r_in) atomicAdd(&C->sum_indel, r_in);
            if (r_sp) atomicMax(&C->max_span, r_sp);
            atomicMin(&C->first_rec, r_first);
            atomicMax(&C->last_rec, r_last);
        }
    }
    if (lane == 0) {
        const u32 slot = ((i0 >> 6) % COUNTER_SLOTS) * 8u;
        if (n_prim) atomicAdd(&g->prim_slots[slot], (u64)n_prim);
        if (cm) atomicAdd(&g->cons_slots[slot], (u64)__popcll(cm));
    }
}

// per-wave running sums for the contig the wave is in (prim / pass / nons / first / last are wave totals in scalar registers)
struct LeanAcc {
    u32 prim, pass, nons, first, last, span;
    u64 nm, indel;
    __device__ __forceinline__ void reset() { prim = pass = nons = 0; first = 0xffffffffu; last = 0; span = 0; nm = indel = 0; }
};

__device__ __forceinline__ void lean_flush(DevContig *ctg, int cur, LeanAcc &a) {
    if (cur >= 0 && a.pass) {
        const u32 span = wave_max_u32(a.span);
        const u64 nm = wave_sum_u64(a.nm), indel = wave_sum_u64(a.indel);
        if (lane_id() == 0) {
            DevContig *C = &ctg[cur];
            if (a.prim) atomicAdd(&C->n_primary, (u64)a.prim);
            atomicAdd(&C->n_pass, (u64)a.pass);
            if (a.nons) atomicAdd(&C->n_nonsupp, (u64)a.nons);
            if (nm) atomicAdd(&C->sum_nm, nm);
            if (indel) atomicAdd(&C->sum_indel, indel);
            if (span) atomicMax(&C->max_span, span);
            atomicMin(&C->first_rec, a.first);
            atomicMax(&C->last_rec, a.last);
        }
    }
    a.reset();
}

template <bool WANT_IDENTITY, bool FILTER, bool MASKED>
__device__ __forceinline__ void prep_lean_body(const PrepHot &h, const PrepArgs *__restrict__ pa) {
    __shared__ u32 blk_cnt[2][4];
    __shared__ u32 wmask[4];
    __shared__ PrepPartial wpart[4];
    const PrepCold *cold = &pa->cold;
    const int lane = lane_id();
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 wave_recs = h.steps * 64u;
    const u32 base = (blockIdx.x * 4u + w) * wave_recs;
    bool flushed_early = false;
    LeanAcc acc; acc.reset();
    int cur = -1;                 // contig the running sums belong to
    u32 g_prim = 0, g_cons = 0;
    u32 left_mask = 0;            // bit s: step s of this wave's records is left to k_prep_generic (h.steps <= 32)

    if (base < h.n) {
        const u32 nlast = h.n - 1u;
        // wave-relative addressing: uniform bases + a small per-lane offset, so that every load is `global_load v, voff, s[base]`
        const int *tid_w = h.tid + base; const int *pos_w = h.pos + base; const uint16_t *flag_w = h.flag + base;
        const uint8_t *nmk_w = h.nmk + base; const u32 *nm_w = h.nm + base; const u32 *coff_w = h.coff + base;
        const uint8_t *mapq_w = h.mapq + base; const u32 *lseq_w = h.lseq + base;
        uint2 *runs_w = h.runs + base;
        const u32 lmax = nlast - base;                 // last wave-relative index of a record (coff may be read one further)
        constexpr u32 LM = 0x1ffffu;                   // (a no-op mask that bounds the offsets for the compiler: steps <= 1024)

        int ptid_e = -2, ppos_e = 0;                   // the record in front of this step's first one
        if (base > 0) { ptid_e = h.tid[base - 1u]; ppos_e = h.pos[base - 1u]; }
        u32 cur_L = 0, cur_t0 = 0, cur_mk = 1u;      // the contig `cur` of the last common step: length, first tile, mask bit
        // the two roots of the dependent loads, one step ahead
        int td; u32 co0;
        { const u32 l = (u32)lane; td = tid_w[min(l, lmax) & LM]; co0 = coff_w[min(l, lmax + 1u) & LM]; }
        asm volatile("" : "+v"(td), "+v"(co0));      // (here, not in the loop: the compiler otherwise waits for ALL loads at the loop's first use of them)

        for (u32 s = 0; s < h.steps; s++) {
            const u32 l0 = s * 64u, i0 = base + l0;
            if (i0 >= h.n) break;
            const u32 l = l0 + (u32)lane, i = i0 + (u32)lane;
            const u32 lc = min(l, lmax) & LM;
            // ---- the next step's roots, this step's independent fields, the first three CIGAR words: one load phase
            const int tdn = tid_w[min(l + 64u, lmax) & LM];
            const u32 con = coff_w[min(l + 64u, lmax + 1u) & LM];
            const u32 flag = flag_w[lc];
            const int pos = pos_w[lc];
            const u32 nk = nmk_w[lc];
            const u32 nmv32 = nm_w[lc];
            const u32 mq = FILTER ? (u32)mapq_w[lc] : 0u;
            const u32 lsq = FILTER ? lseq_w[lc] : 0u;
            u32 cw0 = 4u, cw1 = 4u, cw2 = 4u;          // (an absent operation reads as 0S)
            bool moved = false;
            if (h.cigar_end >= 3u) {
                const u32 cb = min(co0, h.cigar_end - 3u);      // (the store's last records: the triple is moved back to stay inside; such a step goes the generic way)
                moved = cb != co0;
                const CigTriple t3 = *reinterpret_cast<const CigTriple *>(h.cigar + cb);
                cw0 = t3.a; cw1 = t3.b; cw2 = t3.c;
            }

            // ---- is this a common step?  64 records and one behind them, all of the contig of the record in front
            const int tu = __builtin_amdgcn_readfirstlane(td);
            const bool full = i0 + 64u < h.n;
            bool common = full && h.cigar_end >= 3u && tu >= 0 && (u32)tu < h.n_targets && ptid_e == tu && __ballot(td != tu) == 0ull;
            const u32 co1 = (u32)dpp_next((int)co0, __builtin_amdgcn_readfirstlane((int)con));
            // reader stage: ReferenceSortedBamFilter::read single-read branch, filter_out = true (filter.rs:88-116)
            bool survives = true, need_filter_eval = false;
            if (FILTER) {
                survives = false;
                const bool p1 = !(flag & h.p1_mask);
                if (p1 && !(h.min_mapq != 255u && (mq < h.min_mapq || mq == 255u))) need_filter_eval = true;  // :250-254
            }
            const bool scan_gate = ((flag ^ 2u) & h.gate_mask) == 0u;       // FlagFilter::passes (lib.rs:67-78) then !unmapped (contig.rs:125)
            const bool do_walk = FILTER ? need_filter_eval : scan_gate;
            const u32 nops_all = do_walk ? co1 - co0 : 0u;
            common = common && __ballot(nops_all > 3u || moved) == 0ull;
            CigSum cs;
            if (common) {
                if (tu != cur) {      // a new contig: the running sums of the one before leave, this one's length, first tile and mask bit come in
                    if (cur >= 0) flushed_early = true;
                    lean_flush(cold->ctg, cur, acc); cur = tu;
                    cur_L = cold->tlen[tu]; cur_t0 = cold->tile_first[tu]; cur_mk = MASKED ? (u32)cold->mask[tu] : 1u;
                }
                cigar_sum3(0u < nops_all ? cw0 : 4u, 1u < nops_all ? cw1 : 4u, 2u < nops_all ? cw2 : 4u, pos, cur_L, cs);
                common = __ballot(cs.big) == 0ull;           // (an operation of >= 2^24 bases: the literal 64-bit walk)
            }
            if (!common) {      // left to k_prep_generic: a bit in the wave's mask (a scalar register; stored once, behind the loop)
                left_mask |= 1u << (s & 31u);
                ptid_e = __builtin_amdgcn_readlane(td, 63);
                ppos_e = full ? h.pos[i0 + 63u] : 0;
                td = tdn; co0 = con;
                continue;
            }
            const u32 Lc = cur_L, t0c = cur_t0;
            const int ntid_e = __builtin_amdgcn_readfirstlane(tdn);
            const int ppos = dpp_prev(pos, ppos_e);
            // every record read counts towards num_detected_primary_alignments when !secondary && !supplementary
            g_prim += (u32)__popcll(__ballot(!(flag & 0x900u)));
            const bool supp = flag & 0x800u, sec = flag & 0x100u;
            bool nm_err = false;
            if (FILTER && need_filter_eval) {  // single_read_passes_filter, filter.rs:256-278
                if (nk != 1u) nm_err = true;
                else {
                    const u32 al = (u32)cs.aligned;  // u32 accumulation in the reference
                    const float a = (float)al;
                    survives = al >= h.min_aligned_length && a / (float)lsq >= h.min_aligned_percent &&
                               1.0f - (float)nmv32 / a >= h.min_percent_identity;
                }
            }
            const bool considered = survives && scan_gate;
            const bool masked_in = considered && cur_mk;
            u64 nmv = 0;
            double idv = 0.0;
            if (masked_in) {
                if (nk != 1u) nm_err = true;      // nm(&record), contig.rs:206
                else nmv = nmv32;
                if (WANT_IDENTITY && cs.aligned > 0) idv = ((double)cs.aligned - (double)nmv) / (double)cs.aligned;
            }
            // the reference's panics, in one rare branch
            if (__builtin_expect(__any(nm_err || (masked_in && (cs.badcig || cs.oob))), 0)) {
                DevGlobal *g = cold->g;
                if (masked_in) { if (cs.badcig) report_error(g, i, 6u); else if (cs.oob) report_error(g, i, 4u); }
                if (nm_err) report_error(g, i, nk == 0u ? 2u : 3u);
            }
            {
                uint2 rw = make_uint2(0u, 0u);
                bool never = false;       // (at most three operations: never a bucket record)
                if (masked_in && cs.n_runs > 0) rw = run_word(cs, 0u, never);
                runs_w[l & LM] = rw;
                if (WANT_IDENTITY) {   // a NULL stream is one the caller does not need (COV_WANT_IDENTITY_*_ONLY)
                    if (h.identn != nullptr) h.identn[i] = (masked_in && !supp) ? idv : 0.0;
                    if (h.identp != nullptr) h.identp[i] = (masked_in && !supp && !sec) ? idv : 0.0;
                }
            }
            // ---- tile index for k_ranges (every record of a real contig counts towards F, considered or not)
            if (Lc > 0u) {
                const u32 pc = pos < 0 ? 0u : min((u32)pos, Lc - 1u);
                const u32 tl = pc >> h.shift;
                const u32 key = t0c + tl;
                const u32 pkey_e = t0c + ((ppos_e < 0 ? 0u : min((u32)ppos_e, Lc - 1u)) >> h.shift);      // scalar: the record in front of lane 0 is of this contig
                const u32 pkey = (u32)dpp_prev((int)key, (int)pkey_e);
                if (key != pkey) { atomicAdd(&h.tcnt[key], 0u - i); atomicAdd(&h.tcnt[pkey], i); }
                if (__builtin_expect(ntid_e != tu, 0)) { if (lane == 63) atomicAdd(&h.tcnt[key], i + 1u); }      // the group ends with this step's last record
                // records that reach beyond their own tile announce themselves to the tiles they enter
                const u32 e = cs.span > Lc - pc ? Lc : pc + cs.span;                 // end of the record's reference extent, clipped
                const u32 te = cs.span > 0u ? (e - 1u) >> h.shift : 0u;
                const bool cross = masked_in && cs.n_runs > 0u && cs.span > 0u && te > tl;
                if (__any(cross)) {
                    if (__builtin_expect(__all(!cross || te == tl + 1u), 1)) {
                        // usual case: a record enters one more tile; the lowest index of a run of such records with one target is enough
                        const u32 ckey = cross ? key : 0xfffffffeu;
                        const u32 pck = (u32)dpp_prev((int)ckey, (int)0xfffffffdu);
                        if (cross && ckey != pck) atomicMin(&h.fov[key + 1u], i);
                    } else if (cross) for (u32 t2 = tl + 1u; t2 <= te; t2++) atomicMin(&h.fov[t0c + t2], i);
                }
            }
            // ---- the group's order and end (no record of a common step starts a group)
            if (__builtin_expect(__any(ppos > pos), 0)) { if (lane == 0) atomicOr(&cold->ctg[tu].flags, F_POS_UNSORTED); }
            if (__builtin_expect(ntid_e != tu, 0)) { if (lane == 63) atomicMax(&cold->ctg[tu].rec_end, i + 1u); }
            // ---- per-contig counters: wave totals in scalar registers
            const u64 cm = __ballot(considered);
            if (cm != 0ull) {
                g_cons += (u32)__popcll(cm);
                acc.prim += (u32)__popcll(__ballot(considered && !(flag & 0x900u)));
                acc.pass += (u32)__popcll(cm);
                acc.nons += (u32)__popcll(__ballot(considered && !supp));
                acc.first = min(acc.first, i0 + (u32)__builtin_ctzll(cm));
                acc.last = max(acc.last, i0 + 63u - (u32)__builtin_clzll(cm));
                if (masked_in) { acc.nm += nmv; acc.indel += cs.indel; acc.span = max(acc.span, cs.span); }
            }
            // ---- the next step: its roots are here, this step's last record is its predecessor
            ptid_e = tu; ppos_e = __builtin_amdgcn_readlane(pos, 63);
            td = tdn; co0 = con;
        }
    }
    // End of the workgroup's records: if all four waves stayed inside the same single contig, publish ONE partial record for the
    // workgroup (reduced by k_post_prep); otherwise atomics.
    {
        PrepPartial pw;
        pw.tid = flushed_early ? -2 : cur;
        pw.prim = acc.prim; pw.pass = acc.pass; pw.nons = acc.nons;
        pw.span = wave_max_u32(acc.span); pw.first = acc.first; pw.last = acc.last;
        pw.nm = wave_sum_u64(acc.nm); pw.indel = wave_sum_u64(acc.indel); pw.pad = 0;
        if (lane == 0) wpart[w] = pw;
    }
    if (lane == 0) { blk_cnt[0][w] = g_prim; blk_cnt[1][w] = g_cons; wmask[w] = left_mask; }
    __syncthreads();
    // The steps this workgroup leaves to k_prep_generic go on its list with ONE atomic per workgroup (most workgroups of a 5 000-contig sample
    // have none).  (One atomic per STEP was built first: 15 000 appends at 5 000 contigs went unnoticed, 300 000 at 200 000 contigs
    // serialised on the counter's cache line and this kernel took 1.6 ms instead of 0.33 — profiles/r06_c200k_kernel_stats.csv; a flag per
    // step and a mask per wave, read by k_prep_generic itself, left that kernel with 64 x fewer busy waves: 0.11 ms instead of 0.014.)
    if (w == 0u) {
        const u32 m0 = wmask[0], m1 = wmask[1], m2 = wmask[2], m3 = wmask[3];
        const u32 c0 = (u32)__popc(m0), c1 = c0 + (u32)__popc(m1), c2 = c1 + (u32)__popc(m2), total = c2 + (u32)__popc(m3);
        if (total) {
            u32 at = 0;
            if (lane == 0) at = atomicAdd(&cold->g->n_gen, total);
            at = (u32)__builtin_amdgcn_readfirstlane((int)at);
            for (u32 e = (u32)lane; e < total; e += 64u) {      // entry e: the (e - c_{k-1})-th set bit of wave k's mask
                const u32 k = e < c0 ? 0u : e < c1 ? 1u : e < c2 ? 2u : 3u;
                u32 m = k == 0u ? m0 : k == 1u ? m1 : k == 2u ? m2 : m3;
                for (u32 r = e - (k == 0u ? 0u : k == 1u ? c0 : k == 2u ? c1 : c2); r != 0u; r--) m &= m - 1u;
                cold->gen_list[at + e] = (blockIdx.x * 4u + k) * wave_recs + (u32)__builtin_ctz(m) * 64u;
            }
        }
    }
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
            cold->part[blockIdx.x] = o;
        }
    }
    if (!uniform_wg) lean_flush(cold->ctg, cur, acc);
    // device-wide counters: one pair of atomics per workgroup, spread over COUNTER_SLOTS cache lines
    if (threadIdx.x == 0) {
        const u32 p = blk_cnt[0][0] + blk_cnt[0][1] + blk_cnt[0][2] + blk_cnt[0][3];
        const u32 c = blk_cnt[1][0] + blk_cnt[1][1] + blk_cnt[1][2] + blk_cnt[1][3];
        const u32 slot = (blockIdx.x % COUNTER_SLOTS) * 8u;
        DevGlobal *g = cold->g;
        if (p) atomicAdd(&g->prim_slots[slot], (u64)p);
        if (c) atomicAdd(&g->cons_slots[slot], (u64)c);
    }
}

template <bool WANT_IDENTITY, bool FILTER, bool MASKED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_prep_lean(PrepHot h, const PrepArgs *__restrict__ pa) {
    prep_lean_body<WANT_IDENTITY, FILTER, MASKED>(h, pa);
}
// The steps k_prep_lean listed: waves stride over the list, a whole wave per step.  all_steps != 0: every step of the store, no list (the host
// launches this kernel alone when the sample has so many contigs that few steps would be common ones).
template <bool WANT_IDENTITY, bool FILTER, bool MASKED>
__global__ __launch_bounds__(256) void k_prep_generic(const PrepArgs *__restrict__ pa, u32 all_steps) {
    const u32 n = all_steps ? all_steps : pa->cold.g->n_gen;
    const u32 *__restrict__ list = pa->cold.gen_list;
    for (u32 j = blockIdx.x * 4u + (threadIdx.x >> 6); j < n; j += gridDim.x * 4u) {
        const u32 i0 = all_steps ? j * 64u : (u32)__builtin_amdgcn_readfirstlane((int)list[j]);
        GenFields f;
        gen_load<FILTER>(pa->hot, i0, f);
        prep_step_generic<WANT_IDENTITY, FILTER, MASKED>(pa, i0, f);
    }
}
// (Measured at 2 M contigs, where this kernel is all of k_prep — 1.62 ms, profiles/r06_c2M_pmc.json: 2.85 GB of counter traffic for 1.83 GB of
// records —, and changed nothing: the runs' atomics leaving together instead of run by run (~50 -> ~30 memory instructions per step: kept, it
// is simpler), and the next step's loads issued in front of the current step's work (1.66 ms: removed).  What the traffic says: ~30 atomic
// lane-operations per step land on 160-byte accumulators spread over 320 MB — read-modify-writes at the memory side, not in an XCD's L2.)

}  // namespace covk
