// Reader-stage PAIR filter on the device: ReferenceSortedBamFilter::read, pair branch
// (src/filter.rs:117-228) with filter_out = true, over the session's own record store as the device ingest filled it
// (k_bam_extract keeps next_refID and a 96-bit hash of the read name when the session asks for mates).
//
// What the reference does, record by record: among primary proper-pair records (:133-146) it keeps a set of "first" records of
// the current reference keyed by read name, emptied whenever the reference changes (:149-161); a record whose name is not in the
// set is parked there if its mate maps to the same reference (:170-176); a record whose name is in the set takes its partner out,
// the pair is judged (:188-207: the single-read predicate on both when single thresholds are set, then read_pair_passes_filter
// :281-336) and, if it passes, the FIRST record is returned followed by the SECOND (:208-217).
//
// Here: a per-(reference, name) hash join.  Records are sorted by reference (the ingest verified that their keys never decrease,
// else the file goes to the CPU reader), so "the set of the current reference" is "the records with this tid" and chunks of
// whole references are independent.  Per chunk:
//   k_pair_insert    every eligible record finds / claims the table entry of its (tid, name hash) and leaves its index there
//                    (lowest, highest, count) — no order dependence, no locks
//   k_pair_resolve   every entry with exactly two records is one candidate pair (first = lower index, second = higher): parked
//                    only if the first's mate reference is this reference, then judged with the reference's own f32 expressions;
//                    partner[second] = first.  An entry with more than two records (a read name that occurs three or more times
//                    among the primary proper-pair records of one reference) needs the serial park / take-out sequence: counted
//                    per chunk; for such chunks k_pair_collect lists the members of those entries, the host replays the sequence
//                    over the (short) sorted list and k_pair_judge_list judges the candidate pairs it yields.  Only a file with
//                    more than 2^20 such records goes to the CPU reader.
// The output order is the reference's: pairs in the order of their SECOND records, first record then second — an exclusive scan
// over partner[] gives every pair its two output slots (k_scan_*), and k_pair_gather* builds the selected record store.
//
// Name hashes: 96 bits (MurmurHash3-x86-128-style mixing, three of its four words) + the tid compared in full; two different
// names of one reference collide with probability ~n^2 / 2^97, 10^-13 for 2*10^8 records.  The claim protocol (two 64-bit
// compare-and-swaps per entry) never merges different keys whatever the interleaving.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace covp {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u32 NONE = 0xffffffffu;

struct PairEntry {     // 32 B, all-zero = empty
    u64 k1;            // name hash words 0-1 (never 0)
    u64 k2t;           // name hash word 2 | (tid + 2) << 32 (never 0)
    u32 nlo;           // max over the entry's records of ~index: lowest index = ~nlo
    u32 hi;            // highest index
    u32 cnt, pad;
};

struct PairCols {
    const int32_t *tid; const uint16_t *flag; const uint8_t *mapq, *nm_kind;
    const u32 *nm, *l_seq, *cigar_off, *cigar;
    const int32_t *mtid; const u64 *qh1; const u32 *qh2;
};

struct PairFilter {    // cov_pair_filter (covermhip.h)
    int32_t filter_single;
    uint8_t min_mapq, pad[3];
    u32 min_aligned_length_single;
    float min_percent_identity_single, min_aligned_percent_single;
    u32 min_aligned_length_pair;
    float min_percent_identity_pair, min_aligned_percent_pair;
};

__device__ __forceinline__ bool eligible(u32 flag) { return !(flag & 0x900u) && (flag & 0x2u); }     // filter.rs:133-146

// Primary records of the whole store (filter.rs:129-131 counts them before any test).
__global__ __launch_bounds__(256) void k_count_primary(const uint16_t *__restrict__ flag, u32 n, u64 *__restrict__ out) {
    u32 c = 0;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) c += (flag[i] & 0x900u) ? 0u : 1u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (u64)c);
}

// cuts[c], c = 0 .. n_chunks: chunk c = records [cuts[c], cuts[c + 1]) — whole references only: the cut for target c * T is the
// first record whose key exceeds the key of record c * T - 1 (binary search: keys never decrease).
__device__ __forceinline__ u32 key_of(int32_t tid) { return tid < 0 ? 0x7fffffffu : (u32)tid; }
__global__ void k_pair_cuts(const int32_t *__restrict__ tid, u32 n, u32 T, u32 n_chunks, u32 *__restrict__ cuts) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_chunks) return;
    if (c == 0) { cuts[0] = 0; return; }
    if (c == n_chunks || (u64)c * T >= n) { cuts[c] = n; return; }
    const u32 k = key_of(tid[c * T - 1]);
    u32 lo = c * T, hi = n;               // first index in [lo, hi] with key > k
    while (lo < hi) { const u32 mid = lo + (hi - lo) / 2; if (key_of(tid[mid]) > k) hi = mid; else lo = mid + 1; }
    cuts[c] = lo;
}

__device__ __forceinline__ u32 table_slot(u64 k1, u64 k2t, u32 mask) {
    u64 x = k1 ^ (k2t * 0x9e3779b97f4a7c15ull);
    x ^= x >> 29;
    return (u32)x & mask;
}

__global__ __launch_bounds__(256) void k_pair_insert(PairCols C, u32 r0, u32 r1, PairEntry *__restrict__ T, u32 mask, u32 *__restrict__ counters) {
    const u32 i = r0 + blockIdx.x * 256u + threadIdx.x;
    if (i >= r1) return;
    if (!eligible(C.flag[i])) return;
    u64 k1 = C.qh1[i]; if (!k1) k1 = 1;
    const u64 k2t = (u64)C.qh2[i] | ((u64)((u32)C.tid[i] + 2u) << 32);
    u32 h = table_slot(k1, k2t, mask);
    for (u32 probe = 0; probe <= mask; probe++, h = (h + 1u) & mask) {
        PairEntry *e = T + h;
        const u64 p1 = atomicCAS(&e->k1, 0ull, k1);
        if (p1 != 0ull && p1 != k1) continue;
        // the entry carries my first word (mine or an equal one): whoever sets the second word first owns it; a different key with
        // an equal first word moves on to the next slot, and so does every later record of that key
        const u64 p2 = atomicCAS(&e->k2t, 0ull, k2t);
        if (p2 != 0ull && p2 != k2t) continue;
        atomicMax(&e->nlo, ~i); atomicMax(&e->hi, i); atomicAdd(&e->cnt, 1u);
        return;
    }
    atomicAdd(counters + 1, 1u);      // table full: the host sized it at twice the chunk, so this is a bug, reported as such
}

__device__ __forceinline__ u32 aligned_of(const PairCols &C, u32 i, bool with_del) {
    u32 a = 0;
    for (u32 c = C.cigar_off[i], e = C.cigar_off[i + 1]; c < e; c++) {
        const u32 w = C.cigar[c], op = w & 15u, len = w >> 4;
        if (op == 0u || op == 1u || op == 7u || op == 8u || (with_del && op == 2u)) a += len;
    }
    return a;
}

// One candidate pair (first record i1 parked, second record i2 arrived): the reference's judgement (filter.rs:188-207), with its
// short-circuit order; first_err: lowest (second index << 8 | COV_ERR_NM_* code) over the pairs whose judgement needed a missing NM.
__device__ __forceinline__ void judge_pair(const PairCols &C, const PairFilter &f, u32 i1, u32 i2, u32 *__restrict__ partner, u64 *__restrict__ first_err,
                                           u32 err_missing, u32 err_badtype) {
    u32 err = 0;
    auto nm = [&](u32 i, u64 &v) -> bool {    // lib.rs:138-158: the reference panics; here the first such record in file order is reported
        const u32 k = C.nm_kind[i];
        if (k == 1u) { v = C.nm[i]; return true; }
        if (!err) err = k == 0u ? err_missing : err_badtype;
        return false;
    };
    auto single_ok = [&](u32 i) -> bool {     // filter.rs:243-279
        const u32 mq = C.mapq[i];
        if (f.min_mapq != 255 && (mq < f.min_mapq || mq == 255u)) return false;
        u64 ed;
        if (!nm(i, ed)) return false;
        const u32 al = aligned_of(C, i, true);
        return al >= f.min_aligned_length_single && (float)al / (float)C.l_seq[i] >= f.min_aligned_percent_single &&
               1.0f - (float)ed / (float)al >= f.min_percent_identity_single;
    };
    bool pass = !f.filter_single || (single_ok(i1) && single_ok(i2));
    if (pass) {                                // filter.rs:281-336 (record = second, record1 = first)
        const u32 m1 = C.mapq[i1], m2 = C.mapq[i2];
        if (f.min_mapq != 255 && (m1 < f.min_mapq || m2 < f.min_mapq || m1 == 255u || m2 == 255u)) pass = false;
        else {
            u64 e2, e1;
            if (!nm(i2, e2) || !nm(i1, e1)) pass = false;
            else {
                const u64 ed = e2 + e1;
                const u32 al = aligned_of(C, i2, false) + aligned_of(C, i1, false);
                pass = al >= f.min_aligned_length_pair && (float)al / (float)((u64)C.l_seq[i1] + C.l_seq[i2]) >= f.min_aligned_percent_pair &&
                       1.0f - (float)ed / (float)al >= f.min_percent_identity_pair;
            }
        }
    }
    if (err) atomicMin(first_err, ((u64)i2 << 8) | (u64)err);
    if (pass) partner[i2] = i1;
}

// counters: [1] table overflow; multi[chunk]: entries of this chunk with more than two records
__global__ __launch_bounds__(256) void k_pair_resolve(PairCols C, const PairEntry *__restrict__ T, u32 n_entries, PairFilter f, u32 *__restrict__ partner,
                                                      u32 *__restrict__ multi, u64 *__restrict__ first_err, u32 err_missing, u32 err_badtype) {
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_entries) return;
    const PairEntry e = T[g];
    if (e.cnt < 2u) return;
    if (e.cnt > 2u) { atomicAdd(multi, 1u); return; }
    const u32 i1 = ~e.nlo, i2 = e.hi;
    if (C.mtid[i1] != C.tid[i1]) return;      // the first record was not parked (filter.rs:170-176); the second then finds nothing
    judge_pair(C, f, i1, i2, partner, first_err, err_missing, err_badtype);
}

// Members of the entries with more than two records: (entry, record index, would this record be parked).  n_list counts every
// member even when the list is full (the host then knows it was).
struct MultiRec { u32 entry, i, parks, pad; };
__global__ __launch_bounds__(256) void k_pair_collect(PairCols C, u32 r0, u32 r1, const PairEntry *__restrict__ T, u32 mask, MultiRec *__restrict__ list, u32 cap,
                                                      u32 *__restrict__ n_list) {
    const u32 i = r0 + blockIdx.x * 256u + threadIdx.x;
    if (i >= r1) return;
    if (!eligible(C.flag[i])) return;
    u64 k1 = C.qh1[i]; if (!k1) k1 = 1;
    const u64 k2t = (u64)C.qh2[i] | ((u64)((u32)C.tid[i] + 2u) << 32);
    u32 h = table_slot(k1, k2t, mask);
    for (u32 probe = 0; probe <= mask; probe++, h = (h + 1u) & mask) {
        const PairEntry e = T[h];
        if (e.k1 == 0ull) return;              // (cannot happen: every eligible record of the chunk was inserted)
        if (e.k1 != k1 || e.k2t != k2t) continue;
        if (e.cnt > 2u) {
            const u32 slot = atomicAdd(n_list, 1u);
            if (slot < cap) { MultiRec m; m.entry = h; m.i = i; m.parks = C.mtid[i] == C.tid[i] ? 1u : 0u; m.pad = 0; list[slot] = m; }
        }
        return;
    }
}
__global__ __launch_bounds__(256) void k_pair_judge_list(PairCols C, const uint2 *__restrict__ pairs, u32 n, PairFilter f, u32 *__restrict__ partner,
                                                         u64 *__restrict__ first_err, u32 err_missing, u32 err_badtype) {
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n) return;
    judge_pair(C, f, pairs[g].x, pairs[g].y, partner, first_err, err_missing, err_badtype);
}

// ---------------------------------------------------------------------------------------------- device-wide exclusive scan
// Three launches: per-block sums (4096 items per block), one workgroup scans the block sums, every block re-reads its items and
// hands (item, exclusive prefix) to the consumer.  F: value of item i; O: consumer.
constexpr u32 SCAN_ITEMS = 16, SCAN_BLOCK = 256 * SCAN_ITEMS;

__device__ __forceinline__ u32 wave_incl_scan(u32 x) {
    for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(x, o); if ((int)(threadIdx.x & 63) >= o) x += y; }
    return x;
}
// exclusive prefix of x over the 256 threads of the workgroup; *total receives the workgroup's sum
__device__ __forceinline__ u32 block_excl_scan(u32 x, u32 *total) {
    __shared__ u32 wsum[4];
    const u32 incl = wave_incl_scan(x);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 base = 0;
    for (u32 w = 0; w < (threadIdx.x >> 6); w++) base += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + incl - x;
}

template <typename F>
__global__ __launch_bounds__(256) void k_scan_sums(F f, u32 n, u32 *__restrict__ block_sums) {
    const u32 i0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    u32 s = 0;
    for (u32 k = 0; k < SCAN_ITEMS; k++) if (i0 + k < n) s += f(i0 + k);
    u32 total;
    (void)block_excl_scan(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// block_sums[0 .. nb) -> exclusive prefixes in place, grand total at block_sums[nb]  (64-bit total at total64)
__global__ __launch_bounds__(1024) void k_scan_offsets(u32 *__restrict__ block_sums, u32 nb, u64 *__restrict__ total64) {
    __shared__ u64 part[1024];
    const u32 t = threadIdx.x, per = (nb + 1023u) / 1024u, k0 = t * per, k1 = min(nb, k0 + per);
    u64 s = 0;
    for (u32 k = k0; k < k1; k++) s += block_sums[k];
    part[t] = s;
    __syncthreads();
    if (t == 0) { u64 a = 0; for (u32 i = 0; i < 1024; i++) { const u64 x = part[i]; part[i] = a; a += x; } *total64 = a; }
    __syncthreads();
    u64 a = part[t];
    for (u32 k = k0; k < k1; k++) { const u32 x = block_sums[k]; block_sums[k] = (u32)a; a += x; }
    if (t == 0) block_sums[nb] = (u32)*total64;
}
template <typename F, typename O>
__global__ __launch_bounds__(256) void k_scan_apply(F f, O out, u32 n, const u32 *__restrict__ block_offs) {
    const u32 i0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS], s = 0;
    for (u32 k = 0; k < SCAN_ITEMS; k++) { v[k] = i0 + k < n ? f(i0 + k) : 0u; s += v[k]; }
    u32 total;
    u32 p = block_offs[blockIdx.x] + block_excl_scan(s, &total);
    for (u32 k = 0; k < SCAN_ITEMS; k++) { if (i0 + k < n) out(i0 + k, p); p += v[k]; }
}

struct PairSlots { const u32 *partner; __device__ u32 operator()(u32 i) const { return partner[i] != NONE ? 2u : 0u; } };
struct PairEmit {      // order[p] = first, order[p + 1] = second
    const u32 *partner; u32 *order;
    __device__ void operator()(u32 i, u32 p) const { const u32 a = partner[i]; if (a != NONE) { order[p] = a; order[p + 1] = i; } }
};
struct SelCigarLen { const u32 *order, *cigar_off; __device__ u32 operator()(u32 p) const { const u32 j = order[p]; return cigar_off[j + 1] - cigar_off[j]; } };

struct Store {
    int32_t *tid, *pos; uint16_t *flag; uint8_t *mapq, *nm_kind; u32 *nm, *l_seq, *cigar_off, *cigar;
};
// Consumer of the CIGAR-length scan: writes the selected record p (fixed fields, its new CIGAR offset, its CIGAR words).
struct SelGather {
    const u32 *order; Store src, dst;
    __device__ void operator()(u32 p, u32 coff) const {
        const u32 j = order[p];
        dst.tid[p] = src.tid[j]; dst.pos[p] = src.pos[j]; dst.flag[p] = src.flag[j]; dst.mapq[p] = src.mapq[j]; dst.nm_kind[p] = src.nm_kind[j];
        dst.nm[p] = src.nm[j]; dst.l_seq[p] = src.l_seq[j]; dst.cigar_off[p] = coff;
        const u32 c0 = src.cigar_off[j], n = src.cigar_off[j + 1] - c0;
        for (u32 k = 0; k < n; k++) dst.cigar[coff + k] = src.cigar[c0 + k];
    }
};

// 96-bit hash of a read name (n bytes at p, n >= 0; bytes behind the name are not looked at): MurmurHash3_x86_128's block mixing
// over 16-byte blocks of the zero-padded name, its finalisation, words 1-3 kept.
__device__ __forceinline__ u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ u32 fmix32(u32 h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
__device__ __forceinline__ u32 name_word(const uint8_t *p, u32 o, u32 n) {
    if (o >= n) return 0u;
    const u64 a = (u64)(p + o);
    const u32 *w = (const u32 *)(a & ~3ull);
    const u32 sh = (u32)(a & 3u) * 8u;
    u32 v = sh ? (w[0] >> sh) | (w[1] << (32u - sh)) : w[0];
    if (n - o < 4u) v &= (1u << (8u * (n - o))) - 1u;
    return v;
}
__device__ __forceinline__ void name_hash(const uint8_t *p, u32 n, u64 &k1, u32 &k2) {
    u32 h1 = 0x9747b28cu, h2 = 0x2f0b4a27u, h3 = 0x7ed558ccu, h4 = 0x1b873593u;
    const u32 c1 = 0x239b961bu, c2 = 0xab0e9789u, c3 = 0x38b34ae5u, c4 = 0xa1e38b93u;
    for (u32 o = 0; o < n; o += 16u) {
        u32 w1 = name_word(p, o, n), w2 = name_word(p, o + 4u, n), w3 = name_word(p, o + 8u, n), w4 = name_word(p, o + 12u, n);
        w1 *= c1; w1 = rotl32(w1, 15); w1 *= c2; h1 ^= w1; h1 = rotl32(h1, 19); h1 += h2; h1 = h1 * 5u + 0x561ccd1bu;
        w2 *= c2; w2 = rotl32(w2, 16); w2 *= c3; h2 ^= w2; h2 = rotl32(h2, 17); h2 += h3; h2 = h2 * 5u + 0x0bcaa747u;
        w3 *= c3; w3 = rotl32(w3, 17); w3 *= c4; h3 ^= w3; h3 = rotl32(h3, 15); h3 += h4; h3 = h3 * 5u + 0x96cd1c35u;
        w4 *= c4; w4 = rotl32(w4, 18); w4 *= c1; h4 ^= w4; h4 = rotl32(h4, 13); h4 += h1; h4 = h4 * 5u + 0x32ac3b17u;
    }
    h1 ^= n; h2 ^= n; h3 ^= n; h4 ^= n;
    h1 += h2 + h3 + h4; h2 += h1; h3 += h1; h4 += h1;
    h1 = fmix32(h1); h2 = fmix32(h2); h3 = fmix32(h3); h4 = fmix32(h4);
    h1 += h2 + h3 + h4; h2 += h1; h3 += h1; h4 += h1;
    k1 = (u64)h1 | ((u64)h2 << 32);
    k2 = h3 ^ rotl32(h4, 16);
}

}  // namespace covp
