// Device-side ingest: BGZF inflate and BAM record parsing on the GPU.
//
// The reference's only parallel stage is htslib's inflate pool (bam_generator.rs:125-129); on a host with few cores that
// stage bounds the whole path.  Here the compressed file goes to HBM as it is and three groups of kernels replace the
// host decoder:
//
//   k_inflate       one LANE per BGZF block (blocks are independent raw-DEFLATE members of <= 64 KiB, SAM spec 4.1):
//                   a complete RFC 1951 decoder per lane — stored / fixed / dynamic blocks, Huffman decode through per-lane
//                   lookup tables in LDS (7-bit literal/length, 6-bit distance primary tables by default; longer codes resolved
//                   canonically from per-length counts).  Literals are written in place, LZ77 matches become tokens.
//                   64 blocks advance per wave instruction; divergence between lanes is what SIMT costs here, HBM
//                   bandwidth is not the limit.
//   k_lz_resolve    one WAVE per block executes its match tokens (kept in place, in the first bytes of their own match), the
//                   bytes of all matches that are ready spread over the 64 lanes
//   k_crc32_wave    every inflated block against the CRC-32 of its BGZF trailer: a wave per block, 8-byte columns, csrc/crc_wave_core.h
//   k_crc32         the same with a lane per block (with the lane-per-block inflate)
//   k_bam_find      per segment of the inflated stream: first offset from which a chain of 8 plausible records starts
//   k_bam_hop       per segment: hop the records (block_size chain), count records and CIGAR words, note where it landed
//   k_bam_verify    every segment's chain must land exactly on the start the next segment found (then the result equals the
//                   serial hop by induction, as in csrc/host_bam.cpp); exclusive scans of the counts
//   k_bam_extract   per segment: second hop, one record at a time per lane, fields written straight into the session's
//                   record store (tid, pos, flag, mapq, l_seq, NM + its type, CIGAR words)
//
// Anything irregular (malformed stream, CRC mismatch, speculation that does not verify) raises
// a flag and the host decodes that file with the CPU reader instead: the device path is an accelerator, never a different
// semantics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pair_kernels.hip.h"

namespace covi {

typedef unsigned long long u64;
typedef unsigned int u32;

struct BgzfBlock {
    u64 in_off;    // first byte of the raw DEFLATE stream inside the compressed buffer
    u64 out_off;   // where the inflated bytes go
    u32 in_len, isize, crc, pad;
};

// Per-lane LDS, lane-interleaved u16 entries (entry i of lane l at [i * 64 + l]): the primary tables of both alphabets and the
// distance alphabet's symbols sorted by (length, value).  A wave takes 64 x (2^LIT_BITS + 2^DIST_BITS + 32) x 2 B: 44 KiB at
// 8 + 6 bits (three waves per CU), 28 KiB at 7 + 6 (five), 24 KiB at 7 + 5 (six), 16 KiB at 6 + 5 (nine).  In SIMT the
// canonical search for longer codes is paid whenever ANY lane needs it — practically every step — so a smaller table costs
// little by itself, and a lane decodes its block serially at the latency of a dependent instruction chain: more resident
// waves are what raises throughput (COVERM_INFLATE_BITS / COVERM_INFLATE_DIST_BITS).
// sort8: the literal/length alphabet's symbols sorted by (length, value) live in LDS too, one BYTE each (bit 8 of a symbol follows
// from its rank: within a length the symbols >= 256 come last) — +18 KiB per wave, and the canonical path never goes to global
// memory: a global load per symbol step (some lane of 64 always has a long code) was what each step waited for.
constexpr size_t inflate_smem_bytes(int lit_bits, int dist_bits, bool sort8 = false) { return (size_t)64 * ((1u << lit_bits) + (1u << dist_bits) + 32) * 2 + (sort8 ? (size_t)64 * 288 : 0) + (size_t)64 * 8 * 4; }
// Per-lane global scratch: symbols sorted by (code length, value) for codes longer than the primary tables, and the code lengths
// while a table is being built:  u16 lit_sorted[288], dist_sorted[32], u8 lens[320]
constexpr u32 INF_SCRATCH_BYTES = 288 * 2 + 32 * 2 + 320;
// LZ77 matches are not copied by k_inflate (a byte-serial copy by ONE lane would stall the other 63 of its wave on every match):
// it writes the literals at their final positions and one token per match; k_lz_resolve then executes the tokens of each block in
// order with a whole wave per block.  A token lives IN PLACE, in the first three output bytes of its own match (a match is at
// least three bytes long: (dist - 1) | (len - 3) << 15 takes 23 bits); only its 16-bit output position goes to a side list.
// 43 KiB of list per block instead of 171 KiB of 64-bit tokens: the side buffers of a round stay small however many blocks a
// round holds (device allocations beyond ~60 GB in total were measured to cost seconds).
constexpr u32 INF_TOK_CAP = 21888;   // >= 65536 / 3 matches per block
typedef unsigned short tokpos_t;

enum { INF_OK = 0, INF_ERR_FORMAT = 1, INF_ERR_CRC = 2, INF_ERR_SIZE = 3 };

__constant__ unsigned char c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Bit reader over one lane's compressed stream: 64-bit buffer refilled 4 bytes at a time — from a per-lane RESERVOIR of INF_RES
// words in LDS, not from global memory.  A global load inside refill() is waited for on the spot (the refill is conditional per
// lane, so its result is selected right behind it), and in SIMT some lane of the 64 needs a word at almost every refill: three
// global round trips per symbol step, 67 % of the kernel's wave time.  The reservoir is topped up for ALL active lanes at once
// (eight words each, two 16-byte loads) whenever any lane runs low: one wait per four to five symbol steps, and the word for the
// next refill is read from LDS ahead of its use.
constexpr u32 INF_RES = 8;
struct __attribute__((packed, aligned(4))) Words4 { u32 a, b, c, d; };
struct BitReader {
    const u32 *words;   // aligned base
    u32 *res;           // LDS reservoir, word j of this lane at res[j * 64]
    u64 buf; u32 cnt;   // cnt valid bits in buf
    u32 next_word;      // index of the next word to enter buf
    u32 rbase;          // index of the word in reservoir slot 0
    u32 consumed;       // bits consumed so far
    u32 ahead;          // words[next_word], already read from the reservoir
    // Bytes behind the stream's end come from whatever follows it in the buffer (the buffers carry slack for this): a stream that
    // runs past its end is caught by the consumed-bits / size / CRC checks, as the zero padding of earlier versions was.
    __device__ __forceinline__ void reload() {
        const Words4 x = *reinterpret_cast<const Words4 *>(words + next_word), y = *reinterpret_cast<const Words4 *>(words + next_word + 4);
        res[0 * 64] = x.a; res[1 * 64] = x.b; res[2 * 64] = x.c; res[3 * 64] = x.d;
        res[4 * 64] = y.a; res[5 * 64] = y.b; res[6 * 64] = y.c; res[7 * 64] = y.d;
        rbase = next_word;
    }
    __device__ __forceinline__ void init(const uint8_t *base, u64 off, u32 len, u32 *reservoir) {
        const u32 mis = (u32)((u64)(base + off) & 3u);
        words = (const u32 *)(base + off - mis);        // derived from the kernel argument: stays a global (not flat) pointer
        res = reservoir;
        next_word = 0; reload();
        buf = (u64)res[0] | ((u64)res[64] << 32); cnt = 64u;
        next_word = 2; ahead = res[2 * 64];
        if (mis) { buf >>= 8 * mis; cnt -= 8 * mis; }
        consumed = 0;
        (void)len;
    }
    // continue at absolute bit position `bitpos` (counted from the aligned base, i.e. stream bit + 8 * mis)
    __device__ __forceinline__ void seek(u32 bitpos, u32 mis) {
        next_word = bitpos >> 5; reload();
        buf = (u64)res[0] | ((u64)res[64] << 32); cnt = 64u;
        next_word += 2u; ahead = res[2 * 64];
        const u32 d = bitpos & 31u;
        buf >>= d; cnt -= d;
        consumed = bitpos - 8u * mis;
    }
    __device__ __forceinline__ void refill() {
        if (cnt <= 32u) {
            buf |= (u64)ahead << cnt;
            cnt += 32u;
            next_word++;
        }
        if (__any(next_word - rbase >= INF_RES - 1u)) reload();      // decided for the active lanes together; a lane with words left just reloads early
        ahead = res[(next_word - rbase) * 64u];
    }
    __device__ __forceinline__ u32 peek(u32 n) const { return (u32)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(u32 n) { buf >>= n; cnt -= n; consumed += n; }
    __device__ __forceinline__ u32 take(u32 n) { const u32 v = peek(n); drop(n); return v; }   // n <= 16; caller keeps cnt >= n via refill()
};

// Sixteen 16-bit counters of one lane in four registers (counter l in bits (l & 3) * 16 of word l >> 2): per-length code
// counts and running offsets while a table is built, without LDS round trips or dynamically indexed register arrays.
struct Pack16 {
    // four named words, not an array: with an array the compiler turns get()'s select chain into a dynamically indexed load from
    // scratch memory — a memory round trip (and a full s_waitcnt vmcnt(0)) in every symbol step
    u64 w0, w1, w2, w3;
    __device__ __forceinline__ void clear() { w0 = w1 = w2 = w3 = 0; }
    __device__ __forceinline__ u32 get(u32 l) const {
        const u64 lo = (l & 4u) ? w1 : w0, hi = (l & 4u) ? w3 : w2;
        const u64 x = (l & 8u) ? hi : lo;
        return (u32)(x >> ((l & 3u) * 16u)) & 0xffffu;
    }
    template <u32 L> __device__ __forceinline__ u32 at() const {     // compile-time index
        const u64 x = (L >> 2) == 0 ? w0 : (L >> 2) == 1 ? w1 : (L >> 2) == 2 ? w2 : w3;
        return (u32)(x >> ((L & 3u) * 16u)) & 0xffffu;
    }
    __device__ __forceinline__ void add(u32 l, u32 v) {
        const u64 inc = (u64)v << ((l & 3u) * 16u);
        const u32 k = l >> 2;
        w0 += k == 0u ? inc : 0ull; w1 += k == 1u ? inc : 0ull; w2 += k == 2u ? inc : 0ull; w3 += k == 3u ? inc : 0ull;
    }
};

// One alphabet's decoding state of a lane.
struct Huff {
    unsigned short *tab;          // LDS primary table (lane-interleaved)
    unsigned short *sorted;       // symbols by (length, value): global scratch (stride 1) or LDS (lane-interleaved, stride 64)
    u32 sstride;
    // Codes longer than the primary table, resolved WITHOUT a bit-serial walk: with the next 15 stream bits reversed into v (the
    // code left-justified, MSB first), canonical codes of length l occupy [.., limit[l]) with limit[l] = (first code of length l
    // + count[l]) << (15 - l), non-decreasing in l, so l = 1 + #{l' : v >= limit[l']}; the symbol is sorted[off[l] + (v >> (15 - l))]
    // with off[l] = (index of the first symbol of length l) - (first code of length l)  (mod 2^16).  Both arrays live in registers.
    Pack16 limit, off;
    // SORT8 (literal/length alphabet in LDS as bytes): sorted8[j * 64] = low byte of the j-th symbol; thr[l] = index of the first
    // symbol >= 256 among those of length l (= end of the length's run when there is none)
    uint8_t *sorted8;
    u32 s8stride = 64;            // distance between consecutive entries of sorted8 (64: lane-interleaved LDS; 1: the lane's own global scratch)
    Pack16 thr;
};

// Builds the lane's primary lookup table and counts (LDS) and the sorted symbols (global scratch) from code lengths
// (`lens` is 4-byte aligned; one pass for the counts, one for the placement: O(n_sym)).
// Primary entry: bits 0-3 code length (0 = longer than prim_bits: resolve canonically), bits 4-15 symbol.
template <bool SORT8>
__device__ __forceinline__ bool build_table(const uint8_t *lens, u32 n_sym, u32 prim_bits, int lane, Huff &H) {
    unsigned short *sorted = H.sorted; const u32 ss = H.sstride;
    Pack16 cnt; cnt.clear();
    Pack16 cntlit; cntlit.clear();
    const u32 *lw = (const u32 *)lens;
    for (u32 s = 0; s < n_sym; s += 4) {
        u32 w = lw[s >> 2];
#pragma unroll
        for (u32 k = 0; k < 4; k++, w >>= 8) {
            const u32 l = w & 0xffu;
            if (s + k < n_sym && l) { cnt.add(l & 15u, 1u); if (SORT8 && s + k < 256u) cntlit.add(l & 15u, 1u); }
        }
    }
    // an over-subscribed set is an error; incomplete sets are tolerated (a single distance code is legal, anything else
    // decodes to -1 when an unassigned code appears and the block's CRC catches the rest)
    u32 left = 1;
    Pack16 offs; offs.clear();
    H.limit.clear(); H.off.clear(); H.thr.clear();
    {
        u32 run = 0, first = 0;
        for (u32 l = 1; l < 16; l++) {
            const u32 c = cnt.get(l);
            left <<= 1; if (c > left) return false; left -= c;
            offs.add(l, run);
            H.limit.add(l, min((first + c) << (15u - l), 0xffffu));      // 2^15 at most, except past a complete set (saturate)
            H.off.add(l, (run - first) & 0xffffu);
            if (SORT8) H.thr.add(l, run + cntlit.get(l));
            run += c; first = (first + c) << 1;
        }
    }
    for (u32 s = 0; s < n_sym; s += 4) {      // symbols in increasing value: each length's run comes out sorted by value
        u32 w = lw[s >> 2];
#pragma unroll
        for (u32 k = 0; k < 4; k++, w >>= 8) {
            const u32 l = w & 0xffu;
            if (s + k < n_sym && l) {
                if (SORT8) H.sorted8[offs.get(l & 15u) * H.s8stride] = (uint8_t)(s + k); else sorted[offs.get(l & 15u) * ss] = (unsigned short)(s + k);
                offs.add(l & 15u, 1u);
            }
        }
    }
    const u32 N = 1u << prim_bits;
    for (u32 i = 0; i < N; i++) H.tab[i * 64 + lane] = 0;
    // canonical codes in increasing (length, symbol) order; primary entries for lengths <= prim_bits (bit-reversed: DEFLATE
    // packs Huffman codes MSB first into an LSB-first bit stream)
    u32 code = 0, idx = 0;
    for (u32 l = 1; l <= prim_bits; l++) {
        const u32 c = cnt.get(l), th = SORT8 ? H.thr.get(l) : 0u;
        for (u32 k = 0; k < c; k++, idx++, code++) {
            const u32 sym = SORT8 ? (u32)H.sorted8[idx * H.s8stride] + (idx >= th ? 256u : 0u) : (u32)sorted[idx * ss];
            const u32 rev = __brev(code) >> (32 - l);
            const unsigned short e = (unsigned short)((sym << 4) | l);
            for (u32 i = rev; i < N; i += 1u << l) H.tab[i * 64 + lane] = e;
        }
        code <<= 1;
    }
    return true;
}

// l += #{k in [K, 15) : v >= limit[k]}  (compile-time unrolled over the packed limits)
template <u32 K>
__device__ __forceinline__ void count_ge(const Pack16 &lim, u32 v, u32 &l) {
    if constexpr (K < 15u) {
        l += v >= lim.at<K>() ? 1u : 0u;
        count_ge<K + 1u>(lim, v, l);
    }
}

// Decodes one symbol: primary table, else the parallel length search described at Huff (no loop, no memory until the final
// sorted[] lookup).
template <int PRIM, bool SORT8>
__device__ __forceinline__ int decode_sym(BitReader &br, const Huff &H, int lane) {
    const u32 pk = br.peek(PRIM);
    const u32 e = H.tab[pk * 64 + lane];
    if (e & 15u) { br.drop(e & 15u); return (int)(e >> 4); }
    const u32 v = __brev((u32)br.buf) >> 17;          // next 15 bits, first stream bit on top
    u32 l = PRIM + 1;
    count_ge<PRIM + 1>(H.limit, v, l);
    if (v >= H.limit.get(15)) return -1;               // not a code of this (incomplete) set
    br.drop(l);
    const u32 idx = (H.off.get(l) + (v >> (15u - l))) & 0xffffu;
    if (SORT8) { const u32 ic = min(idx, 287u); return (int)((u32)H.sorted8[ic * 64u] + (ic >= H.thr.get(l) ? 256u : 0u)); }
    return (int)H.sorted[idx * H.sstride];
}

// One lane per BGZF block: Huffman decoding only.  Literals go to out + out_off at their final positions, matches become
// tokens (tok + local block index * INF_TOK_CAP, count in n_tok).  status[b] = INF_*.
template <int INF_LIT_BITS, int INF_DIST_BITS, bool SORT8>
__global__ __launch_bounds__(64) void k_inflate(const uint8_t *__restrict__ comp, const BgzfBlock *__restrict__ blocks, u32 n_blocks,
                                                uint8_t *__restrict__ out, uint8_t *__restrict__ scratch, tokpos_t *__restrict__ tok,
                                                u32 *__restrict__ n_tok, u32 *__restrict__ status, u32 *__restrict__ n_failed) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    constexpr int INF_LIT_N = 1 << INF_LIT_BITS;
    const int lane = threadIdx.x & 63;
    Huff HL, HD;
    HL.tab = lds; HD.tab = lds + (size_t)64 * INF_LIT_N;
    unsigned short *dsorted_lds = HD.tab + (size_t)64 * (1 << INF_DIST_BITS);   // 32 x 64
    HL.sorted8 = reinterpret_cast<uint8_t *>(dsorted_lds + 64 * 32) + lane;        // 288 x 64 bytes when SORT8
    unsigned short *lds_res = dsorted_lds + 64 * 32 + (SORT8 ? 32 * 288 : 0);     // INF_RES x 64 words: the bit readers' reservoirs
    HD.sorted8 = nullptr;
    const u32 b = blockIdx.x * 64u + (u32)lane;
    if (b >= n_blocks) return;      // (after the wave-wide table fill above)
    const BgzfBlock B = blocks[b];
    uint8_t *sc = scratch + (size_t)b * INF_SCRATCH_BYTES;
    uint8_t *lens = sc + 640;
    HD.sorted = dsorted_lds + lane; HD.sstride = 64;
    HL.sorted = (unsigned short *)sc; HL.sstride = 1;
    uint8_t *dst = out + B.out_off;
    tokpos_t *my_tok = tok + (size_t)b * INF_TOK_CAP;
    u32 pos = 0, err = INF_OK, nt = 0;
    // Literals are gathered eight at a time and leave as one 8-byte store: a byte store per symbol from 64 lanes is 64 separate
    // L2 transactions, and every wait for an input word also waits for the stores in front of it.
    u64 obuf = 0; u32 on = 0;
    u64 tbuf = 0;      // token positions leave four at a time too (one 8-byte store per four matches)
    auto flush_out = [&]() {       // the pending literals are the `on` bytes that end at dst + pos
        if (on) {
            uint8_t *d = dst + pos - on;
            // one 8-byte store: the zero bytes behind the literals land on positions this lane (later literals) or k_lz_resolve
            // (the match that follows) writes afterwards; byte stores only where the 8 bytes would leave the block
            if (pos - on + 8u <= B.isize) __builtin_memcpy(d, &obuf, 8);
            else for (u32 k = 0; k < on; k++) d[k] = (uint8_t)(obuf >> (8u * k));
            obuf = 0; on = 0;
        }
    };
    if (B.isize != 0u) {
        BitReader br;
        br.init(comp, B.in_off, B.in_len, reinterpret_cast<u32 *>(lds_res) + lane);
        const u32 total_bits = B.in_len * 8u;
        bool last = false;
        while (!last && err == INF_OK) {
            // checked in front of every DEFLATE block (stored blocks `continue` past the check at the bottom): a payload of empty
            // stored blocks cannot walk on behind its own end for longer than one block
            if (br.consumed > total_bits) { err = INF_ERR_FORMAT; break; }
            br.refill();
            last = br.take(1) != 0u;
            const u32 type = br.take(2);
            if (type == 0u) {                    // stored
                br.drop((8u - (br.consumed & 7u)) & 7u);
                br.refill();
                const u32 len = br.take(16);
                br.refill();
                const u32 nlen = br.take(16);
                if ((len ^ nlen) != 0xffffu || pos + len > B.isize) { err = INF_ERR_FORMAT; break; }
                flush_out();
                for (u32 k = 0; k < len; k++) { br.refill(); dst[pos++] = (uint8_t)br.take(8); }
                continue;
            }
            if (type == 3u) { err = INF_ERR_FORMAT; break; }
            u32 hlit, hdist;
            if (type == 1u) {                    // fixed codes
                for (u32 s = 0; s < 144; s++) lens[s] = 8;
                for (u32 s = 144; s < 256; s++) lens[s] = 9;
                for (u32 s = 256; s < 280; s++) lens[s] = 7;
                for (u32 s = 280; s < 288; s++) lens[s] = 8;
                for (u32 s = 0; s < 30; s++) lens[288 + s] = 5;
                hlit = 288; hdist = 30;
            } else {                             // dynamic codes
                br.refill();
                hlit = br.take(5) + 257u; hdist = br.take(5) + 1u;
                const u32 hclen = br.take(4) + 4u;
                if (hlit > 286u || hdist > 30u) { err = INF_ERR_FORMAT; break; }
                // code-length code (19 symbols, at most 7 bits): lengths packed 3 bits each into one u64, decoded canonically
                u64 cl = 0;
                for (u32 i = 0; i < hclen; i++) { br.refill(); cl |= (u64)br.take(3) << (3u * c_clen_order[i]); }
                u32 ccnt = 0;      // 8 counts of 4 bits... up to 19 codes of one length: 5 bits each
                u64 ccnt5 = 0;
                for (u32 sy = 0; sy < 19; sy++) { const u32 l = (u32)(cl >> (3u * sy)) & 7u; if (l) ccnt5 += 1ull << (5u * l); }
                (void)ccnt;
                { u32 left = 1; bool bad = false; for (u32 l = 1; l < 8; l++) { left <<= 1; const u32 c = (u32)(ccnt5 >> (5u * l)) & 31u; if (c > left) { bad = true; break; } left -= c; } if (bad) { err = INF_ERR_FORMAT; break; } }
                u32 n = 0;
                const u32 want = hlit + hdist;
                while (n < want && err == INF_OK) {
                    br.refill();
                    // canonical decode over lengths 1..7; the symbol of rank r among the codes of length l is the r-th symbol (in
                    // value order) whose length is l
                    u32 code = 0, first = 0; int sym = -1;
                    u64 bits = br.buf;
                    for (u32 l = 1; l <= 7; l++) {
                        code |= (u32)(bits & 1u); bits >>= 1;
                        const u32 c = (u32)(ccnt5 >> (5u * l)) & 31u;
                        if (code < first + c) {
                            u32 r = code - first;
                            for (u32 sy = 0; sy < 19; sy++) if (((u32)(cl >> (3u * sy)) & 7u) == l) { if (r == 0) { sym = (int)sy; break; } r--; }
                            br.drop(l);
                            break;
                        }
                        first += c; first <<= 1; code <<= 1;
                    }
                    if (sym < 0) { err = INF_ERR_FORMAT; break; }
                    if (sym < 16) lens[n++] = (uint8_t)sym;
                    else {
                        u32 rep, val = 0;
                        if (sym == 16) { if (n == 0) { err = INF_ERR_FORMAT; break; } val = lens[n - 1]; rep = 3u + br.take(2); }
                        else if (sym == 17) rep = 3u + br.take(3);
                        else rep = 11u + br.take(7);
                        if (n + rep > want) { err = INF_ERR_FORMAT; break; }
                        for (u32 k = 0; k < rep; k++) lens[n++] = (uint8_t)val;
                    }
                }
                if (err != INF_OK) break;
                if (lens[256] == 0) { err = INF_ERR_FORMAT; break; }
                // distance lengths follow the literal/length ones: move them to their own place
                for (u32 s = hdist; s-- > 0;) lens[288 + s] = lens[hlit + s];   // backwards: the two ranges overlap
                for (u32 s = hlit; s < 288; s++) lens[s] = 0;
            }
            if (!build_table<SORT8>(lens, hlit, INF_LIT_BITS, lane, HL)) { err = INF_ERR_FORMAT; break; }
            if (!build_table<false>(lens + 288, hdist, INF_DIST_BITS, lane, HD)) { err = INF_ERR_FORMAT; break; }
            // ---- symbols of this block
            for (;;) {
                br.refill();
                const int sym = decode_sym<INF_LIT_BITS, SORT8>(br, HL, lane);
                if (sym < 256) {
                    if (sym < 0) { err = INF_ERR_FORMAT; break; }
                    if (pos >= B.isize) { err = INF_ERR_SIZE; break; }
                    obuf |= (u64)(u32)sym << (8u * on);
                    on++; pos++;
                    if (on == 8u) { __builtin_memcpy(dst + pos - 8u, &obuf, 8); obuf = 0; on = 0; }
                    continue;
                }
                flush_out();
                if (sym == 256) break;
                const u32 li = (u32)sym - 257u;
                if (li >= 29u) { err = INF_ERR_FORMAT; break; }
                // RFC 1951 3.2.5: codes 257-264 are lengths 3-10; then groups of four share e extra bits; 285 = 258
                const u32 le = li < 8u ? 0u : (li == 28u ? 0u : (li - 4u) >> 2);
                const u32 lb = li < 8u ? 3u + li : (li == 28u ? 258u : 3u + ((4u + (li & 3u)) << le));
                const u32 len = lb + br.take(le);
                br.refill();
                const int ds = decode_sym<INF_DIST_BITS, false>(br, HD, lane);
                if (ds < 0 || ds >= 30) { err = INF_ERR_FORMAT; break; }
                const u32 dsu = (u32)ds;
                const u32 de = dsu < 4u ? 0u : (dsu - 2u) >> 1;            // distance codes come in pairs sharing e extra bits
                // (no refill here: the one in front of the distance code left >= 33 bits, a code and its extra bits take <= 28)
                const u32 dist = (dsu < 4u ? 1u + dsu : 1u + ((2u + (dsu & 1u)) << de)) + (de ? br.take(de) : 0u);
                if (dist > pos || pos + len > B.isize || nt >= INF_TOK_CAP) { err = INF_ERR_FORMAT; break; }
                {
                    tbuf |= (u64)pos << (16u * (nt & 3u));
                    if ((nt & 3u) == 3u) { __builtin_memcpy(my_tok + (nt - 3u), &tbuf, 8); tbuf = 0; }
                    const u32 t24 = (dist - 1u) | ((len - 3u) << 15);
                    uint8_t *d = dst + pos;
                    // one 4-byte store where it fits: the byte behind the token is this match's own or is written later by this lane
                    if (pos + 4u <= B.isize) __builtin_memcpy(d, &t24, 4);
                    else { d[0] = (uint8_t)t24; d[1] = (uint8_t)(t24 >> 8); d[2] = (uint8_t)(t24 >> 16); }
                }
                nt++;
                pos += len;
            }
            if (br.consumed > total_bits) err = INF_ERR_FORMAT;
        }
        flush_out();
        if (err == INF_OK && pos != B.isize) err = INF_ERR_SIZE;
    }
    if (nt & 3u) __builtin_memcpy(my_tok + (nt & ~3u), &tbuf, 8);     // INF_TOK_CAP is a multiple of four: the store stays inside the block's list
    n_tok[b] = err == INF_OK ? nt : 0u;
    status[b] = err;
    if (err != INF_OK) atomicAdd(n_failed, 1u);
}

// ---------------------------------------------------------------------------------------------- k_inflate_wave
// One WAVE per BGZF block (the default): the 64 lanes share one set of Huffman tables in LDS (7.3 KiB per wave against 28 KiB of
// per-lane tables in k_inflate: 16 waves per CU instead of five) and each decodes 1/64 of the block's bit stream — three passes, described
// with the code in csrc/inflate_wave_core.h, which is plain C++ and runs lane by lane on the CPU in tests/test_inflate_wave_core.py.
// Same contract as k_inflate: literals in place, matches as in-place tokens for k_lz_resolve, n_tok / status per block.
// Measured (profiles/r04_wave_variants.log): 20.3 ms per 81 920 blocks against k_inflate's 27.6, 6.4 ms against 26.4 for 12 k blocks.
}  // namespace covi
#define COVW_FN __device__ __forceinline__
#define COVW_PARFOR(lane) for (unsigned lane = threadIdx.x & 63u, covw_once = 1u; covw_once; covw_once = 0u)
#define COVW_SYNC() __syncthreads()
#define covw_brev32(x) __brev(x)
#define COVW_NO_UNROLL _Pragma("clang loop unroll(disable) vectorize(disable)")
#define COVW_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define COVW_ATOMIC_OR(p, v) atomicOr((p), (v))
#include "inflate_wave_core.h"
namespace covi {
static_assert(covw::TOK_CAP == INF_TOK_CAP && covw::OK == INF_OK && covw::ERR_FORMAT == INF_ERR_FORMAT && covw::ERR_SIZE == INF_ERR_SIZE, "the core mirrors k_inflate's contract");
// (A cursor of two words and a funnel shift instead of the 64-bit buffer was tried: no fewer instructions per unit; a look-ahead cursor with
// 16-byte loads was slower; holding the register allocation to five waves per SIMD changed nothing, to six or seven cost 40 % in spills.)
template <class SinkT, int EXTRA>
__device__ __forceinline__ void inflate_wave_body(const uint8_t *__restrict__ comp, const BgzfBlock *__restrict__ blocks, u32 n_blocks,
                                                     uint8_t *__restrict__ out, tokpos_t *__restrict__ tok, u32 *__restrict__ n_tok,
                                                     u32 *__restrict__ status, u32 *__restrict__ n_failed, u32 stop_after) {
    __shared__ covw::Wave W;
    const u32 b = blockIdx.x;
    if (b >= n_blocks) return;
    const BgzfBlock B = blocks[b];
    u32 st = INF_OK, nt = 0;
    if (B.isize != 0u) {
        const u32 mis = (u32)((u64)(comp + B.in_off) & 3u);
        covw::inflate_block<SinkT, EXTRA>(W, reinterpret_cast<const u32 *>(comp + B.in_off - mis), 8u * mis, 8u * B.in_len, out + B.out_off, B.isize,
                            tok + (size_t)b * INF_TOK_CAP, &nt, &st, stop_after);
    }
    if ((threadIdx.x & 63u) == 0u) {
        n_tok[b] = nt; status[b] = st;
        if (st != INF_OK) atomicAdd(n_failed, 1u);
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_inflate_wave(const uint8_t *__restrict__ comp, const BgzfBlock *__restrict__ blocks, u32 n_blocks,
                                                     uint8_t *__restrict__ out, tokpos_t *__restrict__ tok, u32 *__restrict__ n_tok,
                                                     u32 *__restrict__ status, u32 *__restrict__ n_failed, u32 stop_after) {
    inflate_wave_body<covw::Sink16, 1>(comp, blocks, n_blocks, out, tok, n_tok, status, n_failed, stop_after);
}
// (Up to two and three more literals per lock-step were measured as well — 18.0 and 20.3 ms per full round against 17.1 with one more, ingest
// of 100 M reads 0.339 / 0.328 s against 0.320, profiles/r05_extra_literals_*.log: every further literal slot is another dependent table
// lookup, and a refill, in the path every lock-step takes.  One more it is.  The builds with one unit per lock-step and with the 8-byte sink
// of rounds 3-4 — the measurements' other sides — left the library in round 6; csrc/inflate_wave_core.h still takes both as template
// arguments and tests/test_inflate_wave_core.py runs them on the CPU.)
// (Held to 96 registers = five waves per SIMD, one spilled register: 24.5 ms per round against 22.3 on the same box, profiles/r04_waves5_lzscope.log.)

// Inclusive wave64 prefix sum (DPP row shifts + row broadcasts; VALU latency only).
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return (u32)v;
}

// One wave per BGZF block executes the block's match tokens.  64 tokens are taken at a time, one per lane; a token may only
// run once every byte it reads is final: everything below the output position of the lowest unfinished token is (literals were
// written by k_inflate, earlier matches are done), so in each round the lanes whose source range ends at or below that
// position are copied concurrently (their bytes spread over the 64 lanes), the stores are drained, and the frontier moves
// on.  Matches mostly reach a record or more back while a window of 64 tokens spans a few records, so a window takes a few
// rounds instead of 64 dependent load-store round trips.
// (The source loads at workgroup scope — L1 hits allowed, the bytes were written by this very wave — change nothing: 19.6 ms against 19.7,
// profiles/r04_waves5_lzscope.log.)
__global__ __launch_bounds__(256) void k_lz_resolve(const BgzfBlock *__restrict__ blocks, u32 n_blocks, uint8_t *__restrict__ out,
                                                    const tokpos_t *__restrict__ tok, const u32 *__restrict__ n_tok) {
    const int lane = threadIdx.x & 63;
    const u32 b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const u32 nt = n_tok[b];
    if (nt == 0u) return;
    uint8_t *dst = out + blocks[b].out_off;
    const tokpos_t *my = tok + (size_t)b * INF_TOK_CAP;
    for (u32 t0 = 0; t0 < nt; t0 += 64u) {
        const u32 t = t0 + (u32)lane;
        const bool have = t < nt;
        const u32 pos = have ? (u32)my[t] : 0xffffffffu;
        u32 t24 = 0;
        if (have) { const uint8_t *tp = dst + pos; t24 = (u32)tp[0] | ((u32)tp[1] << 8) | ((u32)tp[2] << 16); }    // written by k_inflate (the kernel before this one)
        const u32 len = have ? ((t24 >> 15) & 0xffu) + 3u : 0u, dist = have ? (t24 & 0x7fffu) + 1u : 0u;
        const u32 src_lo = pos - dist, src_end = pos - dist + min(len, dist), dst_end = pos + len;
        // Output ranges of the window's tokens are disjoint and increasing with the lane, so the tokens whose output overlaps
        // this lane's source range form a contiguous lane interval [dep_lo, dep_hi): dep_hi = lanes with pos < src_end,
        // dep_lo = lanes with dst_end <= src_lo.  Two binary searches over the lanes (6 shuffles each), once per window.
        u32 dep_hi = 0, dep_lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const u32 p_hi = (u32)__shfl((int)pos, (int)(dep_hi + step - 1)), e_lo = (u32)__shfl((int)dst_end, (int)(dep_lo + step - 1));
            const bool v_hi = dep_hi + step <= 64u, v_lo = dep_lo + step <= 64u;
            if (v_hi && p_hi < src_end) dep_hi += step;
            const bool lo_have = (u32)__shfl((int)(have ? 1 : 0), (int)(dep_lo + step - 1)) != 0u;
            if (v_lo && lo_have && e_lo <= src_lo) dep_lo += step;
        }
        dep_hi = min(dep_hi, (u32)lane);            // only earlier tokens matter
        const u64 dep_mask = dep_hi > dep_lo ? ((dep_hi - dep_lo >= 64u ? ~0ull : ((1ull << (dep_hi - dep_lo)) - 1ull)) << dep_lo) : 0ull;
        u64 todo = __ballot(have);
        while (todo) {
            const bool go = have && ((todo >> lane) & 1ull) && (todo & dep_mask) == 0ull;
            // The bytes of all ready matches of this round, concatenated, are copied 64 at a time, one byte per lane: a match of
            // 200 bytes costs four wave steps, not 200 dependent round trips of its one lane.  Owner of byte x = the lane whose
            // running length first exceeds x (binary search over the lanes' inclusive prefix sums, 6 shuffles).
            const u32 glen = go ? len : 0u;
            const u32 incl = wave_incl_scan_u32(glen), excl = incl - glen;
            const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
            for (u32 base = 0; base < total; base += 64u) {
                const u32 x = base + (u32)lane;
                u32 o = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const u32 v = (u32)__shfl((int)incl, (int)(o + step - 1));
                    if (v <= x) o += step;             // o + step <= 64 always: o grows by distinct powers of two below 64
                }
                const u32 oc = min(o, 63u);
                const u32 po = (u32)__shfl((int)pos, (int)oc), dd = (u32)__shfl((int)dist, (int)oc), ex = (u32)__shfl((int)excl, (int)oc),
                          ln = (u32)__shfl((int)len, (int)oc);
                if (x < total) {
                    const u32 k = x - ex;
                    const u32 sk = dd >= ln ? k : k % dd;      // overlapping match: its first `dist` bytes are final, the rest repeats them
                    dst[po + k] = __hip_atomic_load(dst + po - dd + sk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            todo &= ~__ballot(go);
        }
    }
}

// k_lz_stage (the default from round 5; k_lz_resolve above = COVERM_LZ_V=1, the second implementation the tests compare with).
// k_lz_resolve pays a global store -> drain -> load round trip for every LEVEL of dependency inside a batch of 64 tokens — ~940 rounds per
// block, 70 % of its wave time waiting (profiles/r05_ingest_pmc_summary.json) — although the dependencies that force a new round are
// between tokens a few hundred bytes apart.  Here a batch is the run of consecutive tokens whose output fits LZ_SPAN bytes; that span of the
// block is staged in LDS (2 KiB per wave: the occupancy stays), and
//   phase A  every token whose source lies entirely IN FRONT of the span (final bytes: earlier batches, literals) is copied global -> LDS,
//            all of them at once, their bytes concatenated over the lanes as in k_lz_resolve: ONE round trip per batch;
//   phase B  the tokens whose source lies inside the span are executed in order, one after the other, each by the whole wave (a lane per
//            byte, LDS -> LDS): an LDS instruction of a wave sees what the wave's earlier LDS instructions wrote, so there is no
//            dependency analysis and nothing to drain; a token costs ~15 instructions, most of them scalar (its fields come through
//            v_readlane), an overlapping match (distance < length) reads its first `distance` bytes repeatedly (final before it begins);
//   then the span goes back to global memory with aligned dword stores (byte stores at its two ends: the bytes next to it may be another
//   wave's).  Three round trips per batch, ~90 batches per block, instead of ~940.
constexpr u32 LZ_SPAN = 2048;
constexpr u32 LZ_LDS = LZ_SPAN + 16;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void k_lz_stage(const BgzfBlock *__restrict__ blocks, u32 n_blocks, uint8_t *__restrict__ out,
                                                  const tokpos_t *__restrict__ tok, const u32 *__restrict__ n_tok) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[4][LZ_LDS];
    const int lane = threadIdx.x & 63;
    uint8_t *L = lds_all[threadIdx.x >> 6];
    const u32 b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const u32 nt = n_tok[b];
    if (nt == 0u) return;
    uint8_t *dst = out + blocks[b].out_off;
    const tokpos_t *my = tok + (size_t)b * INF_TOK_CAP;
    // A token's fields: its position from the block's list, its (length, distance) from the three bytes k_inflate left at that position.
    // Both are two dependent trips to memory, so the NEXT 64 tokens' fields are fetched while a batch is worked on (they lie behind the
    // batch's span: nothing the batch writes can change them), and a batch begins with what the previous one fetched.
    // (in two steps, each issued where the loads in front of it are waited for anyway: positions at the top of a batch, the token bytes once
    // the span is staged)
    auto fetch_pos = [&](u32 first) { const u32 t = first + (u32)lane; return t < nt ? (u32)my[t] : 0xffffffffu; };
    auto fetch_tok = [&](u32 p) {
        u32 w = 0;
        if (p != 0xffffffffu) { const uint8_t *tp = dst + p; w = (u32)tp[0] | ((u32)tp[1] << 8) | ((u32)tp[2] << 16); }    // written by k_inflate (the kernel before this one)
        return w;
    };
    u32 t0 = 0, pos = fetch_pos(0u), t24 = fetch_tok(pos);
    while (t0 < nt) {
        const u32 npos = fetch_pos(t0 + 64u);
        const bool have = t0 + (u32)lane < nt;
        const u32 len = have ? ((t24 >> 15) & 0xffu) + 3u : 0u, dist = have ? (t24 & 0x7fffu) + 1u : 1u;
        const u32 dst_end = pos + len;
        // the span: from the first token's first byte (moved down to a 4-byte boundary of the ADDRESS, so that the staging loads and the
        // write-back stores are aligned dwords) over as many consecutive tokens as end inside LZ_SPAN bytes; offsets are relative to the
        // block and may be negative by up to 3 at its very beginning (bytes of the block in front: read, never written)
        const int P0 = (int)__builtin_amdgcn_readfirstlane((int)pos);
        const int mis = (int)((uintptr_t)(dst + P0) & 3u);
        const int P0a = P0 - mis;
        const u64 fm = __ballot(have && (int)dst_end <= P0a + (int)LZ_SPAN);
        const u32 n_take = ~fm ? (u32)__builtin_ctzll(~fm) : 64u;        // tokens' ends increase with the lane: the ones that fit are the first n_take (>= 1: a match is <= 258 bytes)
        const bool mine = (u32)lane < n_take;
        const int P1 = (int)__builtin_amdgcn_readlane((int)dst_end, (int)n_take - 1);
        const u32 span = (u32)(P1 - P0a);
        // ---- stage the span: aligned dwords, read at agent scope (the dword in front of P0 may hold bytes this wave's previous batch wrote)
#pragma unroll 2
        for (u32 o = (u32)lane * 4u; o < span; o += 256u)
            *reinterpret_cast<u32 *>(L + o) = __hip_atomic_load(reinterpret_cast<const u32 *>(dst + P0a + (int)o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u32 nt24 = fetch_tok(npos);
        // ---- phase A: sources entirely in front of the span, all at once (their bytes are final)
        const int src = (int)pos - (int)dist;
        const bool far = mine && src + (int)len <= P0a;
        {
            const u32 glen = far ? len : 0u;
            const u32 incl = wave_incl_scan_u32(glen), excl = incl - glen;
            const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
            for (u32 base = 0; base < total; base += 64u) {
                const u32 x = base + (u32)lane;
                u32 o = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const u32 v = (u32)__shfl((int)incl, (int)(o + step - 1));
                    if (v <= x) o += step;
                }
                const u32 oc = min(o, 63u);
                const u32 po = (u32)__shfl((int)pos, (int)oc), so = (u32)__shfl(src, (int)oc), ex = (u32)__shfl((int)excl, (int)oc);
                if (x < total) {
                    const u32 k = x - ex;
                    L[(int)(po + k) - P0a] = __hip_atomic_load(dst + so + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // ---- phase B: the others in order, each by the whole wave
        for (u64 todo = __ballot(mine && !far); todo != 0; todo &= todo - 1) {
            const int j = __builtin_ctzll(todo);
            const int pj = (int)__builtin_amdgcn_readlane((int)pos, j), lj = (int)__builtin_amdgcn_readlane((int)len, j), dj = (int)__builtin_amdgcn_readlane((int)dist, j);
            const int sj = pj - dj;
            uint8_t *D = L + (pj - P0a);
            if (sj >= P0a) {                     // the source lies inside the span (or in the staged bytes in front of it)
                const uint8_t *S = L + (sj - P0a);
                if (dj >= lj) {
#pragma unroll 1
                    for (int k = lane; k < lj; k += 64) D[k] = S[k];
                } else {                         // overlapping: the first dj bytes, final before the match begins, repeat
                    const float inv = 1.0f / (float)dj;
#pragma unroll 1
                    for (int k = lane; k < lj; k += 64) {
                        int q = (int)((float)k * inv);
                        int r = k - q * dj;
                        r = r < 0 ? r + dj : (r >= dj ? r - dj : r);
                        D[k] = S[r];
                    }
                }
            } else {                             // the source begins in front of the span and reaches into it (rare): byte by byte from where it lies
#pragma unroll 1
                for (int k = lane; k < lj; k += 64) {
                    const int r = dj >= lj ? k : k % dj;
                    const int sp = sj + r;
                    D[k] = sp >= P0a ? L[sp - P0a] : __hip_atomic_load(dst + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // ---- the span goes back: bytes [P0, P1), aligned dwords in the middle
        {
            const u32 n = (u32)(P1 - P0);
            const u32 head = min(mis ? 4u - (u32)mis : 0u, n);
            const u32 nd = (n - head) >> 2, tail = (n - head) & 3u;
            if ((u32)lane < head) dst[P0 + lane] = L[mis + lane];
            uint8_t *g = dst + P0 + (int)head;
            const uint8_t *l = L + mis + head;
#pragma unroll 2
            for (u32 i = (u32)lane; i < nd; i += 64u) *reinterpret_cast<u32 *>(g + 4u * i) = *reinterpret_cast<const u32 *>(l + 4u * i);
            if ((u32)lane < tail) g[4u * nd + (u32)lane] = l[4u * nd + (u32)lane];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the next batch stages (and phase A reads) at agent scope what this one wrote
        t0 += n_take;
        if (n_take == 64u) { pos = npos; t24 = nt24; }
        else {     // the tokens that did not fit stay, the fetched ones move up behind them
            const int from = lane + (int)n_take;
            const u32 a = (u32)__shfl((int)pos, from & 63), c = (u32)__shfl((int)t24, from & 63);
            const u32 d = (u32)__shfl((int)npos, from & 63), e = (u32)__shfl((int)nt24, from & 63);
            pos = from < 64 ? a : d; t24 = from < 64 ? c : e;
        }
    }
}

// (A SLIDING window of 64 consecutive tokens — token k held by lane k mod 64, the window following the block's lowest unfinished token —
// cuts the rounds from ~940 to ~660 per block and was slower on the device, 25.2 ms per round against 19.4: the dependency search then
// runs every round instead of once per batch, and about half of this kernel's time is instruction issue, not latency.
// profiles/r04_lz_slide.log.)
// CRC-32 of every inflated block against its BGZF trailer (htslib checks it in bgzf.c:inflate_block): one lane per block,
// slicing-by-4 tables in LDS (4 bytes per dependent step).  Only 4 KiB of LDS per workgroup, so the CU is full of waves and the
// per-step LDS latency overlaps.
__global__ __launch_bounds__(256) void k_crc32(const BgzfBlock *__restrict__ blocks, u32 n_blocks, const uint8_t *__restrict__ out,
                                               u32 *__restrict__ status, u32 *__restrict__ n_failed) {
    // slicing-by-8 (round 5; by-4 before): eight bytes per dependent step and per load — a lane's 64 KiB are 8 k steps instead of 16 k, and a
    // wave's load, which touches 64 different cache lines whatever its width, is issued half as often
    __shared__ u32 T[8][256];
    for (int i = threadIdx.x; i < 256; i += 256) {
        u32 c = (u32)i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        T[0][i] = c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 256) {
        u32 c = T[0][i];
        for (int k = 1; k < 8; k++) { c = (c >> 8) ^ T[0][c & 0xffu]; T[k][i] = c; }
    }
    __syncthreads();
    const u32 b = blockIdx.x * 256u + threadIdx.x;
    if (b >= n_blocks) return;
    if (status[b] != INF_OK) return;
    const BgzfBlock B = blocks[b];
    const uint8_t *p = out + B.out_off;
    u32 n = B.isize, crc = 0xffffffffu;
    while (n && ((u64)p & 7u)) { crc = T[0][(crc ^ *p++) & 0xffu] ^ (crc >> 8); n--; }
    const uint2 *w = (const uint2 *)p;
    for (; n >= 8u; n -= 8u) {
        const uint2 v = *w++;
        const u32 a = crc ^ v.x, d = v.y;
        crc = T[7][a & 0xffu] ^ T[6][(a >> 8) & 0xffu] ^ T[5][(a >> 16) & 0xffu] ^ T[4][a >> 24] ^
              T[3][d & 0xffu] ^ T[2][(d >> 8) & 0xffu] ^ T[1][(d >> 16) & 0xffu] ^ T[0][d >> 24];
    }
    p = (const uint8_t *)w;
    while (n) { crc = T[0][(crc ^ *p++) & 0xffu] ^ (crc >> 8); n--; }
    if ((crc ^ 0xffffffffu) != B.crc) { status[b] = INF_ERR_CRC; atomicAdd(n_failed, 1u); }
}

// (One WAVE per block — every lane 1 KiB, the 64 pieces combined through the CRC's linearity, crc(A || B) = crc(A) * x^(8 |B|) mod P xor
// crc(B) with the 64 powers as compile-time constants — was built and measured: 5.6 ms per full round against 2.8 (0.8 against 1.9 for a
// 12 k-block round).  Either way a wave's load touches 64 different cache lines, and that, not the length of a lane's chain, is what the
// kernel's time is made of.  profiles/r04_crc_wave.log.)
// k_crc32_wave (round 6, the default): ONE WAVE per block, the block a matrix of 8-byte words with 64 columns, lane l owning column l — every
// load of the wave is 512 consecutive bytes — and the columns' registers joined by a scan over the lanes.  The arithmetic (tables for a
// distance of 512 bytes, the scan's operator, the unaligned head, the tail) is csrc/crc_wave_core.h, which runs lane by lane on the CPU
// against zlib in tests/test_crc_wave_core.py.  Workgroups of eight waves stay resident and stride over the window's blocks, so the
// 36 KiB of tables (built on the host, crcw::build_tables) come into LDS once per workgroup.
#define CRCW_FN __host__ __device__ __forceinline__
#include "crc_wave_core.h"

__device__ __forceinline__ u32 crc_lane_scan(const u32 *T, u32 s, int lane) {
#pragma unroll
    for (u32 m = 0; m < 6u; m++) {
        u32 left = (u32)__shfl_up((int)s, 1u << m);
        if (lane < (int)(1u << m)) left = 0u;
        s = crcw::scan_combine(T, m, left, s);
    }
    return s;
}

__global__ __launch_bounds__(512) void k_crc32_wave(const BgzfBlock *__restrict__ blocks, u32 n_blocks, const uint8_t *__restrict__ out,
                                                    u32 *__restrict__ status, u32 *__restrict__ n_failed, const u32 *__restrict__ tables) {
    __shared__ u32 T[crcw::TABLE_WORDS];
    for (u32 i = threadIdx.x; i < crcw::TABLE_WORDS; i += 512u) T[i] = tables[i];
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63u);
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 8u + (threadIdx.x >> 6))), n_waves = gridDim.x * 8u;
    for (u32 b = wave; b < n_blocks; b += n_waves) {
        if (status[b] != INF_OK) continue;
        const BgzfBlock B = blocks[b];
        const uint8_t *p = out + B.out_off;
        const u32 n = B.isize;
        u32 crc;
        if (n < crcw::SMALL) crc = crcw::small_block(T, p, n);
        else {
            const crcw::Shape S = crcw::shape_of(p, n);
            const u64 *g = reinterpret_cast<const u64 *>(S.base);
            u32 C = 0;
            if (S.rows) {
                u32 s = 0;
                u64 w = crcw::fix_word(S, (u32)lane, g[lane]);
                for (u32 j = 1; j < S.rows; j++) {       // the next row's word is on its way while this one is absorbed
                    const u64 wn = g[64u * j + (u32)lane];
                    s = crcw::step(T, crcw::T_LO512, crcw::T_HI512, s, w);
                    w = wn;
                }
                s = crcw::step(T, crcw::T_LO8, crcw::T_HI8, s, w);
                C = (u32)__builtin_amdgcn_readlane((int)crc_lane_scan(T, s, lane), 63);
            }
            if (S.tail_words) {      // the partial row's words on the HIGHEST lanes, the register so far entering with the first of them
                const int lp = lane - (int)(64u - S.tail_words);
                u32 t = 0;
                if (lp >= 0) {
                    const u32 idx = 64u * S.rows + (u32)lp;
                    t = crcw::step(T, crcw::T_LO8, crcw::T_HI8, lp == 0 ? C : 0u, crcw::fix_word(S, idx, g[idx]));
                }
                C = (u32)__builtin_amdgcn_readlane((int)crc_lane_scan(T, t, lane), 63);
            }
            crc = crcw::finish(T, S, C, g[64u * S.rows + S.tail_words]);
        }
        if (crc != B.crc && lane == 0) { status[b] = INF_ERR_CRC; atomicAdd(n_failed, 1u); }
    }
}

// ------------------------------------------------------------------------------------ BAM record parsing on the device
// The inflated stream is parsed window by window (one window = the blocks of one k_inflate round, in its own buffer): the bytes
// of a record cut by the window's end (the "tail") are carried in front of the next window's bytes, so a window always starts
// on a record boundary at *p0 (device memory: it depends on the previous window's parse) and is parsed like a whole file.
struct BamScan {
    const uint8_t *u;      // window buffer: [carry - tail, carry) bytes carried over, [carry, N) this window's inflated blocks
    u64 N;                 // end of the valid bytes
    const u64 *p0;         // device: offset of the first record of this window (written by k_carry_in)
    u64 seg_bytes;         // segment size; segment k covers [k * seg_bytes, (k + 1) * seg_bytes)
    u32 n_seg;
    int n_ref;
    const u32 *ref_len;    // n_ref entries
    int final;             // last window of the file: a cut record is an error instead of a tail
    // tid span (one rank of a span-sharded file): only records with key_lo <= key < key_hi are counted and extracted
    // (key = tid, or 0x7fffffff for records without a reference, which sort last); search_first: the first window starts
    // somewhere inside the file, its first record boundary is searched like any other segment's
    long long key_lo, key_hi;
    int search_first;
};
__device__ __forceinline__ long long rec_key(int tid) { return tid < 0 ? 0x7fffffffll : (long long)tid; }

__device__ __forceinline__ u32 ld32(const uint8_t *p) {   // unaligned little-endian load
    const u64 a = (u64)p;
    const u32 *w = (const u32 *)(a & ~3ull);
    const u32 sh = (u32)(a & 3u) * 8u;
    if (sh == 0) return w[0];
    return (w[0] >> sh) | (w[1] << (32u - sh));
}
__device__ __forceinline__ u32 ld16(const uint8_t *p) { return (u32)p[0] | ((u32)p[1] << 8); }

// Can a record start at offset o?  Returns its end offset or 0.  Same tests as csrc/host_bam.cpp plausible_record.
__device__ __forceinline__ u64 plausible_record(const BamScan &S, u64 o) {
    if (o + 36 > S.N) return 0;
    const uint8_t *r = S.u + o;
    const u32 bs = ld32(r);
    if (bs < 32u || o + 4 + (u64)bs > S.N) return 0;
    const int rid = (int)ld32(r + 4), pos = (int)ld32(r + 8), nrid = (int)ld32(r + 24);
    const u32 lname = r[12], ncig = ld16(r + 16), lseq = ld32(r + 20);
    if (rid < -1 || rid >= S.n_ref || nrid < -1 || nrid >= S.n_ref || pos < -1 || lname == 0u) return 0;
    if (rid >= 0 && (u32)pos > S.ref_len[rid]) return 0;
    const u64 fixed = 32ull + lname + 4ull * ncig + ((u64)lseq + 1ull) / 2 + lseq;
    if (fixed > bs) return 0;
    if (r[36 + lname - 1] != 0) return 0;
    // The aux area must parse as tags that end exactly at the record end.  Without this test a stray offset whose "block_size"
    // happens to jump onto a real record start far away passes (the next seven records are then real ones): seen once per
    // ~10^7 tested offsets on a 6 GB stream, where it cost a whole-file fallback to the CPU reader.
    const uint8_t *p = r + 4 + fixed, *end = r + 4 + bs;
    for (int t = 0; t < 32 && p < end; t++) {
        if (p + 3 > end) return 0;
        const uint8_t ty = p[2];
        p += 3;
        if (ty == 'A' || ty == 'c' || ty == 'C') p += 1;
        else if (ty == 's' || ty == 'S') p += 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') p += 4;
        else if (ty == 'Z' || ty == 'H') { int g = 0; while (p < end && *p && g < 4096) { p++; g++; } if (p >= end || *p) return 0; p++; }
        else if (ty == 'B') {
            if (p + 5 > end) return 0;
            const uint8_t st = p[0];
            if (st != 'c' && st != 'C' && st != 's' && st != 'S' && st != 'i' && st != 'I' && st != 'f') return 0;
            const u32 cnt = ld32(p + 1);
            const u32 es = (st == 'c' || st == 'C') ? 1u : (st == 's' || st == 'S') ? 2u : 4u;
            if ((u64)(end - p - 5) / es < cnt) return 0;
            p += 5 + (u64)cnt * es;
        } else return 0;
        if (p > end) return 0;
    }
    return o + 4 + bs;
}

struct SegInfo {
    u64 start;     // first record start found in the segment (~0 = none)
    u64 landed;    // where the hop from `start` ended (first record start at or beyond the segment end, or N)
    u32 n_rec, n_cig;
    u32 flags;     // bit 1: keys decrease inside the hop (bit 0 was: a CG:B,I record, now resolved by the extraction)
    u32 first_key, last_key;     // keys of the first / last record hopped over (any record, not only the span's); valid when n_hop != 0
    u32 n_hop;                   // records hopped over
    // The hop's chain cut in EXT_PARTS pieces for the extraction (a lane per piece instead of a lane per segment: k_bam_extract is a chain of
    // ~37 dependent memory instructions per record and spent 98 % of its wave time waiting at ten waves per CU): piece p + 1 begins at the first
    // record that starts at or beyond start + (p + 1) * EXT_PIECE bytes — part_off[p] bytes behind `start` — with part_rec[p] / part_cig[p]
    // of the span's records / CIGAR words in front of it.  Pieces the chain never reaches are empty (part_off = landed - start).
    u32 part_off[3], part_rec[3], part_cig[3];
};
constexpr u32 EXT_PARTS = 4, EXT_PIECE = 8192;

// One wave per segment: lanes test consecutive offsets for an 8-record plausible chain; the lowest hit wins.  The segment that
// holds *p0 starts there by definition; segments below it are dead.
// A record whose CIGAR has more than 65535 operations stores `<l_seq>S<ref_len>N` in the CIGAR field and the real one in CG:B,I; htslib
// swaps it back in while reading (bam_tag2cigar, sam.c), so record.cigar() at contig.rs:168 sees the real CIGAR.  The conditions of
// csrc/host_bam.cpp real_cigar_from_cg: exactly two operations, mapped, the first one `<l_seq>S`, the first CG tag of type B,I / B,i
// with at least as many words as the placeholder.  Returns the words and their count, or nullptr.
__device__ __forceinline__ const uint8_t *cg_cigar(const uint8_t *r, const uint8_t *end, u32 &cnt) {
    const u32 n_cig = ld16(r + 16), l_read_name = r[12], l_seq = ld32(r + 20);
    if (n_cig != 2u || (int)ld32(r + 4) < 0 || (int)ld32(r + 8) < 0) return nullptr;
    const uint8_t *c = r + 36 + l_read_name;
    if (c + 8 > end) return nullptr;
    const u32 c0 = ld32(c);
    if ((c0 & 15u) != 4u || (c0 >> 4) != l_seq) return nullptr;
    const uint8_t *p = c + 8 + ((u64)l_seq + 1ull) / 2 + l_seq;
    if (p > end) return nullptr;
    while (p + 3 <= end) {
        const bool is_cg = p[0] == 'C' && p[1] == 'G';
        const uint8_t t = p[2];
        p += 3;
        u32 sz;
        if (t == 'A' || t == 'c' || t == 'C') sz = 1;
        else if (t == 's' || t == 'S') sz = 2;
        else if (t == 'i' || t == 'I' || t == 'f') sz = 4;
        else if (t == 'Z' || t == 'H') {
            while (p < end && *p) p++;
            if (p >= end) return nullptr;
            p++;
            continue;
        } else if (t == 'B') {
            if (p + 5 > end) return nullptr;
            const uint8_t st = p[0];
            const u32 n = ld32(p + 1);
            const u32 es = (st == 'c' || st == 'C') ? 1u : (st == 's' || st == 'S') ? 2u : 4u;
            if ((u64)(end - p - 5) / es < n) return nullptr;
            if (is_cg && (st == 'I' || st == 'i')) { cnt = n; return (n >= n_cig && n < (1u << 29)) ? p + 5 : nullptr; }
            p += 5 + (u64)n * es;
            continue;
        } else return nullptr;
        if (p + sz > end) return nullptr;
        p += sz;
    }
    return nullptr;
}

__global__ __launch_bounds__(256) void k_bam_find(BamScan S, SegInfo *__restrict__ seg) {
    const int lane = threadIdx.x & 63;
    const u32 k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= S.n_seg) return;
    const u64 p0 = *S.p0;
    const u64 lo = (u64)k * S.seg_bytes, hi = min(S.N, lo + S.seg_bytes);
    u64 found = ~0ull;
    if (hi <= p0) {}
    else if (lo <= p0 && !S.search_first) found = p0 < S.N ? p0 : ~0ull;
    else {
        for (u64 base = max(lo, p0); base < hi && found == ~0ull; base += 64) {
            const u64 o = base + (u64)lane;
            bool ok = false;
            if (o < hi) {
                u64 q = o; int chain = 0;
                while (chain < 8) { const u64 e = plausible_record(S, q); if (!e) break; q = e; chain++; if (q == S.N) break; }
                ok = chain == 8 || (chain > 0 && q == S.N);
            }
            const u64 m = __ballot(ok);
            if (m) found = base + (u64)(__ffsll((long long)m) - 1);
        }
    }
    if (lane == 0) { SegInfo s; s.start = found; s.landed = 0; s.n_rec = 0; s.n_cig = 0; s.flags = 0; s.first_key = s.last_key = 0; s.n_hop = 0; seg[k] = s; }
}

// One lane per segment that has a start: hop to the first record at or beyond the next live segment's start (bounded by the
// segment end of the LAST segment before that one), counting records and CIGAR words.
__global__ __launch_bounds__(64) void k_bam_hop(BamScan S, SegInfo *__restrict__ seg) {
    const u32 k = blockIdx.x * 64u + threadIdx.x;
    if (k >= S.n_seg) return;
    SegInfo s = seg[k];
    if (s.start == ~0ull) return;
    // the hop ends at the start found by the next segment that has one
    u64 limit = S.N;
    for (u32 j = k + 1; j < S.n_seg; j++) { const u64 st = seg[j].start; if (st != ~0ull) { limit = st; break; } }
    u64 q = s.start; u32 nr = 0, nc = 0, fl = 0, nh = 0, k_first = 0, k_last = 0;
    u32 part = 0;        // pieces closed so far
    while (q < limit && q + 4 <= S.N) {
        while (part < EXT_PARTS - 1u && q >= s.start + (u64)(part + 1u) * EXT_PIECE) { s.part_off[part] = (u32)(q - s.start); s.part_rec[part] = nr; s.part_cig[part] = nc; part++; }
        const uint8_t *r = S.u + q;
        const u32 bs = ld32(r);
        if (bs < 32u || q + 4 + (u64)bs > S.N) break;
        const u32 ncig = ld16(r + 16);
        const long long key = rec_key((int)ld32(r + 4));
        if (nh == 0u) k_first = (u32)key; else if ((u32)key < k_last) fl |= 2u;
        k_last = (u32)key; nh++;
        if (key >= S.key_lo && key < S.key_hi) {
            u32 words = ncig;
            if (ncig == 2u) {    // `<l_seq>S <n>N` may be the placeholder of a CIGAR stored in CG:B,I (SAM spec 4.2.2): the store takes the real one
                u32 cnt = 0;
                if (cg_cigar(r, r + 4 + bs, cnt) != nullptr) words = cnt;
            }
            nr++; nc += words;
        }
        q += 4 + (u64)bs;
    }
    for (; part < EXT_PARTS - 1u; part++) { s.part_off[part] = (u32)(q - s.start); s.part_rec[part] = nr; s.part_cig[part] = nc; }
    s.landed = q; s.n_rec = nr; s.n_cig = nc; s.flags = fl; s.first_key = k_first; s.last_key = k_last; s.n_hop = nh;
    seg[k] = s;
}

// Single workgroup: chain verification + exclusive scans over segments (records, CIGAR words) + the tail.
// result[0] = records of this window, [1] = CIGAR words, [2] = status (0 ok, 1 chain mismatch, 2 truncated, 4 needs the CPU
// reader, 8 tail larger than the carry buffer, 64 record keys (tid) decrease somewhere in the window — looked at for spans
// only: a span trusts the file's order when it drops its neighbours' records, the whole-file path reports disorder from
// cov_finish in file order with the other per-record errors), [3] = offset where the tail starts (first byte no record of this window owns),
// [4..6] = first bad segment / its start / where its hop landed.  The tail [result[3], N) goes to `carry`, its length to *carry_len.
__global__ __launch_bounds__(1024) void k_bam_verify(BamScan S, SegInfo *__restrict__ seg, u64 *__restrict__ rec_base, u64 *__restrict__ cig_base,
                                                     u64 *__restrict__ result, uint8_t *__restrict__ carry, u64 carry_cap, u64 *__restrict__ carry_len,
                                                     u64 *__restrict__ prev_key) {
    __shared__ u64 s_rec[1024], s_cig[1024];
    __shared__ u32 s_kfirst[1024], s_klast[1024], s_khave[1024];
    __shared__ u32 s_bad;
    __shared__ u64 s_tail, s_badk;
    const u32 t = threadIdx.x;
    if (t == 0) { s_bad = 0; s_tail = min(*S.p0, S.N); s_badk = ~0ull; }    // no live segment: everything from p0 on is tail
    __syncthreads();
    const u32 per = (S.n_seg + 1023u) / 1024u;
    const u32 k0 = t * per, k1 = min(S.n_seg, k0 + per);
    u64 nr = 0, nc = 0; u32 bad = 0;
    u32 kf = 0, kl = 0, khave = 0;      // first / last key over this thread's segments
    for (u32 k = k0; k < k1; k++) {
        const SegInfo s = seg[k];
        if (s.start == ~0ull) continue;
        if (s.n_hop) {
            if (s.flags & 2u) bad |= 64u;
            if (khave && s.first_key < kl) bad |= 64u;
            if (!khave) kf = s.first_key;
            kl = s.last_key; khave = 1u;
        }
        u64 want = S.N; bool last_live = true;
        for (u32 j = k + 1; j < S.n_seg; j++) { const u64 st = seg[j].start; if (st != ~0ull) { want = st; last_live = false; break; } }
        if (last_live) {
            s_tail = s.landed;                                  // exactly one segment is the last live one
            if (S.final && s.landed != S.N) { bad |= 2u; atomicMin(&s_badk, (u64)k); }
        } else if (s.landed != want) { bad |= 1u; atomicMin(&s_badk, (u64)k); }
        nr += s.n_rec; nc += s.n_cig;
    }
    if (bad) atomicOr(&s_bad, bad);
    s_rec[t] = nr; s_cig[t] = nc; s_kfirst[t] = kf; s_klast[t] = kl; s_khave[t] = khave;
    __syncthreads();
    const u64 tail = min(s_tail, S.N), tail_len = S.N - tail;
    const bool tail_fits = tail_len <= carry_cap;
    if (t == 0) {
        u64 a = 0, c = 0;
        for (u32 i = 0; i < 1024; i++) { const u64 x = s_rec[i], y = s_cig[i]; s_rec[i] = a; s_cig[i] = c; a += x; c += y; }
        u32 st = s_bad;
        {   // order across the threads' runs of segments, and against the last key of the previous window (*prev_key, bit 63 = valid)
            const u64 pk = *prev_key;
            u32 prev = (u32)pk, have = (u32)(pk >> 63);
            for (u32 i = 0; i < 1024; i++) if (s_khave[i]) { if (have && s_kfirst[i] < prev) st |= 64u; prev = s_klast[i]; have = 1u; }
            *prev_key = (u64)prev | ((u64)have << 63);
        }
        if (S.final && tail_len != 0) st |= 2u;
        if (!tail_fits) st |= 8u;
        result[0] = a; result[1] = c; result[2] = st; result[3] = tail;
        const u64 kb = s_badk;
        result[4] = kb;
        if (kb < S.n_seg) { result[5] = seg[kb].start; result[6] = seg[kb].landed; }
        // key of the record the tail begins with (~0 when there is no tail or its tid is not here yet): a span's reader checks
        // that the record cut by the end of its bytes lies beyond its key range
        result[7] = (tail_len >= 8) ? (u64)rec_key((int)ld32(S.u + tail + 4)) : ~0ull;
        *carry_len = tail_fits ? tail_len : 0ull;
    }
    __syncthreads();
    u64 a = s_rec[t], c = s_cig[t];
    for (u32 k = k0; k < k1; k++) {
        rec_base[k] = a; cig_base[k] = c;
        if (seg[k].start != ~0ull) { a += seg[k].n_rec; c += seg[k].n_cig; }
    }
    if (tail_fits) for (u64 i = t; i < tail_len; i += 1024) carry[i] = S.u[tail + i];
}

// The previous window's tail goes in front of this window's bytes; *p0 = where this window's first record starts.
// `first_off` = offset of the first record behind the BAM header (window 0 only).
__global__ __launch_bounds__(1024) void k_carry_in(uint8_t *__restrict__ win, u64 carry_area, const uint8_t *__restrict__ carry,
                                                   const u64 *__restrict__ carry_len, u64 first_off, u64 *__restrict__ p0) {
    const u64 len = *carry_len;
    uint8_t *d = win + carry_area - len;
    for (u64 i = threadIdx.x; i < len; i += 1024) d[i] = carry[i];
    if (threadIdx.x == 0) *p0 = carry_area - len + first_off;
}

struct RecStore {
    int32_t *tid, *pos;
    uint16_t *flag;
    uint8_t *mapq, *nm_kind;
    u32 *nm, *l_seq, *cigar_off, *cigar;
    u64 rec0, cig0;    // where this file's records / CIGAR words start in the store
    // mates (pair-mode reader filter, csrc/pair_kernels.hip.h): next_refID and a 96-bit hash of the read name; null = not kept
    int32_t *mtid; u64 *qh1; u32 *qh2;
};

// Linear aux scan for NM (csrc/host_bam.cpp scan_aux without the CG part).
__device__ __forceinline__ u32 scan_nm(const uint8_t *p, const uint8_t *end, u32 &nm) {
    while (p + 3 <= end) {
        const bool is_nm = p[0] == 'N' && p[1] == 'M';
        const uint8_t t = p[2];
        p += 3;
        u32 sz;
        if (t == 'A' || t == 'c' || t == 'C') sz = 1;
        else if (t == 's' || t == 'S') sz = 2;
        else if (t == 'i' || t == 'I' || t == 'f') sz = 4;
        else if (t == 'Z' || t == 'H') {
            if (is_nm) return 2u;
            while (p < end && *p) p++;
            if (p >= end) return 0u;
            p++;
            continue;
        } else if (t == 'B') {
            if (p + 5 > end) return 0u;
            const uint8_t st = p[0];
            const u32 cnt = ld32(p + 1);
            const u32 es = (st == 'c' || st == 'C') ? 1u : (st == 's' || st == 'S') ? 2u : 4u;
            if (is_nm) return 2u;
            if ((u64)(end - p - 5) / es < cnt) return 0u;
            p += 5 + (u64)cnt * es;
            continue;
        } else return 0u;
        if (p + sz > end) return 0u;
        if (is_nm) {
            if (t == 'C') { nm = p[0]; return 1u; }
            if (t == 'S') { nm = ld16(p); return 1u; }
            if (t == 'I') { nm = ld32(p); return 1u; }
            return 2u;
        }
        p += sz;
    }
    return 0u;
}

// One lane per segment: second hop; every record's fields go straight into the record store.
__global__ __launch_bounds__(64) void k_bam_extract(BamScan S, const SegInfo *__restrict__ seg, const u64 *__restrict__ rec_base,
                                                    const u64 *__restrict__ cig_base, RecStore R, u32 *__restrict__ n_bad, u32 parts) {
    // one lane per piece of a segment's chain (parts = EXT_PARTS), or per segment (parts = 1: the pieces walked by one lane)
    const u32 kp = blockIdx.x * 64u + threadIdx.x, k = kp / parts, part = parts == 1u ? 0u : kp % parts;
    if (k >= S.n_seg) return;
    const SegInfo s = seg[k];
    if (s.start == ~0ull) return;
    const bool last = parts == 1u || part + 1u == EXT_PARTS;
    const u64 q_end = last ? s.landed : s.start + s.part_off[part];
    const u32 rec_lo = part ? s.part_rec[part - 1u] : 0u, cig_lo = part ? s.part_cig[part - 1u] : 0u;
    const u32 n_mine = (last ? s.n_rec : s.part_rec[part]) - rec_lo;
    u64 q = part ? s.start + s.part_off[part - 1u] : s.start;
    u64 ri = R.rec0 + rec_base[k] + rec_lo, ci = R.cig0 + cig_base[k] + cig_lo;
    // A lane's records are consecutive in the store, so their fields wait in LDS four records at a time (96 bytes per lane; registers
    // would have to be indexed by a counter: 175 of them, two waves per SIMD, in a kernel that lives on its occupancy) and every column
    // leaves as ONE store per four records (16 bytes of tid / pos / l_seq / cigar_off / nm, 8 of flags, 4 of mapq and of the NM kind)
    // instead of four partial-line stores each: the lanes of a wave write ~100 bytes apart, and a line that is written four bytes at a
    // time is evicted several times before it is complete (5.0 GB of WRITE_SIZE for 0.9 GB of store output in round 4's counters).
    // Mate columns (pair-mode filters only) are written one record at a time as before.
    // The store's columns are addressed at an arbitrary record index, so the wide stores below are only as aligned as the column's element type
    // (4, 2, 1 bytes): they go through vector types DECLARED with that alignment (gfx950 global memory takes unaligned accesses; the
    // compiler must not be told 16).
    typedef u32 u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef u32 u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));
    typedef u32 u32_a1 __attribute__((aligned(1)));
    __shared__ __attribute__((aligned(16))) u32 lbuf[64][32];      // per lane: tid[4] pos[4] l_seq[4] cigar_off[4] nm[4] flag[4 x u16] mapq[4 x u8] nm_kind[4 x u8] | 8 CIGAR words
    u32 *B = lbuf[threadIdx.x & 63];
    u32 *Cw = B + 24;          // CIGAR words of the lane's records, consecutive in the store like the records: they leave four at a time too
    u32 ncw = 0;               // words waiting in Cw; they belong at cigar[ci - ncw ..)
    auto flush_cigar = [&](u64 ci_now, bool all) {
        u64 at = ci_now - ncw;
        u32 done = 0;
        while (ncw - done >= 4u) { const u32x4_a4 v = {Cw[done], Cw[done + 1], Cw[done + 2], Cw[done + 3]}; *reinterpret_cast<u32x4_a4 *>(R.cigar + at + done) = v; done += 4u; }
        if (all) { for (; done < ncw; done++) R.cigar[at + done] = Cw[done]; ncw = 0; }
        else { const u32 left = ncw - done; for (u32 x = 0; x < left; x++) Cw[x] = Cw[done + x]; ncw = left; }
    };
    uint16_t *Bf = reinterpret_cast<uint16_t *>(B + 20);
    uint8_t *Bm = reinterpret_cast<uint8_t *>(B + 22), *Bk = reinterpret_cast<uint8_t *>(B + 23);
    u32 nb = 0;
    auto flush = [&](u64 at) {       // the buffered records are records at .. at + nb - 1 of the store
        if (nb == 4u) {
            const u32x4_a4 *B4 = reinterpret_cast<const u32x4_a4 *>(B);      // (the LDS side IS 16-byte aligned)
            *reinterpret_cast<u32x4_a4 *>(R.tid + at) = B4[0];
            *reinterpret_cast<u32x4_a4 *>(R.pos + at) = B4[1];
            *reinterpret_cast<u32x4_a4 *>(R.l_seq + at) = B4[2];
            *reinterpret_cast<u32x4_a4 *>(R.cigar_off + at) = B4[3];
            *reinterpret_cast<u32x4_a4 *>(R.nm + at) = B4[4];
            { const u32x2_a2 v = {B[20], B[21]}; *reinterpret_cast<u32x2_a2 *>(R.flag + at) = v; }
            *reinterpret_cast<u32_a1 *>(R.mapq + at) = B[22];
            *reinterpret_cast<u32_a1 *>(R.nm_kind + at) = B[23];
        } else {
            for (u32 x = 0; x < nb; x++) {
                R.tid[at + x] = (int32_t)B[x]; R.pos[at + x] = (int32_t)B[4 + x]; R.l_seq[at + x] = B[8 + x]; R.cigar_off[at + x] = B[12 + x];
                R.nm[at + x] = B[16 + x]; R.flag[at + x] = Bf[x]; R.mapq[at + x] = Bm[x]; R.nm_kind[at + x] = Bk[x];
            }
        }
        nb = 0;
    };
    for (u32 j = 0; j < n_mine && q < q_end;) {
        const uint8_t *r = S.u + q;
        const u32 bs = ld32(r);
        const uint8_t *end = r + 4 + bs;
        const long long key = rec_key((int)ld32(r + 4));
        if (key < S.key_lo || key >= S.key_hi) { q += 4 + (u64)bs; continue; }
        j++;
        const u32 l_read_name = r[12], n_cig = ld16(r + 16), l_seq = ld32(r + 20);
        B[nb] = ld32(r + 4); B[4 + nb] = ld32(r + 8); Bm[nb] = r[13]; Bf[nb] = (uint16_t)ld16(r + 18); B[8 + nb] = l_seq; B[12 + nb] = (u32)ci;
        if (R.mtid) {      // filter.rs:164-176 needs the mate's reference and the read name (without its NUL)
            R.mtid[ri] = (int32_t)ld32(r + 24);
            u64 k1; u32 k2;
            covp::name_hash(r + 36, l_read_name ? l_read_name - 1u : 0u, k1, k2);
            R.qh1[ri] = k1; R.qh2[ri] = k2;
        }
        const uint8_t *c = r + 36 + l_read_name;
        const uint8_t *aux = c + 4ull * n_cig + ((u64)l_seq + 1ull) / 2 + l_seq;
        u32 words = n_cig;
        if (aux > end) { atomicAdd(n_bad, 1u); aux = end; if (ncw) flush_cigar(ci, true); }     // (its words are skipped, not written: what waits goes out first)
        else {
            u32 cnt = 0;
            const uint8_t *cg = n_cig == 2u ? cg_cigar(r, end, cnt) : nullptr;      // (the same test k_bam_hop counted the words with)
            if (cg) { words = cnt; c = cg; }
            if (words <= 4u) {         // short reads: the words wait with those of the lane's neighbouring records
                for (u32 x = 0; x < words; x++) Cw[ncw + x] = ld32(c + 4ull * x);
                ncw += words;
                if (ncw >= 4u) flush_cigar(ci + words, false);
            } else {                   // a long CIGAR goes out directly, behind whatever was waiting
                if (ncw) flush_cigar(ci, true);
                for (u32 x = 0; x < words; x++) R.cigar[ci + x] = ld32(c + 4ull * x);
            }
        }
        u32 nm = 0;
        Bk[nb] = (uint8_t)scan_nm(aux, end, nm);
        B[16 + nb] = nm;
        nb++; ri++; ci += words;
        if (nb == 4u) flush(ri - 4);
        q += 4 + (u64)bs;
    }
    if (nb) flush(ri - nb);
    if (ncw) flush_cigar(ci, true);
}

}  // namespace covi
