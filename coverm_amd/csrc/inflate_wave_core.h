// k_inflate_wave — ONE WAVE PER BGZF BLOCK: the algorithm, written once, run two ways.
//
// Lane-per-block inflate (k_inflate) keeps a private Huffman table per lane in LDS, which bounds the resident lanes, and decodes a
// block serially, which sets the latency of a round.  Here the 64 lanes of a wave share ONE set of tables and each decodes 1/64 of
// a CHUNK of the block's bit stream (share i = bits [cur + i * S, cur + (i + 1) * S); the first DEFLATE block of a payload is one
// chunk, see inflate_block):
//   pass 1  where does the first unit (literal | length + extra + distance + extra | end of block) of share i begin?  Lane i starts
//           OVERLAP bits in front of its share as if a unit began there and decodes up to the share: Huffman streams self-synchronise
//           — on BAM blocks a decoder started at an arbitrary bit is in step with the true sequence after a median of 6 units, 70 at most
//           (profiles/r03_deflate_sync_probe.log) — so where it crosses into its share is, almost always, a true unit boundary;
//   pass 2  lane i decodes exactly the units that begin in its share, from where pass 1 says they begin, and counts output bytes and
//           matches.  Where it ends must be where lane i + 1 began: if not, lane i + 1 takes the end as its beginning and the pass is
//           repeated (lane k is exact after round k whatever pass 1 guessed; normally there is one round).  The lanes up to the first
//           end-of-block symbol are what is left of the block; the rest of the wave decoded what belongs to the next DEFLATE block with
//           the wrong tables and is discarded;
//   pass 3  output offsets = prefix sums of the counts; every lane decodes its units once more and writes: literals at their final
//           positions, a match as a 3-byte token in place + its 16-bit position in the block's token list (k_lz_resolve's contract,
//           unchanged).  No store touches a byte that another lane owns; how the bytes leave is a policy (Sink<1..7>).
// The block header and the code lengths are parsed by lane 0 (a Huffman-coded list is serial); both codes are then built by the whole
// wave — symbols counted and marked per length with LDS atomics, ranked by population count — and the lookup tables filled one INDEX
// per lane step (an index is decoded canonically like a long code), so no lane writes more entries than another.
//
// This file contains no HIP: it is a sequence of `COVW_PARFOR(lane) { ... }` regions over wave-shared state, separated by wave
// barriers.  csrc/ingest_kernels.hip.h instantiates it with lane = the thread and LDS-resident state; tests/c/inflate_wave_host.cpp
// instantiates it with a loop over 64 lanes and checks the bytes against zlib — every line of the algorithm runs on the CPU in the
// test, only the barrier and the bit-reversal intrinsic differ.  The includer defines:
//   COVW_FN                    function qualifier
//   COVW_PARFOR(lane)          `for`-like header: runs the following block for lane = 0 .. 63 (device: once, lane = the thread's lane)
//   COVW_SYNC()                wave barrier + LDS fence (host: nothing)
//   covw_brev32(x)             bit reversal of a 32-bit word
//   COVW_NO_UNROLL             (optional) loop pragma that keeps the following loop rolled
//   COVW_ATOMIC_ADD / _OR(p, v) (optional on the host) atomic update of a wave-shared word
// Statements outside COVW_PARFOR regions are executed by all lanes with identical (wave-uniform) values.
#pragma once
#include <stdint.h>

#ifndef COVW_ATOMIC_ADD      // wave-shared counters and bit sets touched by several lanes of one region (host: the lanes run one after the other)
#define COVW_ATOMIC_ADD(p, v) (*(p) += (v))
#define COVW_ATOMIC_OR(p, v) (*(p) |= (v))
#endif
#ifndef COVW_NO_UNROLL
#define COVW_NO_UNROLL
#endif
#ifndef COVW_TRACE_STORE
#define COVW_TRACE_STORE(ptr, width) do { } while (0)       // tools/proto/wave_cost_model.cpp: store shapes per block
#endif
#ifndef COVW_TRACE_UNIT
#define COVW_TRACE_UNIT(mode, flags) do { } while (0)      // tools/proto/wave_cost_model.cpp: lock-step cost model
#endif

namespace covw {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16;
typedef unsigned char u8;

#ifndef COVW_LB
#define COVW_LB 10      // 2 KiB + 1 KiB of lookup tables: some lane of 64 meets a longer literal code in 67 % of the steps (62 % at 11 bits), a longer
#define COVW_DB 8       // distance code in 6 % (1 % at 9 bits) — profiles/r03_wave_cost_model.log — and the wave state stays at 7.3 KiB: 21 waves per CU
#endif
constexpr u32 LB = COVW_LB, DB = COVW_DB;  // index bits of the lookup tables of the literal/length and the distance alphabet
constexpr u32 TOK_CAP = 21888;            // = covi::INF_TOK_CAP
enum { OK = 0, ERR_FORMAT = 1, ERR_CRC = 2, ERR_SIZE = 3 };       // = covi::INF_*
constexpr u32 NO_EOB = 0xffffffffu;
// A share shorter than the distance over which a decoder falls in step makes pass 1 guess wrong and pass 2 repeat: small blocks use
// fewer lanes.  OVERLAP: p99 of that distance is 50 units (~480 bits), the largest seen 670 bits.
constexpr u32 MIN_SHARE_BITS = 512, OVERLAP_BITS = 768;
constexpr u32 DIST_INVALID = 0x80000000u;

struct Tables {
    union {
        u16 lit[1u << LB];                // bits 0-3 code length (0: not decidable from LB bits); literal: bit 15 = 0, bits 4-11 the byte;
                                          // length symbol: bit 15 = 1, bits 4-11 base - 3, bits 12-14 extra bits (<= 5); extra bits = 7 marks the
                                          // two entries that are neither: 0xf000 end of block, 0xfff0 no symbol (286 / 287, or no code at all)
        u32 mask[16 * 9 + 16];            // while the code is built: the set of symbols of every code length (nine words per literal/length
    };                                    // length, one per distance length) — a symbol's rank among its length = population count below it
    u32 dist[1u << DB];                   // bits 0-3 code length, bits 4-7 extra bits, bits 8-22 base; bit 31: symbols 30 / 31
    u16 lit_limit[16], dist_limit[16];    // limit[l] = (first code of length l + count[l]) << (15 - l)
    u16 lit_off[16], dist_off[16];        // (index of the first symbol of length l in sorted[]) - (first code of length l)  (mod 2^16)
    u16 lit_sorted[288];                  // the symbols sorted by (code length, value), as table entries without the length
    u8 dist_sorted[32];
    u8 lens[320];                         // code lengths: [0, 288) literal/length, [288, 320) distance
};

static_assert(sizeof(u16) << LB >= sizeof(u32) * (16 * 9 + 16), "the symbol sets of the code lengths live where the literal/length table goes");

struct Wave {                             // wave-shared state (LDS on the device)
    Tables T;
    u32 end[64];                          // bit position where lane i's units end = where lane i + 1's begin
    u32 tmp[64];
    u32 flags[64];                        // bit 0: the lane met end-of-block, bit 1: it met an invalid code / ran off the payload
    u32 nbytes[64], ntok[64];             // output bytes / matches of the lane's units
    u32 obase[64], tbase[64];             // exclusive prefix sums of the two
    u32 hdr[8];                           // [0] type, [1] last, [2] hlit, [3] hdist, [4] first unit bit (stored: first data bit), [5] stored length / tokens,
                                          // [6] error, [7] bytes
    u32 changed, n_valid, eob_at, rounds; // rounds: pass-2 rounds of the last Huffman block (statistics)
    u32 n_deflate_blocks, n_chunks, pad_[2];   // DEFLATE blocks in the payload, chunks they were decoded in (statistics)
    u32 cnt[32], start[32];               // symbols per code length and where each length begins in sorted[]: [0, 16) literal/length, [16, 32) distance
    u16 climit[16], coff[16];             // the code-length code ...
    u8 csorted[32], cl[32];
    u8 cltab[128];                        // ... and its lookup table: (symbol << 3) | code length by the next 7 bits, 0 = no code
};

struct Src { const u32 *w; u32 total_bits; };     // aligned words of the payload (readable 64 bytes past its end); bit 0 = bit 0 of w[0]

COVW_FN u32 bits_at(u32 x, u32 off, u32 n) { return (x >> off) & ((1u << n) - 1u); }      // off + n <= 31

// Bit cursor of one lane: `cnt` valid bits of the stream at `pos` in buf, the next word already requested.
struct Cursor {
    const u32 *w; u64 buf; u32 cnt, wi, ahead, pos;
    COVW_FN void init(const Src &s, u32 p) {
        w = s.w; pos = p;
        const u32 i = p >> 5, d = p & 31u;
        buf = ((u64)w[i] | ((u64)w[i + 1u] << 32)) >> d; cnt = 64u - d;
        wi = i + 2u; ahead = w[wi];
    }
    COVW_FN void refill() {               // afterwards cnt >= 33
        if (cnt <= 32u) { buf |= (u64)ahead << cnt; cnt += 32u; wi++; ahead = w[wi]; }
    }
    COVW_FN u32 low32() const { return (u32)buf; }
    COVW_FN void drop(u32 n) { buf >>= n; cnt -= n; pos += n; }
};
// (RingCursor — pass 3 reading its words through an 8-word window per lane in the wave's LDS, two 16-byte loads per refill, on the theory
// that waiting for the next word also waits for the stores issued in front of it (gfx9 counts loads and stores with one in-order counter) —
// was built, passed this file's CPU emulation and the device tests, and measured SLOWER: 21.3 ms per full round against 19.4 with plain
// loads, profiles/r05_inflate_sink_ring_kernel_times.log (128 registers instead of 100, eight LDS writes and a branch per refill).  Removed.)
// ---- canonical code of `n` symbols with code lengths lens[0 .. n): per-length limits / offsets and the symbols sorted by (length, value).
// Serial; false when the set is over-subscribed.
template <class WS>
COVW_FN bool build_lengths(WS &W, const u8 *lens, u32 n, u16 *limit, u16 *off, u16 *sorted16, u8 *sorted8) {
    u32 *cnt = W.cnt, *start = W.start;
    for (u32 l = 0; l < 16; l++) cnt[l] = 0;
    for (u32 s = 0; s < n; s++) cnt[lens[s] & 15u]++;
    cnt[0] = 0;
    u32 left = 1, run = 0, first = 0;
    limit[0] = 0; off[0] = 0; start[0] = 0;
    for (u32 l = 1; l < 16; l++) {
        left <<= 1;
        if (cnt[l] > left) return false;
        left -= cnt[l];
        start[l] = run;
        limit[l] = (u16)((first + cnt[l]) << (15u - l));      // <= 2^15 when not over-subscribed
        off[l] = (u16)((run - first) & 0xffffu);
        run += cnt[l]; first = (first + cnt[l]) << 1;
    }
    for (u32 s = 0; s < n; s++) {
        const u32 l = lens[s] & 15u;
        if (!l) continue;
        if (sorted16) sorted16[start[l]] = (u16)s; else sorted8[start[l]] = (u8)s;
        start[l]++;
    }
    return true;
}

// Per-length limits / offsets of a canonical code from its per-length counts; start[l] = index of length l's first symbol in sorted[].
// False when the set is over-subscribed.
COVW_FN bool code_offsets(const u32 *cnt, u32 *start, u16 *limit, u16 *off) {
    u32 left = 1, run = 0, first = 0;
    limit[0] = 0; off[0] = 0; start[0] = 0;
    for (u32 l = 1; l < 16; l++) {
        left <<= 1;
        if (cnt[l] > left) return false;
        left -= cnt[l];
        start[l] = run;
        limit[l] = (u16)((first + cnt[l]) << (15u - l));
        off[l] = (u16)((run - first) & 0xffffu);
        run += cnt[l]; first = (first + cnt[l]) << 1;
    }
    return true;
}

// Code length and sorted-symbol index of the code that begins the left-justified 15-bit value v, for a code known to be longer than
// FROM bits (FROM = 0: any code); 0 when v starts no code of the set.
template <u32 FROM>
COVW_FN u32 canonical(const u16 *limit, const u16 *off, u32 v, u32 &idx) {
    u32 l = FROM + 1u;
    for (u32 k = FROM + 1u; k < 15u; k++) l += v >= (u32)limit[k] ? 1u : 0u;      // limits do not decrease with the length
    if (v >= (u32)limit[15]) return 0;
    idx = ((u32)off[l] + (v >> (15u - l))) & 0xffffu;
    return l;
}

COVW_FN u32 lit_entry(u32 sym) {                   // table entry of a literal/length symbol, without the code length
    if (sym < 256u) return sym << 4;
    if (sym == 256u) return 0xf000u;
    if (sym > 285u) return 0xfff0u;
    const u32 li = sym - 257u;                     // RFC 1951 3.2.5: 257-264 are lengths 3-10, then groups of four share e extra bits, 285 = 258
    const u32 le = li < 8u ? 0u : (li == 28u ? 0u : (li - 4u) >> 2);
    const u32 lb = li < 8u ? 3u + li : (li == 28u ? 258u : 3u + ((4u + (li & 3u)) << le));
    return 0x8000u | (le << 12) | ((lb - 3u) << 4);
}
COVW_FN u32 dist_entry(u32 ds) {                   // ... of a distance symbol: distance codes come in pairs sharing e extra bits
    if (ds >= 30u) return DIST_INVALID;
    const u32 de = ds < 4u ? 0u : (ds - 2u) >> 1;
    const u32 base = ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << de);
    return (base << 8) | (de << 4);
}

// Lookup entries by INDEX: the code that starts the index's bits, if it is decidable from them.
COVW_FN void fill_lit_index(Tables &T, u32 i) {
    u32 idx = 0;
    const u32 l = canonical<0>(T.lit_limit, T.lit_off, covw_brev32(i) >> 17, idx);
    T.lit[i] = (u16)((l == 0 || l > LB) ? 0u : ((u32)T.lit_sorted[idx < 288u ? idx : 287u] | l));
}
COVW_FN void fill_dist_index(Tables &T, u32 i) {
    u32 idx = 0;
    const u32 l = canonical<0>(T.dist_limit, T.dist_off, covw_brev32(i) >> 17, idx);
    T.dist[i] = (l == 0 || l > DB) ? 0u : (dist_entry(T.dist_sorted[idx & 31u]) | l);
}

// ---- header of the DEFLATE block at `pos` (serial): fills W.hdr; fixed codes: W.T.lens; dynamic codes: the code-length code (W.cl and its
// canonical tables), hdr[4] = where the code lengths begin.
template <class WS>
COVW_FN void parse_header(WS &W, const Src &s, u32 pos) {
    u32 *h = W.hdr;
    h[6] = OK;
    if (pos + 3u > s.total_bits) { h[6] = ERR_FORMAT; return; }
    Cursor c; c.init(s, pos);
    h[1] = c.low32() & 1u; h[0] = (c.low32() >> 1) & 3u;
    c.drop(3);
    if (h[0] == 3u) { h[6] = ERR_FORMAT; return; }
    if (h[0] == 0u) {                                   // stored: LEN, NLEN at the next byte boundary
        c.drop((8u - (c.pos & 7u)) & 7u);
        if (c.pos + 32u > s.total_bits) { h[6] = ERR_FORMAT; return; }
        c.refill();
        const u32 w = c.low32();
        if (((w & 0xffffu) ^ (w >> 16)) != 0xffffu) { h[6] = ERR_FORMAT; return; }
        h[5] = w & 0xffffu; h[4] = c.pos + 32u;
        if ((u64)h[4] + 8ull * h[5] > s.total_bits) h[6] = ERR_FORMAT;
        return;
    }
    u8 *lens = W.T.lens;
    if (h[0] == 1u) {                                   // fixed codes
        for (u32 k = 0; k < 144; k++) lens[k] = 8;
        for (u32 k = 144; k < 256; k++) lens[k] = 9;
        for (u32 k = 256; k < 280; k++) lens[k] = 7;
        for (u32 k = 280; k < 288; k++) lens[k] = 8;
        for (u32 k = 0; k < 30; k++) lens[288 + k] = 5;
        lens[318] = 0; lens[319] = 0;
        h[2] = 288; h[3] = 30; h[4] = c.pos;
        return;
    }
    if (c.pos + 14u > s.total_bits) { h[6] = ERR_FORMAT; return; }
    c.refill();
    const u32 x = c.low32();
    const u32 hlit = (x & 31u) + 257u, hdist = ((x >> 5) & 31u) + 1u, hclen = ((x >> 10) & 15u) + 4u;
    c.drop(14);
    if (hlit > 286u || hdist > 30u) { h[6] = ERR_FORMAT; return; }
    u8 *cl = W.cl;
    for (u32 k = 0; k < 19; k++) cl[k] = 0;
    for (u32 k = 0; k < hclen; k++) {
        // the order in which the code-length code's own lengths are sent: 16 17 18 0 | 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
        const u32 sym = k < 3u ? 16u + k : (k == 3u ? 0u : ((k & 1u) ? 8u - ((k - 3u) >> 1) : 8u + ((k - 4u) >> 1)));
        c.refill();
        cl[sym] = (u8)(c.low32() & 7u); c.drop(3);
    }
    if (!build_lengths(W, cl, 19, W.climit, W.coff, nullptr, W.csorted)) { h[6] = ERR_FORMAT; return; }
    h[2] = hlit; h[3] = hdist; h[4] = c.pos;
}

// ---- the code lengths of a dynamic block (serial, through W.cltab): fills W.T.lens, hdr[4] = first unit bit
template <class WS>
COVW_FN void parse_code_lengths(WS &W, const Src &s) {
    u32 *h = W.hdr;
    u8 *lens = W.T.lens;
    const u32 hlit = h[2], hdist = h[3], want = hlit + hdist;
    Cursor c; c.init(s, h[4]);
    u32 n = 0;
    while (n < want) {
        if (c.pos >= s.total_bits) { h[6] = ERR_FORMAT; return; }
        c.refill();
        const u32 x = c.low32();
        const u32 e = W.cltab[x & 127u], l = e & 7u, sym = e >> 3;
        if (l == 0u) { h[6] = ERR_FORMAT; return; }
        if (sym < 16u) { lens[n++] = (u8)sym; c.drop(l); continue; }
        u32 rep, val = 0;
        if (sym == 16u) { if (n == 0) { h[6] = ERR_FORMAT; return; } val = lens[n - 1]; rep = 3u + bits_at(x, l, 2); c.drop(l + 2u); }
        else if (sym == 17u) { rep = 3u + bits_at(x, l, 3); c.drop(l + 3u); }
        else { rep = 11u + bits_at(x, l, 7); c.drop(l + 7u); }
        if (n + rep > want) { h[6] = ERR_FORMAT; return; }
        for (u32 k = 0; k < rep; k++) lens[n++] = (u8)val;
    }
    if (lens[256] == 0) { h[6] = ERR_FORMAT; return; }
    for (u32 k = hdist; k-- > 0;) lens[288 + k] = lens[hlit + k];      // backwards: the ranges overlap
    for (u32 k = hlit; k < 288; k++) lens[k] = 0;
    for (u32 k = 288 + hdist; k < 320; k++) lens[k] = 0;
    h[4] = c.pos;
}

COVW_FN void store8(u8 *d, u64 v) { COVW_TRACE_STORE(d, 8); __builtin_memcpy(d, &v, 8); }
COVW_FN void store4(u8 *d, u32 v) { COVW_TRACE_STORE(d, 4); __builtin_memcpy(d, &v, 4); }
COVW_FN void store_bytes(u8 *d, u64 v, u32 n) {        // exactly n <= 8 bytes of v
    if (n == 8u) { store8(d, v); return; }
    if (n & 4u) { store4(d, (u32)v); d += 4; v >>= 32; }
    if (n & 2u) { const u16 x = (u16)v; COVW_TRACE_STORE(d, 2); __builtin_memcpy(d, &x, 2); d += 2; v >>= 16; }
    if (n & 1u) { COVW_TRACE_STORE(d, 1); *d = (u8)v; }
}
// ---- pass 3's output of one lane.  A lane writes its bytes front to back, bytes [lo, own_end) of the block; it may run over bytes of its
// OWN range that come later (it overwrites them, or they are a match's and k_lz_resolve does), never past own_end, where the next lane's
// bytes begin.  literal(p, b): byte b belongs at p; match(p, len, t24, k): a match of len bytes begins at p, its token t24 goes into its
// first three bytes and p into token slot k; finish(p): p = own_end, everything pending leaves.
// The pending literals and the token leave together, as one 4- or 8-byte store when the padding falls into the match's own bytes; token
// positions leave four at a time.  (Seven store policies were measured — exact stores only, aligned words only, aligned 16- and 64-byte lines
// through per-lane line buffers in LDS, ...: profiles/r04_wave_variants.log.  A scattered store instruction of a wave costs ~70 ns of a CU's
// memory pipeline whatever its width (profiles/r04_store_probe.log), so the policy with the fewest store INSTRUCTIONS per lock-step wins.)
struct Sink {
    u8 *out; u16 *tok; u64 obuf, tbuf; u32 on, tn, last_k;
    COVW_FN void init(u8 *o, u16 *t, u32 /* own_end */) { out = o; tok = t; obuf = 0; tbuf = 0; on = 0; tn = 0; last_k = 0; }
    COVW_FN void literal(u32 p, u32 b) {
        obuf |= (u64)b << (8u * on);
        if (++on == 8u) { store8(out + p - 7u, obuf); obuf = 0; on = 0; }
    }
    // n <= 8 bytes of v (zero above them); up to `room` bytes behind them may be overwritten
    COVW_FN void store_padded(u8 *d, u64 v, u32 n, u32 room) {
        if (n > 4u) { if (n + room >= 8u) store8(d, v); else store_bytes(d, v, n); }
        else { if (n + room >= 4u) store4(d, (u32)v); else store_bytes(d, v, n); }
    }
    COVW_FN void match(u32 p, u32 len, u32 t24, u32 k) {
        const u32 room = len - 3u;
        u8 *d = out + p - on;
        const u64 v = obuf | ((u64)t24 << (8u * on));                 // on <= 7: at least one token byte fits
        if (on <= 5u) store_padded(d, v, on + 3u, room);
        else { store8(d, v); store_padded(d + 8, t24 >> (8u * (8u - on)), on - 5u, room); }
        obuf = 0; on = 0;
        tbuf |= (u64)p << (16u * tn);
        if (++tn == 4u) { store8(reinterpret_cast<u8 *>(tok + k - 3u), tbuf); tbuf = 0; tn = 0; }
        last_k = k;
    }
    COVW_FN void finish(u32 p) {
        if (on) store_bytes(out + p - on, obuf, on);
        if (tn) store_bytes(reinterpret_cast<u8 *>(tok + last_k + 1u - tn), tbuf, 2u * tn);
    }
};

COVW_FN void store16(u8 *d, u64 a, u64 b) { COVW_TRACE_STORE(d, 16); const u64 x[2] = {a, b}; __builtin_memcpy(d, x, 16); }
// ---- Sink16 (round 5, the default: 19.4 ms per full round against 20.7 with the 8-byte sink, ingest of 100 M reads 0.331 s against 0.344,
// alternating runs on one box, profiles/r05_inflate_sink_ring_*.log): a 16-byte window per lane instead of 8 pending bytes.  tools/ubench/store_probe (profiles/
// r05_store_probe.log) settled what a scattered store costs: ~1.1 ns of a CU's memory pipeline per LANE that stores, whatever the width up to
// 16 bytes and however many lanes take part in the instruction — so the only way to make pass 3's stores cheaper is to make them FEWER.
// A lane's bytes come in short segments (a couple of literals and a match's 3-byte token) separated by the rest of the match, which
// k_lz_resolve / k_lz_stage fill in later and which this lane may therefore overwrite with anything: all segments that begin inside
// the same 16 bytes leave as ONE store.  The window starts at the first byte put into it and is flushed when the next byte does not
// fit; a flush is one store of 4, 8 or 16 bytes (the padding lies in front of own_end: bytes this lane or the match resolution write
// later), exact stores only where the lane's range ends.
struct Sink16 {
    u8 *out; u16 *tok; u64 lo, hi, tbuf, tbuf2; u32 base, on, own_end, tn, last_k;
    COVW_FN void init(u8 *o, u16 *t, u32 end) { out = o; tok = t; lo = 0; hi = 0; tbuf = 0; tbuf2 = 0; base = 0; on = 0; own_end = end; tn = 0; last_k = 0; }
    COVW_FN void flush() {
        if (!on) return;
        u8 *d = out + base;
        if (on > 8u) { if (base + 16u <= own_end) store16(d, lo, hi); else { store8(d, lo); store_bytes(d + 8, hi, on - 8u); } }
        else if (on > 4u) { if (base + 8u <= own_end) store8(d, lo); else store_bytes(d, lo, on); }
        else { if (base + 4u <= own_end) store4(d, (u32)lo); else store_bytes(d, lo, on); }
        lo = 0; hi = 0; on = 0;
    }
    COVW_FN void put(u32 p, u32 v, u32 n) {               // the n = 1 or 3 low bytes of v belong at p (positions only ever grow)
        if (on && p - base + n > 16u) flush();
        if (!on) base = p;
        const u32 rel = p - base, sh = 8u * (rel & 7u);
        if (rel < 8u) { lo |= (u64)v << sh; if (rel + n > 8u) hi |= (u64)v >> (64u - sh); }      // (a token that straddles the halves: rel = 6 or 7)
        else hi |= (u64)v << sh;
        on = rel + n;
    }
    COVW_FN void literal(u32 p, u32 b) { put(p, b, 1u); }
    COVW_FN void match(u32 p, u32 len, u32 t24, u32 k) {
        (void)len;
        put(p, t24, 3u);
        if (tn < 4u) tbuf |= (u64)p << (16u * tn); else tbuf2 |= (u64)p << (16u * (tn - 4u));       // token positions leave eight at a time
        if (++tn == 8u) { store16(reinterpret_cast<u8 *>(tok + k - 7u), tbuf, tbuf2); tbuf = 0; tbuf2 = 0; tn = 0; }
        last_k = k;
    }
    COVW_FN void finish(u32) {
        flush();
        if (tn) {
            u8 *d = reinterpret_cast<u8 *>(tok + last_k + 1u - tn);
            if (tn > 4u) { store8(d, tbuf); store_bytes(d + 8, tbuf2, 2u * (tn - 4u)); } else store_bytes(d, tbuf, 2u * tn);
        }
    }
};

// Decodes one lane's units from `from` until the position reaches `until`.  MODE 0: positions only, and an invalid code is skipped over
// bit by bit (the start is a guess); 1: also counts output bytes and matches; 2: writes them through Sink (out + opos = where the
// lane's first byte goes, tok + tpos = its first token position; *err receives what went wrong).
// Returns the end position; *flags: bit 0 end of block met, bit 1 invalid code / ran off the payload.
struct NoSink {      // passes 1 and 2 write nothing
    COVW_FN void literal(u32, u32) {}
    COVW_FN void match(u32, u32, u32, u32) {}
    COVW_FN void finish(u32) {}
};
// (SinkT: where pass 3's bytes go — Sink: the block's place in global memory.  opos: position of the
// lane's first byte in the sink's coordinates, pmin: position of the block's first byte — a match may not reach in front of it.)
template <int MODE, class SinkT, int EXTRA = 1>
COVW_FN u32 run_share(const Tables &T, const Src &s, u32 from, u32 until, u32 *flags, u32 *nb, u32 *nt, SinkT &sink, u32 opos, u32 pmin, u32 tpos, u32 *err) {
    Cursor c; c.init(s, from);
    u32 f = 0, bytes = 0, toks = 0;
    // (Bounds in MODE 2: pass 2 counted this lane's bytes and matches with the same decoder and inflate_block checked the block's totals
    // against isize and TOK_CAP before pass 3, so only a match's distance is left to check here.)
    while (c.pos < until) {
        const u32 unit_at = c.pos;
        u32 trace = 0;
        (void)trace; (void)unit_at;
        c.refill();
        u32 x = c.low32();
        u32 e = T.lit[x & ((1u << LB) - 1u)];
        if ((e & 15u) == 0u) {
            trace |= 1u;
            u32 idx = 0;
            const u32 l = canonical<LB>(T.lit_limit, T.lit_off, covw_brev32(x) >> 17, idx);
            e = l ? ((u32)T.lit_sorted[idx < 288u ? idx : 287u] | l) : 0xfff0u;
        }
        const u32 n = e & 15u;
        if (!(e & 0x8000u)) {                       // a literal
            COVW_TRACE_UNIT(MODE, trace);
            c.drop(n);
            if (MODE == 2) sink.literal(opos + bytes, e >> 4);
            bytes++;
            // More literals in the same lock-step (two thirds of a BAM block's units are literals, in runs of 2.4 on average, and a wave's
            // step costs what its slowest lane's does): up to EXTRA more while the next unit begins inside the share and a table entry
            // says "literal" — such a code is at most LB bits long, and 33 refilled bits minus (1 + EXTRA) codes of <= LB leave >= LB for
            // the next lookup as long as (1 + EXTRA) * LB + LB <= 33, i.e. EXTRA <= 1 at LB = 10; beyond that the cursor is refilled first.
            // Anything else is left to the next step as it stands.
            for (int r = 0; r < EXTRA; r++) {
                if (c.pos >= until) break;
                if (r >= 1) c.refill();
                const u32 e2 = T.lit[c.low32() & ((1u << LB) - 1u)];
                if ((e2 & 15u) == 0u || (e2 & 0x8000u)) break;
                COVW_TRACE_UNIT(MODE, 0u);
                c.drop(e2 & 15u);
                if (MODE == 2) sink.literal(opos + bytes, e2 >> 4);
                bytes++;
            }
            continue;
        }
        if ((e & 0x7000u) == 0x7000u) {             // neither literal nor length
            if ((e & 0x0ff0u) == 0u) {              // end of block
                COVW_TRACE_UNIT(MODE, trace);
                c.drop(n);
                if (MODE == 0) continue;           // (a guessed start may see an end-of-block that is none; a true one ends what anybody uses of this lane)
                f |= 1u; break;
            }
            if (MODE == 0 && unit_at + 1u < until) { c.init(s, unit_at + 1u); continue; }
            f |= 2u; break;
        }
        const u32 le = (e >> 12) & 7u;
        const u32 len = 3u + ((e >> 4) & 0xffu) + bits_at(x, n, le);      // n + le <= 22 of the >= 33 bits
        c.drop(n + le);
        c.refill();
        x = c.low32();
        u32 ed = T.dist[x & ((1u << DB) - 1u)];
        trace |= 2u;
        if ((ed & 15u) == 0u) {
            trace |= 4u;
            u32 idx = 0;
            const u32 l = canonical<DB>(T.dist_limit, T.dist_off, covw_brev32(x) >> 17, idx);
            ed = l ? (dist_entry(T.dist_sorted[idx & 31u]) | l) : DIST_INVALID;
        }
        COVW_TRACE_UNIT(MODE, trace);
        const u32 dl = ed & 15u, de = (ed >> 4) & 15u;
        const u32 dist = ((ed >> 8) & 0x7fffu) + bits_at(x, dl, de);      // dl + de <= 28
        c.drop(dl + de);
        if ((ed & DIST_INVALID) != 0u) {            // no code of the set / symbols 30, 31
            if (MODE == 0 && unit_at + 1u < until) { c.init(s, unit_at + 1u); continue; }
            f |= 2u; break;
        }
        if (MODE == 2) {
            const u32 p = opos + bytes;
            if (dist > p - pmin) { *err = ERR_FORMAT; break; }
            sink.match(p, len, (dist - 1u) | ((len - 3u) << 15), tpos + toks);      // k_lz_resolve's token
        }
        bytes += len; toks++;
        // (A literal behind the match in the same lock-step — seven matches in ten are followed by one — was measured too: 20.0 ms per full
        // round against 17.8, profiles/r05_literal_after_match_*.log: it lengthens the match path, which nearly every lock-step takes.)
    }
    if (MODE != 0 && c.pos > s.total_bits) f |= 2u;      // only a lane's last unit can run off the payload: `until` lies inside it
    if (MODE == 2) sink.finish(opos + bytes);
    *flags = f; *nb = bytes; *nt = toks;
    return c.pos;
}

COVW_FN u32 share_begin_of(u32 B0, u32 S, u32 lane, u32 total_bits) {
    const u64 b = (u64)B0 + (u64)lane * S;
    return b < total_bits ? (u32)b : total_bits;
}

// One BGZF block.  comp_words: aligned words holding the raw DEFLATE payload from bit `bit0` on; out: the block's `isize` output
// bytes; tok: its token-position list.  *status = OK / ERR_*, *n_tok = matches written (0 unless OK).
// stop_after (measurements only, 0 in production): 1 = give up after the tables are built, 2 = after pass 1, 3 = after pass 2.
template <class SinkT = Sink16, int EXTRA = 1>
COVW_FN void inflate_block(Wave &W, const u32 *comp_words, u32 bit0, u32 payload_bits, u8 *out, u32 isize, u16 *tok, u32 *n_tok, u32 *status, u32 stop_after = 0) {
    Src s; s.w = comp_words; s.total_bits = bit0 + payload_bits;
    u32 pos = bit0, opos = 0, ntok = 0, err = OK, nblk = 0, chunk_bits = 0;
    bool last = false;
    while (!last && err == OK) {
        nblk++;
        COVW_PARFOR(lane) { if (lane == 0u) { parse_header(W, s, pos); W.n_deflate_blocks = nblk; if (nblk == 1u) W.n_chunks = 0; } }
        COVW_SYNC();
        if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
        last = W.hdr[1] != 0u;
        if (W.hdr[0] == 0u) {                       // stored block: the lanes copy its bytes
            const u32 len = W.hdr[5], src_bit = W.hdr[4];
            if (opos + len > isize) { err = ERR_FORMAT; break; }
            const u8 *src = reinterpret_cast<const u8 *>(s.w) + (src_bit >> 3);
            COVW_PARFOR(lane) {
                COVW_NO_UNROLL                 // (unrolled eight times this loop's address registers were the kernel's register peak)
                for (u32 k = lane; k < len; k += 64u) out[opos + k] = src[k];
            }
            opos += len; pos = src_bit + 8u * len;
            COVW_SYNC();
            continue;
        }
        if (W.hdr[0] == 2u) {                       // dynamic codes: the code-length code's lookup table, then the code lengths
            COVW_PARFOR(lane) {
                for (u32 i = lane; i < 128u; i += 64u) {
                    u32 idx = 0;
                    const u32 l = canonical<0>(W.climit, W.coff, covw_brev32(i) >> 17, idx);
                    W.cltab[i] = (u8)((l == 0u || l > 7u) ? 0u : (((u32)W.csorted[idx & 31u] << 3) | l));
                }
            }
            COVW_SYNC();
            COVW_PARFOR(lane) { if (lane == 0u) parse_code_lengths(W, s); }
            COVW_SYNC();
            if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
        }
        // ---- both codes from their lengths, with the whole wave: count and mark every symbol under its length, offsets per length (serial,
        // 2 x 15 steps), then every symbol finds its place among the symbols of its length by counting the marks below it
        COVW_PARFOR(lane) {
            for (u32 i = lane; i < 16u * 9u + 16u; i += 64u) W.T.mask[i] = 0;
            if (lane < 32u) W.cnt[lane] = 0;
        }
        COVW_SYNC();
        COVW_PARFOR(lane) {
            for (u32 sy = lane; sy < 320u; sy += 64u) {
                const u32 l = W.T.lens[sy] & 15u;
                if (!l) continue;
                if (sy < 288u) { COVW_ATOMIC_ADD(&W.cnt[l], 1u); COVW_ATOMIC_OR(&W.T.mask[l * 9u + (sy >> 5)], 1u << (sy & 31u)); }
                else { COVW_ATOMIC_ADD(&W.cnt[16u + l], 1u); COVW_ATOMIC_OR(&W.T.mask[144u + l], 1u << (sy - 288u)); }
            }
        }
        COVW_SYNC();
        COVW_PARFOR(lane) {
            if (lane == 0u) {
                W.cnt[0] = 0; W.cnt[16] = 0;
                const bool a = code_offsets(W.cnt, W.start, W.T.lit_limit, W.T.lit_off);
                const bool b = a && code_offsets(W.cnt + 16, W.start + 16, W.T.dist_limit, W.T.dist_off);
                if (!b) W.hdr[6] = ERR_FORMAT;
            }
        }
        COVW_SYNC();
        if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
        COVW_PARFOR(lane) {
            for (u32 sy = lane; sy < 320u; sy += 64u) {
                const u32 l = W.T.lens[sy] & 15u;
                if (!l) continue;
                if (sy < 288u) {
                    const u32 *m = W.T.mask + l * 9u, w = sy >> 5;
                    u32 rank = (u32)__builtin_popcount(m[w] & ((1u << (sy & 31u)) - 1u));
                    for (u32 j = 0; j < w; j++) rank += (u32)__builtin_popcount(m[j]);
                    W.T.lit_sorted[W.start[l] + rank] = (u16)lit_entry(sy);          // entries, not symbols: what the decoder wants
                } else {
                    const u32 d = sy - 288u;
                    W.T.dist_sorted[W.start[16u + l] + (u32)__builtin_popcount(W.T.mask[144u + l] & ((1u << d) - 1u))] = (u8)d;
                }
            }
        }
        COVW_SYNC();          // (the masks share their memory with the lookup table that is filled next)
        COVW_PARFOR(lane) {
            for (u32 i = lane; i < (1u << LB); i += 64u) fill_lit_index(W.T, i);
            for (u32 i = lane; i < (1u << DB); i += 64u) fill_dist_index(W.T, i);
        }
        COVW_SYNC();
        if (stop_after == 1u) { err = ERR_FORMAT; break; }
        // ---- the block's units, CHUNK by chunk: 64 shares of the bits [cur, span_end), three passes, and on behind the chunk's last unit
        // with the same tables until a lane meets end-of-block.  The first DEFLATE block of a payload takes the whole payload as its
        // one chunk (zlib and this repository's writer emit one block per BGZF block); if it ends early, the blocks behind it are
        // taken in chunks of about its length, so that the lanes are not spent on bits that belong to later blocks (other tables).
        const u32 B0 = W.hdr[4];
        u32 cur = B0;
        for (bool eob = false; !eob && err == OK;) {
            if (cur >= s.total_bits) { err = ERR_FORMAT; break; }            // no end-of-block inside the payload
            const u32 span_end = chunk_bits && s.total_bits - cur > chunk_bits ? cur + chunk_bits : s.total_bits;
            const u32 span = span_end - cur;
            const u32 S = span > 64u * MIN_SHARE_BITS ? (span + 63u) / 64u : MIN_SHARE_BITS;
            // ---- pass 1: where lane i's units begin = end[i - 1]
            COVW_PARFOR(lane) {
                u32 f, nb, nt;
                NoSink ns;
                const u32 g = share_begin_of(cur, S, lane, span_end);
                if (lane) W.end[lane - 1u] = g >= span_end ? span_end
                                                           : run_share<0, NoSink, EXTRA>(W.T, s, g - cur > OVERLAP_BITS ? g - OVERLAP_BITS : cur, g, &f, &nb, &nt, ns, 0, 0, 0, nullptr);
                if (lane == 63u) W.end[63] = span_end;
            }
            COVW_SYNC();
            if (stop_after == 2u) { err = ERR_FORMAT; break; }
            // ---- pass 2: from the left neighbour's end, until no end moves
            for (u32 round = 0;; round++) {
                COVW_PARFOR(lane) {
                    if (lane == 0u) { W.rounds = round + 1u; if (round == 0u) W.n_chunks++; }
                    const u32 from = lane ? W.end[lane - 1u] : cur;
                    const u32 ge = share_begin_of(cur, S, lane + 1u, span_end);
                    u32 f = 0, nb = 0, nt = 0, e = from;
                    NoSink ns;
                    if (from < ge) e = run_share<1, NoSink, EXTRA>(W.T, s, from, ge, &f, &nb, &nt, ns, 0, 0, 0, nullptr);
                    W.flags[lane] = f; W.nbytes[lane] = nb; W.ntok[lane] = nt; W.tmp[lane] = e;
                }
                COVW_SYNC();          // every lane has read its neighbour's old end
                // The lanes in front of F (F = the first lane that stopped at end-of-block or an invalid code) decide: when none of their
                // ends moved, they are a fixed point of the chain that starts at the exact `cur`, i.e. exact — and with them lane F's
                // start.  What the lanes behind F decode belongs to the next DEFLATE block: it never settles and nobody uses it.
                COVW_PARFOR(lane) {
                    if (lane == 0u) {
                        u32 ch = 0;
                        for (u32 i = 0; i < 63u && !W.flags[i]; i++) ch |= W.tmp[i] != W.end[i] ? 1u : 0u;
                        W.changed = ch;
                    }
                }
                COVW_SYNC();
                COVW_PARFOR(lane) { W.end[lane] = W.tmp[lane]; }
                COVW_SYNC();
                if (!W.changed) break;
                if (round >= 65u) { err = ERR_FORMAT; break; }
            }
            if (err != OK) break;
            if (stop_after == 3u) { err = ERR_FORMAT; break; }
            // ---- the lanes up to the first one that met end-of-block are what is left of the block (all 64 when none did); prefix sums
            COVW_PARFOR(lane) {
                if (lane == 0u) {
                    u32 nv = 64, eob_at = NO_EOB, ob = 0, tb = 0, bad = 0;
                    for (u32 i = 0; i < 64u; i++) {
                        W.obase[i] = ob; W.tbase[i] = tb;
                        ob += W.nbytes[i]; tb += W.ntok[i];
                        if (W.flags[i]) { nv = i + 1u; if (W.flags[i] == 1u) eob_at = W.end[i]; else bad = 1; break; }
                    }
                    W.n_valid = nv; W.eob_at = eob_at;
                    W.hdr[7] = ob; W.hdr[5] = tb;
                    if (bad) W.hdr[6] = ERR_FORMAT;                          // an invalid code, or a unit that runs off the payload
                }
            }
            COVW_SYNC();
            if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
            if (opos + W.hdr[7] > isize) { err = ERR_SIZE; break; }
            if (ntok + W.hdr[5] > TOK_CAP) { err = ERR_FORMAT; break; }
            // ---- pass 3: write
            COVW_PARFOR(lane) {
                if (lane < W.n_valid) {
                    const u32 from = lane ? W.end[lane - 1u] : cur;
                    const u32 ge = share_begin_of(cur, S, lane + 1u, span_end);
                    u32 f, nb, nt, e2 = OK;
                    SinkT sink; sink.init(out, tok, opos + W.obase[lane] + W.nbytes[lane]);
                    if (from < ge) (void)run_share<2, SinkT, EXTRA>(W.T, s, from, ge, &f, &nb, &nt, sink, opos + W.obase[lane], 0, ntok + W.tbase[lane], &e2);
                    if (e2 != OK) W.hdr[6] = e2;
                }
            }
            COVW_SYNC();
            if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
            opos += W.hdr[7]; ntok += W.hdr[5];
            if (W.eob_at != NO_EOB) { eob = true; pos = W.eob_at; }
            else cur = W.end[63];                                            // exact: the chunk's last unit ends here
        }
        if (err != OK) break;
        // the blocks behind an early end are taken in chunks of about this block's length (+ 1/8), at least the smallest full set of shares
        { const u32 body = pos - B0; chunk_bits = body + (body >> 3); if (chunk_bits < 64u * MIN_SHARE_BITS) chunk_bits = 64u * MIN_SHARE_BITS; }
    }
    if (err == OK && opos != isize) err = ERR_SIZE;
    *n_tok = err == OK ? ntok : 0u;
    *status = err;
}

}  // namespace covw
