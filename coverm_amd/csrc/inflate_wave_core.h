// k_inflate_wave — ONE WAVE PER BGZF BLOCK: the algorithm, written once, run two ways.
//
// Lane-per-block inflate (k_inflate) keeps a private Huffman table per lane in LDS, which bounds the resident lanes, and decodes a
// block serially, which sets the latency of a round.  Here the 64 lanes of a wave share ONE set of tables and each decodes 1/64 of
// the block's bit stream:
//   pass 1  lane i starts at the GUESSED offset B0 + i * S as if a unit (literal | length + extra + distance + extra | end of block)
//           began there and decodes to the end of its share: Huffman streams self-synchronise — on BAM blocks a decoder started at an
//           arbitrary bit is in step with the true sequence after a median of 6 units, 70 at most (profiles/r03_deflate_sync_probe.log)
//           — so the position where it leaves its share is, almost always, a true unit boundary;
//   pass 2  lane i starts where lane i - 1 ended (lane 0 at the true first unit) and decodes exactly its units, counting output bytes
//           and matches; repeated while any end moves (normally once; lane k is exact after round k whatever pass 1 guessed).  The
//           lanes up to the first end-of-block symbol are the Huffman block; the rest of the wave decoded what belongs to the next
//           DEFLATE block with the wrong tables and is discarded;
//   pass 3  output offsets = prefix sums of the counts; every lane decodes its units once more and writes: literals at their final
//           positions, a match as a 3-byte token in place + its 16-bit position in the block's token list (k_lz_resolve's contract,
//           unchanged).  No store touches a byte that another lane owns: literals leave as exact 8/4/2/1-byte stores, a token as
//           2 + 1 bytes (4 when the match is longer than three bytes: the fourth byte is the match's own).
// Header parsing and the per-length tables are serial (lane 0); the primary tables are filled in parallel, one INDEX per lane step
// (an index is decoded canonically like a long code), so no lane writes more entries than another.
//
// This file contains no HIP: it is a sequence of `COVW_PARFOR(lane) { ... }` regions over wave-shared state, separated by wave
// barriers.  csrc/ingest_kernels.hip.h instantiates it with lane = the thread and LDS-resident state; tests/c/inflate_wave_host.cpp
// instantiates it with a loop over 64 lanes and checks the bytes against zlib — every line of the algorithm runs on the CPU in the
// test, only the barrier and the bit-reversal intrinsic differ.  The includer defines:
//   COVW_FN                    function qualifier
//   COVW_PARFOR(lane)          `for`-like header: runs the following block for lane = 0 .. 63 (device: once, lane = the thread's lane)
//   COVW_SYNC()                wave barrier + LDS fence (host: nothing)
//   covw_brev32(x)             bit reversal of a 32-bit word
// Statements outside COVW_PARFOR regions are executed by all lanes with identical (wave-uniform) values.
#pragma once
#include <stdint.h>

namespace covw {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16;
typedef unsigned char u8;

constexpr u32 LB = 11, DB = 9;            // primary table bits of the literal/length and the distance alphabet
constexpr u32 TOK_CAP = 21888;            // = covi::INF_TOK_CAP
enum { OK = 0, ERR_FORMAT = 1, ERR_CRC = 2, ERR_SIZE = 3 };       // = covi::INF_*
constexpr u32 NO_EOB = 0xffffffffu;
constexpr u32 MIN_SHARE_BITS = 512;

struct Tables {
    u16 lit[1u << LB];                    // bits 0-3 code length (0: not decidable from LB bits); literal / end of block: bit 15 = 0, bits 4-12 symbol;
                                          // length symbol: bit 15 = 1, bits 4-11 base - 3, bits 12-14 extra bits; 0xfff0 | length: symbols 286 / 287
    u16 dist[1u << DB];                   // bits 0-3 code length, bits 4-8 symbol
    u16 lit_limit[16], dist_limit[16];    // limit[l] = (first code of length l + count[l]) << (15 - l)
    u16 lit_off[16], dist_off[16];        // (index of the first symbol of length l in sorted[]) - (first code of length l)  (mod 2^16)
    u16 lit_sorted[288];
    u8 dist_sorted[32];
    u8 lens[320];                         // code lengths: [0, 288) literal/length, [288, 320) distance
};

struct Wave {                             // wave-shared state (LDS on the device): 8.2 KiB
    Tables T;
    u32 end[64];                          // bit position where lane i's units end (the first unit boundary at or behind its share's end, or where it stopped)
    u32 tmp[64];
    u32 flags[64];                        // bit 0: the lane met end-of-block, bit 1: it met an invalid code / ran off the payload
    u32 nbytes[64], ntok[64];             // output bytes / matches of the lane's units
    u32 obase[64], tbase[64];             // exclusive prefix sums of the two
    u32 hdr[8];                           // [0] type, [1] last, [2] hlit, [3] hdist, [4] first unit bit (stored: first data bit), [5] stored length / tokens,
                                          // [6] error, [7] bytes
    u32 changed, n_valid, eob_at, rounds; // rounds: pass-2 rounds of the last Huffman block (statistics)
    u32 cnt[16], start[16];               // build_lengths' scratch (indexed dynamically: in LDS, not in private memory)
    u16 climit[16], coff[16];             // the code-length code
    u8 csorted[32], cl[32];
};

struct Src { const u32 *w; u32 total_bits; };     // aligned words of the payload (readable 16 bytes past its end); bit 0 = bit 0 of w[0]

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
    COVW_FN u32 peek(u32 n) const { return (u32)buf & ((1u << n) - 1u); }
    COVW_FN u32 low32() const { return (u32)buf; }
    COVW_FN void drop(u32 n) { buf >>= n; cnt -= n; pos += n; }
};

// ---- canonical code of `n` symbols with code lengths lens[0 .. n): per-length limits / offsets and the symbols sorted by (length, value).
// Serial; false when the set is over-subscribed.
COVW_FN bool build_lengths(Wave &W, const u8 *lens, u32 n, u16 *limit, u16 *off, u16 *sorted16, u8 *sorted8) {
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

// Code length and sorted-symbol index of the code that begins the left-justified 15-bit value v; 0 when v starts no code of the set.
COVW_FN u32 canonical(const u16 *limit, const u16 *off, u32 v, u32 &idx) {
    u32 l = 1;
    for (u32 k = 1; k < 15; k++) l += v >= (u32)limit[k] ? 1u : 0u;      // limits do not decrease with the length
    if (v >= (u32)limit[15]) return 0;
    idx = ((u32)off[l] + (v >> (15u - l))) & 0xffffu;
    return l;
}

COVW_FN u32 lit_entry(u32 sym, u32 len) {          // table entry of a literal/length symbol
    if (sym <= 256u) return (sym << 4) | len;
    if (sym > 285u) return 0xfff0u | len;
    const u32 li = sym - 257u;                     // RFC 1951 3.2.5: 257-264 are lengths 3-10, then groups of four share e extra bits, 285 = 258
    const u32 le = li < 8u ? 0u : (li == 28u ? 0u : (li - 4u) >> 2);
    const u32 lb = li < 8u ? 3u + li : (li == 28u ? 258u : 3u + ((4u + (li & 3u)) << le));
    return 0x8000u | (le << 12) | ((lb - 3u) << 4) | len;
}

// Primary entries by INDEX: the code that starts the index's bits, if it is decidable from them.
COVW_FN void fill_lit_index(Tables &T, u32 i) {
    u32 idx = 0;
    const u32 l = canonical(T.lit_limit, T.lit_off, covw_brev32(i) >> 17, idx);
    T.lit[i] = (u16)((l == 0 || l > LB) ? 0u : lit_entry(T.lit_sorted[idx < 288u ? idx : 287u], l));
}
COVW_FN void fill_dist_index(Tables &T, u32 i) {
    u32 idx = 0;
    const u32 l = canonical(T.dist_limit, T.dist_off, covw_brev32(i) >> 17, idx);
    T.dist[i] = (u16)((l == 0 || l > DB) ? 0u : (((u32)T.dist_sorted[idx & 31u] << 4) | l));
}

// ---- header of the DEFLATE block at `pos` (serial): fills W.hdr and, for Huffman blocks, W.T.lens.
COVW_FN void parse_header(Wave &W, const Src &s, u32 pos) {
    u32 *h = W.hdr;
    h[6] = OK;
    if (pos + 3u > s.total_bits) { h[6] = ERR_FORMAT; return; }
    Cursor c; c.init(s, pos);
    h[1] = c.peek(1); c.drop(1);
    h[0] = c.peek(2); c.drop(2);
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
    const u32 hlit = c.peek(5) + 257u; c.drop(5);
    const u32 hdist = c.peek(5) + 1u; c.drop(5);
    const u32 hclen = c.peek(4) + 4u; c.drop(4);
    if (hlit > 286u || hdist > 30u) { h[6] = ERR_FORMAT; return; }
    u8 *cl = W.cl;
    for (u32 k = 0; k < 19; k++) cl[k] = 0;
    for (u32 k = 0; k < hclen; k++) {
        // the order in which the code-length code's own lengths are sent: 16 17 18 0 | 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
        const u32 sym = k < 3u ? 16u + k : (k == 3u ? 0u : ((k & 1u) ? 8u - ((k - 3u) >> 1) : 8u + ((k - 4u) >> 1)));
        c.refill();
        cl[sym] = (u8)c.peek(3); c.drop(3);
    }
    if (!build_lengths(W, cl, 19, W.climit, W.coff, nullptr, W.csorted)) { h[6] = ERR_FORMAT; return; }
    const u32 want = hlit + hdist;
    u32 n = 0;
    while (n < want) {
        if (c.pos >= s.total_bits) { h[6] = ERR_FORMAT; return; }
        c.refill();
        u32 idx = 0;
        const u32 l = canonical(W.climit, W.coff, covw_brev32(c.low32()) >> 17, idx);
        if (l == 0 || l > 7u) { h[6] = ERR_FORMAT; return; }
        const u32 sym = W.csorted[idx & 31u];
        c.drop(l);
        if (sym < 16u) { lens[n++] = (u8)sym; continue; }
        u32 rep, val = 0;
        if (sym == 16u) { if (n == 0) { h[6] = ERR_FORMAT; return; } val = lens[n - 1]; rep = 3u + c.peek(2); c.drop(2); }
        else if (sym == 17u) { rep = 3u + c.peek(3); c.drop(3); }
        else { rep = 11u + c.peek(7); c.drop(7); }
        if (n + rep > want) { h[6] = ERR_FORMAT; return; }
        for (u32 k = 0; k < rep; k++) lens[n++] = (u8)val;
    }
    if (lens[256] == 0) { h[6] = ERR_FORMAT; return; }
    for (u32 k = hdist; k-- > 0;) lens[288 + k] = lens[hlit + k];      // backwards: the ranges overlap
    for (u32 k = hlit; k < 288; k++) lens[k] = 0;
    for (u32 k = 288 + hdist; k < 320; k++) lens[k] = 0;
    h[2] = hlit; h[3] = hdist; h[4] = c.pos;
}

COVW_FN void store_bytes(u8 *d, u64 v, u32 n) {        // exactly n <= 8 bytes of v
    if (n == 8u) { __builtin_memcpy(d, &v, 8); return; }
    if (n & 4u) { const u32 x = (u32)v; __builtin_memcpy(d, &x, 4); d += 4; v >>= 32; }
    if (n & 2u) { const u16 x = (u16)v; __builtin_memcpy(d, &x, 2); d += 2; v >>= 16; }
    if (n & 1u) *d = (u8)v;
}

// Decodes one lane's units from `from` up to the end of its share.  MODE 0: positions only; 1: also counts output bytes and matches;
// 2: writes them (out + opos = where the lane's first byte goes, tok + tpos = its first token position; *err receives what went wrong).
// Returns the end position; *flags: bit 0 end of block met, bit 1 invalid code / ran off the payload.
template <int MODE>
COVW_FN u32 run_share(const Tables &T, const Src &s, u32 from, u32 share_end, u32 *flags, u32 *nb, u32 *nt, u8 *out, u32 opos, u32 isize, u16 *tok, u32 tpos,
                      u32 *err) {
    Cursor c; c.init(s, from);
    u32 f = 0, bytes = 0, toks = 0, on = 0;
    u64 obuf = 0;
    while (c.pos < share_end) {
        c.refill();
        u32 e = T.lit[c.peek(LB)];
        if ((e & 15u) == 0u) {
            u32 idx = 0;
            const u32 l = canonical(T.lit_limit, T.lit_off, covw_brev32(c.low32()) >> 17, idx);
            if (l == 0) { f |= 2u; break; }
            e = lit_entry(T.lit_sorted[idx < 288u ? idx : 287u], l);
        }
        c.drop(e & 15u);
        if (!(e & 0x8000u)) {
            const u32 sym = (e >> 4) & 0x1ffu;
            if (sym == 256u) { f |= 1u; break; }
            if (c.pos > s.total_bits) { f |= 2u; break; }
            if (MODE == 2) {
                if (opos + bytes >= isize) { *err = ERR_SIZE; break; }
                obuf |= (u64)sym << (8u * on);
                if (++on == 8u) { store_bytes(out + opos + bytes - 7u, obuf, 8); obuf = 0; on = 0; }
            }
            bytes++;
            continue;
        }
        if ((e & 0xfff0u) == 0xfff0u) { f |= 2u; break; }
        const u32 le = (e >> 12) & 7u;
        const u32 len = 3u + ((e >> 4) & 0xffu) + c.peek(le);
        c.drop(le);
        c.refill();
        u32 ed = T.dist[c.peek(DB)];
        if ((ed & 15u) == 0u) {
            u32 idx = 0;
            const u32 l = canonical(T.dist_limit, T.dist_off, covw_brev32(c.low32()) >> 17, idx);
            if (l == 0) { f |= 2u; break; }
            ed = ((u32)T.dist_sorted[idx & 31u] << 4) | l;
        }
        const u32 ds = ed >> 4;
        if (ds >= 30u) { f |= 2u; break; }
        c.drop(ed & 15u);
        const u32 de = ds < 4u ? 0u : (ds - 2u) >> 1;                  // distance codes come in pairs sharing e extra bits
        const u32 dist = (ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << de)) + c.peek(de);    // (>= 33 bits after the refill: code + extra <= 28)
        c.drop(de);
        if (c.pos > s.total_bits) { f |= 2u; break; }
        if (MODE == 2) {
            const u32 p = opos + bytes;
            if (on) { store_bytes(out + p - on, obuf, on); obuf = 0; on = 0; }
            if (dist > p || p + len > isize || tpos + toks >= TOK_CAP) { *err = ERR_FORMAT; break; }
            const u32 t24 = (dist - 1u) | ((len - 3u) << 15);          // k_lz_resolve's token: in the first three bytes of the match's own destination
            store_bytes(out + p, t24, len > 3u ? 4u : 3u);
            tok[tpos + toks] = (u16)p;
        }
        bytes += len; toks++;
    }
    if (MODE == 2 && on) store_bytes(out + opos + bytes - on, obuf, on);
    *flags = f; *nb = bytes; *nt = toks;
    return c.pos;
}

COVW_FN u32 share_end_of(u32 B0, u32 S, u32 lane, u32 total_bits) {
    const u64 e = (u64)B0 + (u64)(lane + 1u) * S;
    return e < total_bits ? (u32)e : total_bits;
}

// One BGZF block.  comp_words: aligned words holding the raw DEFLATE payload from bit `bit0` on; out: the block's `isize` output
// bytes; tok: its token-position list.  *status = OK / ERR_*, *n_tok = matches written (0 unless OK).
// stop_after (measurements only, 0 in production): 1 = give up after the tables are built, 2 = after pass 1, 3 = after pass 2.
COVW_FN void inflate_block(Wave &W, const u32 *comp_words, u32 bit0, u32 payload_bits, u8 *out, u32 isize, u16 *tok, u32 *n_tok, u32 *status, u32 stop_after = 0) {
    Src s; s.w = comp_words; s.total_bits = bit0 + payload_bits;
    u32 pos = bit0, opos = 0, ntok = 0, err = OK;
    bool last = false;
    while (!last && err == OK) {
        COVW_PARFOR(lane) { if (lane == 0u) parse_header(W, s, pos); }
        COVW_SYNC();
        if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
        last = W.hdr[1] != 0u;
        if (W.hdr[0] == 0u) {                       // stored block: the lanes copy its bytes
            const u32 len = W.hdr[5], src_bit = W.hdr[4];
            if (opos + len > isize) { err = ERR_FORMAT; break; }
            const u8 *src = reinterpret_cast<const u8 *>(s.w) + (src_bit >> 3);
            COVW_PARFOR(lane) { for (u32 k = lane; k < len; k += 64u) out[opos + k] = src[k]; }
            opos += len; pos = src_bit + 8u * len;
            COVW_SYNC();
            continue;
        }
        COVW_PARFOR(lane) {
            if (lane == 0u) {
                Tables &T = W.T;
                const bool a = build_lengths(W, T.lens, W.hdr[2], T.lit_limit, T.lit_off, T.lit_sorted, nullptr);
                const bool b = a && build_lengths(W, T.lens + 288, W.hdr[3], T.dist_limit, T.dist_off, nullptr, T.dist_sorted);
                if (!b) W.hdr[6] = ERR_FORMAT;
            }
        }
        COVW_SYNC();
        if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
        COVW_PARFOR(lane) {
            for (u32 i = lane; i < (1u << LB); i += 64u) fill_lit_index(W.T, i);
            for (u32 i = lane; i < (1u << DB); i += 64u) fill_dist_index(W.T, i);
        }
        COVW_SYNC();
        if (stop_after == 1u) { err = ERR_FORMAT; break; }
        const u32 B0 = W.hdr[4];
        const u32 span = s.total_bits > B0 ? s.total_bits - B0 : 0u;
        // a share shorter than the distance over which a decoder falls in step (p99: 50 units, ~480 bits) makes pass 1 guess wrong and
        // pass 2 repeat: small blocks use fewer lanes
        const u32 S = span > 64u * MIN_SHARE_BITS ? (span + 63u) / 64u : MIN_SHARE_BITS;
        // ---- pass 1: from the guessed offsets
        COVW_PARFOR(lane) {
            u32 f, nb, nt;
            const u64 g = (u64)B0 + (u64)lane * S;
            W.end[lane] = g >= s.total_bits ? s.total_bits
                                            : run_share<0>(W.T, s, (u32)g, share_end_of(B0, S, lane, s.total_bits), &f, &nb, &nt, nullptr, 0, 0, nullptr, 0, nullptr);
        }
        COVW_SYNC();
        if (stop_after == 2u) { err = ERR_FORMAT; break; }
        // ---- pass 2: from the left neighbour's end, until no end moves (lane k is exact after round k)
        for (u32 round = 0;; round++) {
            COVW_PARFOR(lane) {
                if (lane == 0u) W.rounds = round + 1u;
                const u32 from = lane ? W.end[lane - 1u] : B0;
                const u32 ge = share_end_of(B0, S, lane, s.total_bits);
                u32 f = 0, nb = 0, nt = 0, e = from;
                if (from < ge) e = run_share<1>(W.T, s, from, ge, &f, &nb, &nt, nullptr, 0, 0, nullptr, 0, nullptr);
                W.flags[lane] = f; W.nbytes[lane] = nb; W.ntok[lane] = nt; W.tmp[lane] = e;
            }
            COVW_SYNC();          // every lane has read its neighbour's old end
            // Lanes 0 .. F (F = the first lane that stopped at end-of-block or an invalid code) decide: when none of their ends moved, they are
            // a fixed point of the chain that starts at the exact B0, i.e. exact.  What the lanes behind F decode belongs to the next DEFLATE
            // block (other tables): it never settles and nobody uses it.
            COVW_PARFOR(lane) {
                if (lane == 0u) {
                    u32 ch = 0;
                    for (u32 i = 0; i < 64u; i++) { ch |= W.tmp[i] != W.end[i] ? 1u : 0u; if (W.flags[i]) break; }
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
        // ---- the Huffman block = the lanes up to the first one that met end-of-block (or an invalid code); prefix sums
        COVW_PARFOR(lane) {
            if (lane == 0u) {
                u32 nv = 64, eob_at = NO_EOB, ob = 0, tb = 0;
                for (u32 i = 0; i < 64u; i++) {
                    W.obase[i] = ob; W.tbase[i] = tb;
                    ob += W.nbytes[i]; tb += W.ntok[i];
                    if (W.flags[i]) { nv = i + 1u; if (W.flags[i] == 1u) eob_at = W.end[i]; break; }
                }
                W.n_valid = nv; W.eob_at = eob_at;
                W.hdr[7] = ob; W.hdr[5] = tb;
            }
        }
        COVW_SYNC();
        if (W.eob_at == NO_EOB) { err = ERR_FORMAT; break; }               // no end-of-block inside the payload, or an invalid code in front of it
        if (opos + W.hdr[7] > isize) { err = ERR_SIZE; break; }
        if (ntok + W.hdr[5] > TOK_CAP) { err = ERR_FORMAT; break; }
        // ---- pass 3: write
        COVW_PARFOR(lane) {
            if (lane < W.n_valid) {
                const u32 from = lane ? W.end[lane - 1u] : B0;
                const u32 ge = share_end_of(B0, S, lane, s.total_bits);
                u32 f, nb, nt, e2 = OK;
                if (from < ge) (void)run_share<2>(W.T, s, from, ge, &f, &nb, &nt, out, opos + W.obase[lane], isize, tok, ntok + W.tbase[lane], &e2);
                if (e2 != OK) W.hdr[6] = e2;
            }
        }
        COVW_SYNC();
        if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
        opos += W.hdr[7]; ntok += W.hdr[5]; pos = W.eob_at;
    }
    if (err == OK && opos != isize) err = ERR_SIZE;
    *n_tok = err == OK ? ntok : 0u;
    *status = err;
}

}  // namespace covw
