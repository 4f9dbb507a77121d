// k_inflate_lds — ONE WORKGROUP (256 lanes) PER BGZF BLOCK, THE BLOCK ASSEMBLED IN LDS: the algorithm, written once, run two ways.
//
// k_inflate_wave (csrc/inflate_wave_core.h) decodes a block with the 64 lanes of one wave and leaves two things to the memory system that it
// is bad at: every lane stores into its own KiB of the output (a scattered store instruction costs a CU's memory pipeline ~70 ns whatever
// its width, profiles/r04_store_probe.log — 9 of the kernel's 20 ms per round), and the matches are resolved by a second kernel through
// global memory (k_lz_resolve: 19.5 ms per round, 2.5-5 x read amplification).  A BGZF block inflates to at most 64 KiB and a gfx950 CU
// has 160 KiB of LDS, so here the block never leaves the CU before it is complete:
//   stage    the compressed payload is copied into LDS with coalesced 16-byte loads (it lies where the output will be assembled: passes
//            1 and 2 produce no output), so the header parse and two of the three decode passes read their bit stream from LDS instead
//            of through 256 scattered global loads per step;
//   tables   header and code lengths by lane 0, both Huffman codes built by all lanes (as in inflate_wave_core.h);
//   pass 1/2 as in inflate_wave_core.h with 256 shares instead of 64: where each share's first unit begins (speculative, then verified),
//            how many bytes each share produces; prefix sums give every lane its place in the output;
//   pass 3   every lane decodes its units once more (bit stream from global memory: the LDS image is being overwritten) and writes
//            literals into the block's image in LDS; a match leaves its 3-byte token in its first bytes and marks its bytes PENDING in a
//            bitmap (one bit per output byte);
//   resolve  every lane walks its own matches in order (the pending bits of its output range ARE the list) and copies a match as soon as no
//            byte of its source is pending; copying clears the bits.  No rounds and no barrier: a lane whose source is not final yet tries
//            again; the lowest unresolved match of the block is always ready, so the chain cannot stall;
//   write    the finished image leaves as aligned 16-byte stores, every byte of the block written to global memory exactly once.
// Per block: compressed bytes read twice (stage, pass 3), output written once; LDS: 64 KiB image + 8 KiB bitmap + 4 KiB tables + 3 KiB of
// per-lane state = 79.2 KiB, two workgroups per CU.
//
// Like inflate_wave_core.h this file contains no HIP: `COVL_PARFOR(lane) { ... }` regions over workgroup-shared state, separated by
// barriers.  tests/c/inflate_lds_host.cpp instantiates it with a loop over 256 lanes and checks the bytes against zlib.  The includer
// defines, beside what inflate_wave_core.h wants:
//   COVL_PARFOR(lane)          runs the following block for lane = 0 .. 255 (device: once, lane = threadIdx.x)
//   COVL_SYNC()                workgroup barrier + LDS fence (host: nothing)
//   COVL_LD(p)                 load of a workgroup-shared word that another lane may be changing (device: relaxed atomic load)
//   COVL_RELEASE() / COVL_ACQUIRE()   order this lane's LDS writes before / reads after (host: nothing)
//   COVL_RELAX()               a lane that found its source pending is about to look again (device: s_sleep)
//   COVL_SPIN_LIMIT            how often a lane looks again before it gives the round up (device: never gives up; host: 1 — the lanes run
//                              one after the other there, and the resolve step is repeated until every lane is done)
//   COVW_ATOMIC_MIN / COVW_ATOMIC_AND (optional on the host)
#pragma once
#include "inflate_wave_core.h"

#ifndef COVW_ATOMIC_MIN
#define COVW_ATOMIC_MIN(p, v) (*(p) = *(p) < (v) ? *(p) : (v))
#define COVW_ATOMIC_AND(p, v) (*(p) &= (v))
#endif

namespace covl {

using covw::u8; using covw::u16; using covw::u32; using covw::u64;
using covw::Tables; using covw::Src; using covw::Cursor; using covw::OK; using covw::ERR_FORMAT; using covw::ERR_SIZE; using covw::NO_EOB;

constexpr u32 NL = 256;                           // lanes per block
constexpr u32 NONE = 0xffffffffu;
constexpr u32 IMG_BYTES = 65536 + 64;              // the block's image: 15 bytes of alignment in front of 64 KiB; the compressed payload + 16 + 64 bytes of slack
constexpr u32 PEND_WORDS = IMG_BYTES / 32 + 2;
constexpr u32 CH = 8;                              // bytes a lane copies per step of the resolve loop
constexpr u32 MIN_SHARE_BITS = 256, OVERLAP_BITS = covw::OVERLAP_BITS;

struct Block {                                     // workgroup-shared state (LDS on the device)
    union { u8 img[IMG_BYTES]; u32 cin[IMG_BYTES / 4]; };      // passes 1 and 2: the compressed payload; from pass 3 on: the output
    u32 pend[PEND_WORDS];                          // bit p: byte p of the image belongs to a match that is not copied yet
    Tables T;
    u32 end[NL];                                   // bit position where lane i's units end = where lane i + 1's begin
    union {
        struct { u32 tmp[NL], cnt2[NL]; };         // pass 2: new end | bytes of the lane's units; pass 3 / resolve: first match | end of the lane's output
        struct {                                   // while a block's tables are built (covw::parse_header ...)
            u32 cnt[32], start[32];
            u16 climit[16], coff[16];
            u8 csorted[32], cl[32];
            u8 cltab[128];
        };
    };
    u8 flags[NL];                                  // bit 0: the lane met end-of-block, bit 1: it met an invalid code / ran off the payload
    u32 hdr[8];                                    // as covw::Wave::hdr
    u32 changed, first_flag, n_valid, eob_at, rounds, n_deflate_blocks, n_chunks, remaining, err3;
    u32 gsum[4];
};

COVW_FN u32 lo_mask(u32 a) { return ~0u << (a & 31u); }                 // bits (a mod 32) .. 31
COVW_FN u32 hi_mask(u32 b) { return ~0u >> (31u - ((b - 1u) & 31u)); }  // bits 0 .. ((b - 1) mod 32)

// bits [a, b) of the bitmap, b > a
COVW_FN void set_range(u32 *pend, u32 a, u32 b) {
    const u32 w0 = a >> 5, w1 = (b - 1u) >> 5;
    if (w0 == w1) { COVW_ATOMIC_OR(&pend[w0], lo_mask(a) & hi_mask(b)); return; }
    COVW_ATOMIC_OR(&pend[w0], lo_mask(a));
    for (u32 w = w0 + 1u; w < w1; w++) COVW_ATOMIC_OR(&pend[w], ~0u);
    COVW_ATOMIC_OR(&pend[w1], hi_mask(b));
}
COVW_FN void clear_range(u32 *pend, u32 a, u32 b) {
    const u32 w0 = a >> 5, w1 = (b - 1u) >> 5;
    if (w0 == w1) { COVW_ATOMIC_AND(&pend[w0], ~(lo_mask(a) & hi_mask(b))); return; }
    COVW_ATOMIC_AND(&pend[w0], ~lo_mask(a));
    for (u32 w = w0 + 1u; w < w1; w++) COVW_ATOMIC_AND(&pend[w], 0u);
    COVW_ATOMIC_AND(&pend[w1], ~hi_mask(b));
}
COVW_FN bool any_in_range(const u32 *pend, u32 a, u32 b) {
    const u32 w0 = a >> 5, w1 = (b - 1u) >> 5;
    u32 x = COVL_LD(&pend[w0]) & lo_mask(a);
    if (w0 == w1) return (x & hi_mask(b)) != 0u;
    for (u32 w = w0 + 1u; w < w1; w++) x |= COVL_LD(&pend[w]);
    x |= COVL_LD(&pend[w1]) & hi_mask(b);
    return x != 0u;
}
// the lowest set bit in [from, end), NONE if there is none
COVW_FN u32 next_set(const u32 *pend, u32 from, u32 end) {
    if (from >= end) return NONE;
    u32 w = from >> 5;
    u32 x = COVL_LD(&pend[w]) & lo_mask(from);
    const u32 wl = (end - 1u) >> 5;
    while (x == 0u) {
        if (w >= wl) return NONE;
        w++;
        x = COVL_LD(&pend[w]);
    }
    const u32 p = (w << 5) + (u32)__builtin_ctz(x);
    return p < end ? p : NONE;
}

// Pass 3's output of one lane: the block's image in LDS.  Positions are image positions (the block's first byte at `bias`).
struct SinkLds {
    u8 *img; u32 *pend; u32 first;
    COVW_FN void init(u8 *i, u32 *p) { img = i; pend = p; first = NONE; }
    COVW_FN void literal(u32 p, u32 b) { img[p] = (u8)b; }
    COVW_FN void match(u32 p, u32 len, u32 t24, u32) {
        img[p] = (u8)t24; img[p + 1u] = (u8)(t24 >> 8); img[p + 2u] = (u8)(t24 >> 16);
        set_range(pend, p, p + len);
        if (first == NONE) first = p;
    }
    COVW_FN void finish(u32) {}
};

// ---- the resolve step of one lane: its matches in order, from W.tmp[lane] (the first one, NONE: none left) up to the end of its output
// W.cnt2[lane].  Returns with W.tmp[lane] = NONE when all are copied, or = the match whose source stayed pending for COVL_SPIN_LIMIT looks.
template <class BK>
COVW_FN void resolve_lane(BK &W, u32 lane) {
    u32 t = W.tmp[lane];
    const u32 own_end = W.cnt2[lane];
    u8 *img = W.img;
    u32 *pend = W.pend;
    u32 looks = 0;
    bool loaded = false, ready = false;
    u32 d = 0, n = 0, s = 0, k = 0, m = 0;
    while (t != NONE) {
        if (!loaded) {                              // the match at t: its token lies in its own first three bytes
            const u32 t24 = (u32)img[t] | ((u32)img[t + 1u] << 8) | ((u32)img[t + 2u] << 16);
            d = (t24 & 0x7fffu) + 1u; n = (t24 >> 15) + 3u; s = t - d;
            loaded = true; ready = false;
        }
        if (!ready) {
            if (any_in_range(pend, s, s + (n < d ? n : d))) {      // (an overlapping match repeats its first d bytes: only those are its source)
                if (++looks >= COVL_SPIN_LIMIT) break;
                COVL_RELAX();
                continue;
            }
            COVL_ACQUIRE();
            ready = true; k = 0; m = 0; looks = 0;
        }
        // up to CH bytes: all loads, then all stores (every source byte is final: byte k of the match = source byte k mod d)
        u8 r[CH];
        const u32 c = n - k < CH ? n - k : CH;
        u32 mm = m;
        for (u32 j = 0; j < CH; j++) {
            if (j < c) { r[j] = img[s + mm]; mm++; if (mm == d) mm = 0; }
        }
        for (u32 j = 0; j < CH; j++) if (j < c) img[t + k + j] = r[j];
        m = mm; k += c;
        if (k == n) {
            COVL_RELEASE();
            clear_range(pend, t, t + n);
            t = next_set(pend, t + n, own_end);     // what is still pending in this lane's range are this lane's later matches
            loaded = false;
        }
    }
    W.tmp[lane] = t;
}

// ---- tables of the DEFLATE block at `pos` (covw::inflate_block's first half, with NL lanes): W.hdr as covw::parse_header leaves it;
// Huffman blocks: both codes and their lookup tables in W.T, hdr[4] = first unit bit.  LDS_IN only names the instantiation: one per
// address space of s.w, so that the device code of each reads its bit stream with the right instruction.
template <bool LDS_IN, class BK>
COVW_FN void block_tables(BK &W, const Src &s, u32 pos) {
    COVL_PARFOR(lane) { if (lane == 0u) covw::parse_header(W, s, pos); }
    COVL_SYNC();
    if (W.hdr[6] != OK || W.hdr[0] == 0u) return;
    if (W.hdr[0] == 2u) {                           // dynamic codes: the code-length code's lookup table, then the code lengths
        COVL_PARFOR(lane) {
            if (lane < 128u) {
                u32 idx = 0;
                const u32 l = covw::canonical<0>(W.climit, W.coff, covw_brev32(lane) >> 17, idx);
                W.cltab[lane] = (u8)((l == 0u || l > 7u) ? 0u : (((u32)W.csorted[idx & 31u] << 3) | l));
            }
        }
        COVL_SYNC();
        COVL_PARFOR(lane) { if (lane == 0u) covw::parse_code_lengths(W, s); }
        COVL_SYNC();
        if (W.hdr[6] != OK) return;
    }
    COVL_PARFOR(lane) {
        if (lane < 16u * 9u + 16u) W.T.mask[lane] = 0;
        if (lane < 32u) W.cnt[lane] = 0;
    }
    COVL_SYNC();
    COVL_PARFOR(lane) {
        for (u32 sy = lane; sy < 320u; sy += NL) {
            const u32 l = W.T.lens[sy] & 15u;
            if (!l) continue;
            if (sy < 288u) { COVW_ATOMIC_ADD(&W.cnt[l], 1u); COVW_ATOMIC_OR(&W.T.mask[l * 9u + (sy >> 5)], 1u << (sy & 31u)); }
            else { COVW_ATOMIC_ADD(&W.cnt[16u + l], 1u); COVW_ATOMIC_OR(&W.T.mask[144u + l], 1u << (sy - 288u)); }
        }
    }
    COVL_SYNC();
    COVL_PARFOR(lane) {
        if (lane == 0u) {
            W.cnt[0] = 0; W.cnt[16] = 0;
            const bool a = covw::code_offsets(W.cnt, W.start, W.T.lit_limit, W.T.lit_off);
            const bool b = a && covw::code_offsets(W.cnt + 16, W.start + 16, W.T.dist_limit, W.T.dist_off);
            if (!b) W.hdr[6] = ERR_FORMAT;
        }
    }
    COVL_SYNC();
    if (W.hdr[6] != OK) return;
    COVL_PARFOR(lane) {
        for (u32 sy = lane; sy < 320u; sy += NL) {
            const u32 l = W.T.lens[sy] & 15u;
            if (!l) continue;
            if (sy < 288u) {
                const u32 *mk = W.T.mask + l * 9u, w = sy >> 5;
                u32 rank = (u32)__builtin_popcount(mk[w] & ((1u << (sy & 31u)) - 1u));
                for (u32 j = 0; j < w; j++) rank += (u32)__builtin_popcount(mk[j]);
                W.T.lit_sorted[W.start[l] + rank] = (u16)covw::lit_entry(sy);
            } else {
                const u32 dd = sy - 288u;
                W.T.dist_sorted[W.start[16u + l] + (u32)__builtin_popcount(W.T.mask[144u + l] & ((1u << dd) - 1u))] = (u8)dd;
            }
        }
    }
    COVL_SYNC();          // (the masks share their memory with the lookup table that is filled next)
    COVL_PARFOR(lane) {
        for (u32 i = lane; i < (1u << covw::LB); i += NL) covw::fill_lit_index(W.T, i);
        for (u32 i = lane; i < (1u << covw::DB); i += NL) covw::fill_dist_index(W.T, i);
    }
    COVL_SYNC();
}

// ---- passes 1 and 2 over the chunk [cur, span_end) in NL shares of S bits, and the lanes' places in the output.  Leaves W.end[] (exact unit
// boundaries), W.tmp[lane] = the lane's first output byte relative to the chunk's, W.cnt2[lane] = its bytes, W.n_valid / W.eob_at,
// hdr[7] = the chunk's bytes; hdr[6] on error.
template <bool LDS_IN, class BK>
COVW_FN void chunk_plan(BK &W, const Src &s, u32 cur, u32 span_end, u32 S) {
    COVL_PARFOR(lane) {
        u32 f, nb, nt;
        covw::NoSink ns;
        const u32 g = covw::share_begin_of(cur, S, lane, span_end);
        if (lane) W.end[lane - 1u] = g >= span_end ? span_end
                                                   : covw::run_share<0>(W.T, s, g - cur > OVERLAP_BITS ? g - OVERLAP_BITS : cur, g, &f, &nb, &nt, ns, 0, 0, 0, nullptr);
        if (lane == NL - 1u) W.end[NL - 1u] = span_end;
    }
    COVL_SYNC();
    for (u32 round = 0;; round++) {
        COVL_PARFOR(lane) {
            if (lane == 0u) { W.rounds = round + 1u; if (round == 0u) W.n_chunks++; W.changed = 0; W.first_flag = NL; }
            const u32 from = lane ? W.end[lane - 1u] : cur;
            const u32 ge = covw::share_begin_of(cur, S, lane + 1u, span_end);
            u32 f = 0, nb = 0, nt = 0, e = from;
            covw::NoSink ns;
            if (from < ge) e = covw::run_share<1>(W.T, s, from, ge, &f, &nb, &nt, ns, 0, 0, 0, nullptr);
            W.flags[lane] = (u8)f; W.cnt2[lane] = nb; W.tmp[lane] = e;
        }
        COVL_SYNC();          // every lane has read its neighbour's old end
        COVL_PARFOR(lane) { if (W.flags[lane]) COVW_ATOMIC_MIN(&W.first_flag, lane); }
        COVL_SYNC();
        // The lanes in front of F (the first lane that stopped at end-of-block or an invalid code) decide: when none of their ends moved,
        // they are a fixed point of the chain that starts at the exact `cur`, i.e. exact (see inflate_wave_core.h).
        COVL_PARFOR(lane) { if (lane < W.first_flag && lane < NL - 1u && W.tmp[lane] != W.end[lane]) COVW_ATOMIC_OR(&W.changed, 1u); }
        COVL_SYNC();
        COVL_PARFOR(lane) { W.end[lane] = W.tmp[lane]; }
        COVL_SYNC();
        if (!W.changed) break;
        if (round >= NL + 1u) { COVL_PARFOR(lane) { if (lane == 0u) W.hdr[6] = ERR_FORMAT; } COVL_SYNC(); return; }
    }
    // ---- the lanes up to the first one that met end-of-block are what is left of the block (all when none did); prefix sums of their bytes:
    // four lanes sum 64 lanes each, then every lane adds what lies in front of its group
    COVL_PARFOR(lane) {
        if ((lane & 63u) == 0u) {
            const u32 F = W.first_flag;
            u32 ob = 0;
            for (u32 i = lane; i < lane + 64u; i++) {
                const u32 nb = W.cnt2[i];
                W.tmp[i] = ob;
                if (i <= F) ob += nb < 0x20000u ? nb : 0x20000u;      // (a corrupt stream: keep the sums of 256 lanes inside 32 bits)
            }
            W.gsum[lane >> 6] = ob;
        }
        if (lane == 0u) {
            const u32 F = W.first_flag;
            W.n_valid = F < NL ? F + 1u : NL;
            W.eob_at = NO_EOB;
            if (F < NL) { if (W.flags[F] == 1u) W.eob_at = W.end[F]; else W.hdr[6] = ERR_FORMAT; }      // an invalid code, or a unit that runs off the payload
        }
    }
    COVL_SYNC();
    COVL_PARFOR(lane) {
        u32 off = 0;
        for (u32 g = 0; g < (lane >> 6); g++) off += W.gsum[g];
        W.tmp[lane] += off;
        if (lane == 0u) W.hdr[7] = W.gsum[0] + W.gsum[1] + W.gsum[2] + W.gsum[3];
    }
    COVL_SYNC();
}

// One BGZF block.  comp_words: aligned words holding the raw DEFLATE payload from bit `bit0` on (readable 64 bytes past its end); `staged`:
// W.cin holds a copy of them (payload + slack <= IMG_BYTES).  The block's `isize` bytes are assembled at W.img[bias ...] (bias < 16: the
// caller's choice, so that the image and its place in global memory agree modulo 16) and W.pend must be all zero — it is again afterwards.
// *status = OK / ERR_*.  stop_after (measurements only): 1 = give up after the tables are built, 2 = after the plan (passes 1 and 2),
// 3 = after pass 3.
template <class BK>
COVW_FN void inflate_block_lds(BK &W, const u32 *comp_words, u32 bit0, u32 payload_bits, bool staged, u32 bias, u32 isize, u32 *status, u32 stop_after = 0) {
    Src sg; sg.w = comp_words; sg.total_bits = bit0 + payload_bits;
    Src sl; sl.w = W.cin; sl.total_bits = sg.total_bits;
    u32 pos = bit0, opos = 0, err = OK, nblk = 0, chunk_bits = 0;
    bool last = false, lds_in = staged;
    while (!last && err == OK) {
        nblk++;
        COVL_PARFOR(lane) { if (lane == 0u) { W.n_deflate_blocks = nblk; if (nblk == 1u) W.n_chunks = 0; W.err3 = OK; } }
        if (lds_in) block_tables<true>(W, sl, pos); else block_tables<false>(W, sg, pos);
        if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
        last = W.hdr[1] != 0u;
        if (W.hdr[0] == 0u) {                       // stored block: the lanes copy its bytes (from global memory: the image may hold output already)
            const u32 len = W.hdr[5], src_bit = W.hdr[4];
            if (opos + len > isize) { err = ERR_FORMAT; break; }
            const u8 *src = reinterpret_cast<const u8 *>(sg.w) + (src_bit >> 3);
            COVL_SYNC();          // (the staged payload is read no more)
            COVL_PARFOR(lane) {
                COVW_NO_UNROLL
                for (u32 k = lane; k < len; k += NL) W.img[bias + opos + k] = src[k];
            }
            lds_in = false;
            opos += len; pos = src_bit + 8u * len;
            COVL_SYNC();
            continue;
        }
        if (stop_after == 1u) { err = ERR_FORMAT; break; }
        // ---- the block's units, CHUNK by chunk (see inflate_wave_core.h: the first DEFLATE block of a payload takes the whole payload as its
        // one chunk; if it ends early, the blocks behind it are taken in chunks of about its length)
        const u32 B0 = W.hdr[4];
        u32 cur = B0;
        for (bool eob = false; !eob && err == OK;) {
            if (cur >= sg.total_bits) { err = ERR_FORMAT; break; }            // no end-of-block inside the payload
            const u32 span_end = chunk_bits && sg.total_bits - cur > chunk_bits ? cur + chunk_bits : sg.total_bits;
            const u32 span = span_end - cur;
            const u32 S = span > NL * MIN_SHARE_BITS ? (span + NL - 1u) / NL : MIN_SHARE_BITS;
            if (lds_in) chunk_plan<true>(W, sl, cur, span_end, S); else chunk_plan<false>(W, sg, cur, span_end, S);
            if (W.hdr[6] != OK) { err = W.hdr[6]; break; }
            if (opos + W.hdr[7] > isize) { err = ERR_SIZE; break; }
            if (stop_after == 2u) { err = ERR_FORMAT; break; }
            // ---- pass 3: literals and tokens into the image, matches marked pending (bit stream from global memory)
            lds_in = false;
            COVL_PARFOR(lane) {
                u32 first = NONE, own_end = 0;
                if (lane < W.n_valid) {
                    const u32 from = lane ? W.end[lane - 1u] : cur;
                    const u32 ge = covw::share_begin_of(cur, S, lane + 1u, span_end);
                    const u32 p0 = bias + opos + W.tmp[lane];
                    u32 f, nb, nt, e2 = OK;
                    SinkLds sink; sink.init(W.img, W.pend);
                    if (from < ge) (void)covw::run_share<2>(W.T, sg, from, ge, &f, &nb, &nt, sink, p0, bias, 0, &e2);
                    if (e2 != OK) W.err3 = e2;
                    first = sink.first; own_end = p0 + W.cnt2[lane];
                }
                W.tmp[lane] = first; W.cnt2[lane] = own_end;
            }
            COVL_SYNC();
            if (W.err3 != OK) { err = W.err3; break; }
            if (stop_after == 3u) { err = ERR_FORMAT; break; }
            // ---- resolve: on the device every lane stays in resolve_lane until its matches are copied (one trip through this loop)
            for (;;) {
                COVL_PARFOR(lane) { if (lane == 0u) W.remaining = 0; }
                COVL_SYNC();
                COVL_PARFOR(lane) {
                    resolve_lane(W, lane);
                    if (W.tmp[lane] != NONE) COVW_ATOMIC_OR(&W.remaining, 1u);
                }
                COVL_SYNC();
                if (!W.remaining) break;
            }
            opos += W.hdr[7];
            if (W.eob_at != NO_EOB) { eob = true; pos = W.eob_at; }
            else cur = W.end[NL - 1u];                                       // exact: the chunk's last unit ends here
        }
        if (err != OK) break;
        { const u32 body = pos - B0; chunk_bits = body + (body >> 3); if (chunk_bits < NL * MIN_SHARE_BITS) chunk_bits = NL * MIN_SHARE_BITS; }
    }
    if (err == OK && opos != isize) err = ERR_SIZE;
    *status = err;
}

}  // namespace covl
