// k_crc32_wave — the CRC-32 of a BGZF block by ONE WAVE with coalesced loads: the arithmetic, written once, run two ways (the kernel in
// csrc/ingest_kernels.hip.h; lane by lane on the CPU in tests/c/crc_wave_host.cpp against zlib's crc32).  What is checked is the trailer
// htslib checks in bgzf.c:inflate_block for every block the reference reads (bam_generator.rs:125-129 hands the file to it).
//
// The lane-per-block kernel (k_crc32) walks 64 blocks with 64 private pointers: every load of the wave touches 64 cache lines, a block is a
// chain of 8 192 dependent steps and a round of 81 920 blocks is only 1 280 waves — 5.6 ms per round of 5.2 GB.  Here the block is a
// matrix of 8-byte words, 64 to a row (512 bytes, one coalesced load of the wave), and lane l owns column l:
//
//   * the CRC register is linear in (register, message) over GF(2), so the CRC of the block is the XOR of the CRCs of its columns, each
//     taken with the other columns' bytes as zeros.  A lane that has absorbed its word of row j is 512 bytes — 8 of its own, 504 of zeros —
//     away from its word of row j + 1: one step with tables built for that distance (LO512 / HI512) instead of slicing-by-8's
//     distance of 8.  Same eight lookups per eight bytes;
//   * in the last full row the lanes step by 8 (the ordinary slicing-by-8 tables), which leaves lane l exactly 8 (63 - l) bytes short of the
//     row's end: an inclusive scan over the lanes with the operator (left, right) -> advance(left, 8 * 2^m) ^ right, six levels with one
//     4-table set per distance (LO8 .. LO256), gives lane 63 the register at the row's end;
//   * the words of the last, partial row go to the HIGHEST lanes, so that the same scan finishes them; the last 0-7 bytes bytewise;
//   * a block starts at any byte: the words are read from the 8-byte boundary in front of it with the foreign bytes zeroed (leading zeros
//     do not move a zero register), and the initial value 0xffffffff is XORed into the message's first four bytes, which is the same thing.
//
// Tables (36 KiB, built on the host by crcw_build_tables, copied to LDS by every workgroup): a set is 4 x 256 words, LO_D[j][b] = the
// register value (b << 8 j) advanced over D zero bytes, HI_D[j][b] = the same over D - 4 (the upper half of an 8-byte word enters 4 bytes
// later).  Order: LO8 HI8 LO16 LO32 LO64 LO128 LO256 LO512 HI512.  HI8[3] is the classic byte table.
#pragma once
#include <stdint.h>

#ifndef CRCW_FN
#define CRCW_FN inline
#endif

namespace crcw {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned char u8;

constexpr u32 SET = 1024u;                 // words per 4-table set
constexpr u32 T_LO8 = 0u, T_HI8 = SET, T_LO16 = 2u * SET, T_LO512 = 7u * SET, T_HI512 = 8u * SET;
constexpr u32 TABLE_WORDS = 9u * SET;      // 36 KiB
constexpr u32 SMALL = 16u;                 // blocks shorter than this are done bytewise (the empty EOF block among them)

// the register advanced over D zero bytes, D = the distance of the set at T + lo
CRCW_FN u32 advance(const u32 *T, u32 lo, u32 s) {
    return T[lo + (s & 0xffu)] ^ T[lo + 256u + ((s >> 8) & 0xffu)] ^ T[lo + 512u + ((s >> 16) & 0xffu)] ^ T[lo + 768u + (s >> 24)];
}
// one 8-byte word absorbed, the register then standing D bytes behind the word's first byte
CRCW_FN u32 step(const u32 *T, u32 lo, u32 hi, u32 s, u64 w) {
    const u32 d = (u32)(w >> 32);
    return advance(T, lo, s ^ (u32)w) ^ T[hi + (d & 0xffu)] ^ T[hi + 256u + ((d >> 8) & 0xffu)] ^ T[hi + 512u + ((d >> 16) & 0xffu)] ^ T[hi + 768u + (d >> 24)];
}
CRCW_FN u32 byte_step(const u32 *T, u32 s, u32 b) { return T[T_HI8 + 768u + ((s ^ b) & 0xffu)] ^ (s >> 8); }

// Where a block's bytes lie relative to the 8-byte grid.
struct Shape {
    const u8 *base;      // 8-byte boundary at or in front of the block's first byte
    u32 a;               // foreign bytes in front (0..7)
    u32 rows;            // full rows of 64 words in [base, end)
    u32 tail_words;      // full words behind them (0..63)
    u32 tail_bytes;      // bytes behind those (0..7)
};
CRCW_FN Shape shape_of(const u8 *p, u32 n) {
    Shape S;
    S.a = (u32)((uintptr_t)p & 7u);
    S.base = p - S.a;
    const u32 m = n + S.a;
    S.rows = m >> 9; S.tail_words = (m & 511u) >> 3; S.tail_bytes = m & 7u;
    return S;
}
// word `idx` of the block's grid, as the arithmetic wants it (the caller loads it: on the device that is the coalesced part)
CRCW_FN u64 fix_word(const Shape &S, u32 idx, u64 w) {
    if (idx == 0u) { w &= ~0ull << (8u * S.a); w ^= 0xffffffffull << (8u * S.a); }
    else if (idx == 1u && S.a > 4u) w ^= 0xffffffffull >> (64u - 8u * S.a);
    return w;
}

// One level of the scan: `left` = the value of lane (lane - 2^m), 0 for the lanes that have none.
CRCW_FN u32 scan_combine(const u32 *T, u32 m, u32 left, u32 own) { return advance(T, m == 0u ? T_LO8 : T_LO16 + (m - 1u) * SET, left) ^ own; }

// The last 0-7 bytes and the final XOR.  `w` = the 8 bytes at base + 8 (64 rows + tail_words) (only the first tail_bytes of them count).
CRCW_FN u32 finish(const u32 *T, const Shape &S, u32 s, u64 w) {
    for (u32 k = 0; k < S.tail_bytes; k++) s = byte_step(T, s, (u32)(w >> (8u * k)) & 0xffu);
    return s ^ 0xffffffffu;
}

// Blocks below SMALL bytes.
CRCW_FN u32 small_block(const u32 *T, const u8 *p, u32 n) {
    u32 s = 0xffffffffu;
    for (u32 k = 0; k < n; k++) s = byte_step(T, s, p[k]);
    return s ^ 0xffffffffu;
}

// ---- the tables (host)
inline void build_tables(u32 *T) {
    u32 t0[256], S8[8][256];
    for (u32 i = 0; i < 256u; i++) {
        u32 c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        t0[i] = c;
    }
    for (u32 i = 0; i < 256u; i++) {
        u32 c = t0[i];
        S8[0][i] = c;
        for (int k = 1; k < 8; k++) { c = (c >> 8) ^ t0[c & 0xffu]; S8[k][i] = c; }       // one more zero byte each
    }
    for (u32 j = 0; j < 4u; j++)
        for (u32 b = 0; b < 256u; b++) { T[T_LO8 + 256u * j + b] = S8[7 - j][b]; T[T_HI8 + 256u * j + b] = S8[3 - j][b]; }
    // doubling: LO_2D[j][b] = LO_D applied to LO_D[j][b]
    u32 prev = T_LO8;
    for (u32 lvl = 0; lvl < 6u; lvl++) {            // -> LO16, LO32, LO64, LO128, LO256, LO512
        const u32 next = T_LO16 + lvl * SET;
        for (u32 e = 0; e < SET; e++) T[next + e] = advance(T, prev, T[prev + e]);
        prev = next;
    }
    // HI512 = HI8 advanced over 504 = 256 + 128 + 64 + 32 + 16 + 8 zero bytes
    for (u32 e = 0; e < SET; e++) {
        u32 v = T[T_HI8 + e];
        v = advance(T, T_LO8, v);
        for (u32 lvl = 0; lvl < 5u; lvl++) v = advance(T, T_LO16 + lvl * SET, v);
        T[T_HI512 + e] = v;
    }
}

}  // namespace crcw
