// Host instantiation of csrc/crc_wave_core.h: the 64 lanes of the wave are a loop, a shuffle is an array access.  Test infrastructure
// (tests/test_crc_wave_core.py builds it with g++ and compares with zlib's crc32); not part of the product.
#include <stdint.h>
#include <string.h>

#include "../../coverm_amd/csrc/crc_wave_core.h"

using namespace crcw;

static u32 g_T[TABLE_WORDS];
static bool g_built = false;

static void scan(u32 *s) {
    for (u32 m = 0; m < 6u; m++) {
        u32 o[64];
        memcpy(o, s, sizeof o);
        for (u32 l = 0; l < 64u; l++) s[l] = scan_combine(g_T, m, l >= (1u << m) ? o[l - (1u << m)] : 0u, o[l]);
    }
}

extern "C" {
const uint32_t *crcw_host_tables(void) { if (!g_built) { build_tables(g_T); g_built = true; } return g_T; }
uint32_t crcw_host_table_words(void) { return TABLE_WORDS; }

// p: the block's first byte inside a buffer with at least 8 bytes on either side; reverse = the lanes run 63..0 (no result may depend on it)
uint32_t crcw_host_crc(const uint8_t *p, uint32_t n, int reverse) {
    crcw_host_tables();
    if (n < SMALL) return small_block(g_T, p, n);
    const Shape S = shape_of(p, n);
    auto load = [&](u32 idx) { u64 w; memcpy(&w, S.base + 8ull * idx, 8); return fix_word(S, idx, w); };
    u32 C = 0, s[64];
    if (S.rows) {
        for (u32 k = 0; k < 64u; k++) {
            const u32 lane = reverse ? 63u - k : k;
            u32 r = 0;
            for (u32 j = 0; j + 1u < S.rows; j++) r = step(g_T, T_LO512, T_HI512, r, load(64u * j + lane));
            s[lane] = step(g_T, T_LO8, T_HI8, r, load(64u * (S.rows - 1u) + lane));
        }
        scan(s);
        C = s[63];
    }
    if (S.tail_words) {
        for (u32 k = 0; k < 64u; k++) {
            const u32 lane = reverse ? 63u - k : k;
            const int lp = (int)lane - (int)(64u - S.tail_words);
            s[lane] = lp >= 0 ? step(g_T, T_LO8, T_HI8, lp == 0 ? C : 0u, load(64u * S.rows + (u32)lp)) : 0u;
        }
        scan(s);
        C = s[63];
    }
    u64 w;
    memcpy(&w, S.base + 8ull * (64u * S.rows + S.tail_words), 8);
    return finish(g_T, S, C, w);
}
}
