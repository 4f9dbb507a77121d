// Host instantiation of csrc/inflate_wave_core.h: the 64 lanes of the wave are a loop, the barrier is nothing.  Test infrastructure
// (tests/test_inflate_wave_core.py builds it with g++ and compares every block with zlib); not part of the product.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define COVW_FN inline
#ifdef COVW_REVERSE      // the lanes of a region run concurrently on the device: no result may depend on their order here
#define COVW_PARFOR(lane) for (unsigned lane##_r = 0, lane = 63u; lane##_r < 64u; lane##_r++, lane = 63u - lane##_r)
#else
#define COVW_PARFOR(lane) for (unsigned lane = 0; lane < 64u; lane++)
#endif
#define COVW_SYNC() do { } while (0)
static inline unsigned covw_brev32(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(x);
}
#include "../../coverm_amd/csrc/inflate_wave_core.h"

static covw::Wave g_wave;

extern "C" {

// payload: raw DEFLATE stream of one BGZF block; misalign 0..3 = where its first byte sits in the aligned word buffer (the device reads
// the compressed file in place: a block starts at any byte).  out must hold isize + 16 bytes: [0, 8) and [8 + isize, 16 + isize) are
// canaries this function checks.  Returns the status (covw::OK ...), -1 when a canary was overwritten.
int covw_host_inflate(const uint8_t *payload, uint32_t nbytes, uint32_t misalign, uint32_t slack_fill, uint8_t *out, uint32_t isize, uint16_t *tok, uint32_t *n_tok,
                      uint32_t *rounds) {
    std::vector<uint32_t> words((nbytes + misalign + 3) / 4 + 20);
    memset(words.data(), (int)slack_fill, words.size() * 4);
    memcpy(reinterpret_cast<uint8_t *>(words.data()) + misalign, payload, nbytes);
    memset(out, 0xC3, 8); memset(out + 8 + isize, 0xC3, 8);
    uint32_t status = 0;
    covw::Wave &W = g_wave;
    W.rounds = 0;
#ifdef COVW_SINK_OLD      // the 8-byte sink (k_inflate_wave8, COVERM_INFLATE_SINK=8)
    covw::inflate_block<covw::Sink>(W, words.data(), 8u * misalign, 8u * nbytes, out + 8, isize, tok, n_tok, &status, 0);
#elif defined(COVW_EXTRA_LITS)      // more literals per lock-step than the shipped kernel takes
    covw::inflate_block<covw::Sink16, COVW_EXTRA_LITS>(W, words.data(), 8u * misalign, 8u * nbytes, out + 8, isize, tok, n_tok, &status, 0);
#else
    covw::inflate_block(W, words.data(), 8u * misalign, 8u * nbytes, out + 8, isize, tok, n_tok, &status, 0);
#endif
    if (rounds) *rounds = W.rounds;
    for (int k = 0; k < 8; k++) if (out[k] != 0xC3 || out[8 + isize + k] != 0xC3) return -1;
    return (int)status;
}

// What k_lz_resolve does, serially: the tokens in list order; a token sits in the first three bytes of its own match.
int covw_host_resolve(uint8_t *out, uint32_t isize, const uint16_t *tok, uint32_t n_tok) {
    for (uint32_t t = 0; t < n_tok; t++) {
        const uint32_t p = tok[t];
        if (p + 3 > isize) return 1;
        const uint32_t t24 = (uint32_t)out[p] | ((uint32_t)out[p + 1] << 8) | ((uint32_t)out[p + 2] << 16);
        const uint32_t dist = (t24 & 0x7fffu) + 1u, len = (t24 >> 15) + 3u;
        if (dist > p || p + len > isize) return 2;
        for (uint32_t k = 0; k < len; k++) out[p + k] = out[p + k - dist];
    }
    return 0;
}

uint32_t covw_host_wave_bytes(void) { return (uint32_t)sizeof(g_wave); }
uint32_t covw_host_last_deflate_blocks(void) { return g_wave.n_deflate_blocks; }
void covw_host_last_share_bytes(uint32_t *nbytes64, uint32_t *ntok64) { for (int i = 0; i < 64; i++) { nbytes64[i] = i < (int)g_wave.n_valid ? g_wave.nbytes[i] : 0; ntok64[i] = i < (int)g_wave.n_valid ? g_wave.ntok[i] : 0; } }
uint32_t covw_host_last_chunks(void) { return g_wave.n_chunks; }
}
