// AddressSanitizer + UBSan run of csrc/inflate_wave_core.h over valid and damaged DEFLATE streams: the buffers are exactly as large as the
// kernel's contract says (payload + 64 readable bytes, isize output bytes, TOK_CAP token positions), so any access outside them aborts.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -o fuzz tests/c/inflate_wave_fuzz.cpp -lz && ./fuzz [rounds] [seed]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#define COVW_FN inline
#define COVW_PARFOR(lane) for (unsigned lane = 0; lane < 64u; lane++)
#define COVW_SYNC() do { } while (0)
static inline unsigned covw_brev32(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(x);
}
#include "../../coverm_amd/csrc/inflate_wave_core.h"

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

// pieces > 1: the input goes in that many pieces with a flush between them, i.e. several DEFLATE blocks (and empty stored blocks) per stream
static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &in, int level, int strategy, int pieces) {
    z_stream z; memset(&z, 0, sizeof z);
    deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(in.size() * 2 + 4096 + 64 * (size_t)pieces);      // (deflateBound does not cover Z_FIXED on incompressible input)
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    size_t at = 0;
    for (int k = 0; k < pieces; k++) {
        const size_t end = k + 1 == pieces ? in.size() : at + rnd() % (in.size() - at + 1);
        z.next_in = const_cast<uint8_t *>(in.data()) + at; z.avail_in = (uInt)(end - at);
        const int flush = k + 1 == pieces ? Z_FINISH : (rnd() & 1u) ? Z_FULL_FLUSH : Z_BLOCK;
        const int rc = deflate(&z, flush);
        if (k + 1 == pieces ? rc != Z_STREAM_END : (rc != Z_OK && rc != Z_BUF_ERROR)) { fprintf(stderr, "deflate did not finish\n"); exit(2); }
        at = end;
    }
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// -> status; `want` non-null: the resolved bytes must equal it when the status is OK
static int run(const std::vector<uint8_t> &payload, uint32_t misalign, uint32_t isize, const std::vector<uint8_t> *want) {
    const size_t nbytes = misalign + payload.size() + 64;                    // the contract: 64 readable bytes behind the payload
    uint32_t *words = static_cast<uint32_t *>(malloc((nbytes + 3) / 4 * 4));
    for (size_t k = 0; k < (nbytes + 3) / 4; k++) words[k] = rnd();
    memcpy(reinterpret_cast<uint8_t *>(words) + misalign, payload.data(), payload.size());
    const uint32_t off = rnd() & 63u;                                        // a block's output begins at any address
    uint8_t *out_alloc = static_cast<uint8_t *>(malloc(isize + off + (isize + off ? 0 : 1)));
    memset(out_alloc, 0x5A, off);
    uint8_t *out = out_alloc + off;
    const uint32_t toff = rnd() & 3u;                                        // the list begins at any 2-byte address
    uint16_t *tok_alloc = static_cast<uint16_t *>(malloc((covw::TOK_CAP + toff) * 2));
    uint16_t *tok = tok_alloc + toff;
    uint32_t nt = 0, st = 0;
    const uint32_t b0 = 8u * misalign, nb = 8u * (uint32_t)payload.size();
    static covw::Wave W;
    covw::inflate_block(W, words, b0, nb, out, isize, tok, &nt, &st, 0);
    int rc = (int)st;
    if (st == covw::OK) {

        for (uint32_t t = 0; t < nt; t++) {                                  // k_lz_resolve, serially
            const uint32_t p = tok[t];
            if (p + 3 > isize) { rc = -2; break; }
            const uint32_t t24 = (uint32_t)out[p] | ((uint32_t)out[p + 1] << 8) | ((uint32_t)out[p + 2] << 16);
            const uint32_t dist = (t24 & 0x7fffu) + 1u, len = (t24 >> 15) + 3u;
            if (dist > p || p + len > isize) { rc = -2; break; }
            for (uint32_t k = 0; k < len; k++) out[p + k] = out[p + k - dist];
        }
        if (rc == 0 && want && (want->size() != isize || memcmp(want->data(), out, isize) != 0)) rc = -3;
    }
    for (uint32_t k = 0; k < off; k++) if (out_alloc[k] != 0x5A) rc = -4;    // wrote in front of the block
    free(words); free(out_alloc); free(tok_alloc);
    return rc;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    if (argc > 2) rng_state ^= strtoull(argv[2], nullptr, 10) * 0x9e3779b97f4a7c15ull;
    long ok = 0, rejected = 0, differ = 0;
    for (int r = 0; r < rounds; r++) {
        const uint32_t size = 1 + rnd() % 65280u;
        std::vector<uint8_t> data(size);
        const uint32_t kind = rnd() % 4u;
        for (uint32_t k = 0; k < size; k++)
            data[k] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? (uint8_t)("ACGTN!#I"[rnd() & 7u]) : kind == 2 ? (uint8_t)(rnd() % 3u ? 0 : rnd()) : (uint8_t)(k * 7u >> (rnd() & 3u));
        const int level = (int)(rnd() % 10u), strategy = (rnd() & 7u) == 0 ? Z_FIXED : (rnd() & 7u) == 1 ? Z_HUFFMAN_ONLY : (rnd() & 7u) == 2 ? Z_RLE : Z_DEFAULT_STRATEGY;
        const std::vector<uint8_t> comp = deflate_raw(data, level, strategy, (rnd() & 3u) ? 1 : 2 + (int)(rnd() % 6u));
        const int a = run(comp, rnd() & 3u, size, &data);
        if (a != 0) {
            fprintf(stderr, "round %d: valid stream (size %u level %d strategy %d) -> %d\n", r, size, level, strategy, a);
            if (FILE *f = fopen("/tmp/covw_fuzz_fail.bin", "wb")) { fwrite(comp.data(), 1, comp.size(), f); fclose(f); }
            return 1;
        }
        ok++;
        for (int m = 0; m < 6; m++) {                                        // damaged copies: any status is fine, any bad access aborts
            std::vector<uint8_t> bad = comp;
            const uint32_t how = rnd() % 5u;
            if (how == 0) bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() & 7u));
            else if (how == 1) bad.resize(rnd() % bad.size());
            else if (how == 2) for (int k = 0; k < 8; k++) bad[rnd() % bad.size()] = (uint8_t)rnd();
            else if (how == 3) { const size_t at = rnd() % bad.size(); for (size_t k = at; k < bad.size(); k++) bad[k] = (uint8_t)rnd(); }
            else bad[0] = (uint8_t)rnd();
            const uint32_t isz = (rnd() & 3u) ? size : (rnd() % 65536u);
            const int b = run(bad, rnd() & 3u, isz, nullptr);
            if (b == -2 || b == -4) { fprintf(stderr, "round %d: status OK with a token outside the block, or bytes written in front of it (%d)\n", r, b); return 1; }
            if (b != 0) rejected++; else differ++;
        }
    }
    printf("%ld valid streams exact, %ld damaged rejected, %ld damaged decoded to something (the CRC pass judges those)\n", ok, rejected, differ);
    return 0;
}
