// Lock-step cost model of k_inflate_wave on real blocks: runs csrc/inflate_wave_core.h lane by lane on the CPU, records what every lane
// did in every step of every pass, and replays the 64 traces in lock step the way a wave executes them — a code path costs its
// instructions whenever ANY lane takes it.
//   g++ -O2 -DCOVW_LB=11 -DCOVW_DB=9 -o /tmp/wave_cost tools/proto/wave_cost_model.cpp && /tmp/wave_cost file.bam [max_blocks]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

static unsigned g_lane = 0;
static std::vector<uint8_t> g_trace[3][64];
#define COVW_FN inline
#define COVW_PARFOR(lane) for (unsigned lane = 0; lane < 64u && ((g_lane = lane), true); lane++)
#define COVW_SYNC() do { } while (0)
#define COVW_TRACE_UNIT(mode, flags) g_trace[mode][g_lane].push_back((uint8_t)(flags))
// store shapes: [width 1 2 4 8 16][aligned to its width?] lane-stores, and bytes written
static double g_stores[5][2], g_store_bytes[2];
static inline void trace_store(const void *p, unsigned w) {
    const int wi = w == 1 ? 0 : w == 2 ? 1 : w == 4 ? 2 : w == 8 ? 3 : 4, al = ((uintptr_t)p & (w - 1)) == 0;
    g_stores[wi][al]++; g_store_bytes[al] += w;
}
#define COVW_TRACE_STORE(ptr, width) trace_store((ptr), (width))
static inline unsigned covw_brev32(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(x);
}
#include "../../coverm_amd/csrc/inflate_wave_core.h"

struct Acc { double steps = 0, lit_slow = 0, match = 0, dist_slow = 0, active = 0, u_iters = 0, u_slow = 0, u_dist = 0, u_active = 0, units = 0, matches = 0; };

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(n);
    if (fread(buf.data(), 1, n, f) != (size_t)n) return 2;
    fclose(f);
    const long max_blocks = argc > 2 ? atol(argv[2]) : 400;
    static covw::Wave W;
    std::vector<uint8_t> out_v(65536 + 64 + 64);
    uint8_t *const out_al = out_v.data() + (64 - ((uintptr_t)out_v.data() & 63));      // 64-byte aligned; the blocks' output begins at varying offsets behind it
    std::vector<uint16_t> tok(covw::TOK_CAP);
    Acc A[3];
    long p = 0, nb = 0, rounds = 0;
    while (p + 18 <= n && nb < max_blocks) {
        const uint32_t xlen = buf[p + 10] | (buf[p + 11] << 8);
        const uint32_t bsize = (buf[p + 16] | (buf[p + 17] << 8)) + 1u;      // files of our own writer / htslib: BC is the only subfield
        const uint32_t isize = buf[p + bsize - 4] | (buf[p + bsize - 3] << 8) | (buf[p + bsize - 2] << 16) | ((uint32_t)buf[p + bsize - 1] << 24);
        const uint8_t *pay = buf.data() + p + 12 + xlen;
        const uint32_t plen = bsize - 12 - xlen - 8;
        p += bsize;
        if (isize < 30000) continue;
        std::vector<uint32_t> words(plen / 4 + 20, 0);
        memcpy(words.data(), pay, plen);
        for (int m = 0; m < 3; m++) for (int l = 0; l < 64; l++) g_trace[m][l].clear();
        uint32_t nt = 0, st = 0;
        covw::inflate_block(W, words.data(), 0, 8u * plen, out_al + (nb * 13) % 64, isize, tok.data(), &nt, &st, 0);
        if (st != 0) { fprintf(stderr, "block %ld: status %u\n", nb, st); return 1; }
        rounds += W.rounds;
        for (int m = 0; m < 3; m++) {
            size_t steps = 0;
            for (int l = 0; l < 64; l++) steps = std::max(steps, g_trace[m][l].size());
            for (size_t t = 0; t < steps; t++) {
                unsigned any = 0, act = 0;
                for (int l = 0; l < 64; l++) if (t < g_trace[m][l].size()) { any |= g_trace[m][l][t]; act++; }
                A[m].lit_slow += any & 1u; A[m].match += (any >> 1) & 1u; A[m].dist_slow += (any >> 2) & 1u; A[m].active += act;
            }
            A[m].steps += steps;
            // one Huffman symbol per iteration (a match takes two: length, then distance)
            std::vector<uint8_t> sym[64];
            size_t iters = 0;
            for (int l = 0; l < 64; l++) {
                for (uint8_t u : g_trace[m][l]) {
                    sym[l].push_back(u & 1u);                                 // bit 0 slow
                    if (u & 2u) sym[l].push_back(2u | ((u >> 2) & 1u));       // bit 1 distance symbol
                    A[m].units++; A[m].matches += (u >> 1) & 1u;
                }
                iters = std::max(iters, sym[l].size());
            }
            for (size_t t = 0; t < iters; t++) {
                unsigned any = 0, act = 0;
                for (int l = 0; l < 64; l++) if (t < sym[l].size()) { any |= sym[l][t]; act++; }
                A[m].u_slow += any & 1u; A[m].u_dist += (any >> 1) & 1u; A[m].u_active += act;
            }
            A[m].u_iters += iters;
        }
        nb++;
    }
    printf("LB %u DB %u: %ld blocks, pass-2 rounds per block %.3f\n", covw::LB, covw::DB, nb, (double)rounds / nb);
    printf("lane-stores per block (width: unaligned + aligned):");
    const int widths[5] = {1, 2, 4, 8, 16};
    double tot = 0;
    for (int w = 0; w < 5; w++) { if (g_stores[w][0] + g_stores[w][1] > 0) printf("  %dB: %.0f + %.0f", widths[w], g_stores[w][0] / nb, g_stores[w][1] / nb); tot += g_stores[w][0] + g_stores[w][1]; }
    printf("  = %.0f stores, %.0f bytes through unaligned and %.0f through aligned ones\n", tot / nb, g_store_bytes[0] / nb, g_store_bytes[1] / nb);
    const char *names[3] = {"pass 1 (positions)", "pass 2 (count)    ", "pass 3 (write)    "};
    for (int m = 0; m < 3; m++) {
        const Acc &a = A[m];
        printf("%s: %.0f lock-steps/block (units/lane %.0f, matches %.2f of units), P(any lit slow) %.2f, P(any match) %.2f, P(any dist slow) %.2f, active lanes %.1f\n",
               names[m], a.steps / nb, a.units / nb / 64, a.matches / a.units, a.lit_slow / a.steps, a.match / a.steps, a.dist_slow / a.steps, a.active / a.steps);
        printf("      one symbol per iteration: %.0f iterations/block, P(any slow) %.2f, P(any distance symbol) %.2f, active lanes %.1f\n", a.u_iters / nb,
               a.u_slow / a.u_iters, a.u_dist / a.u_iters, a.u_active / a.u_iters);
    }
    return 0;
}
