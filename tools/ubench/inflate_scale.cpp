// Host micro-benchmark: how does BGZF inflate scale with threads on this box, and what limits it?
//   g++ -O3 -std=c++17 tools/ubench/inflate_scale.cpp -o /tmp/inflate_scale -ldl -lz -lpthread && /tmp/inflate_scale file.bam
// Modes: (a) inflate into one big pre-touched buffer (DRAM writes), (b) into a 64 KiB per-thread scratch (cache only),
//        (c) into a big buffer that is first touched by the inflate itself (page faults on the critical path).
#include <dlfcn.h>
#include <sys/mman.h>
#include <zlib.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Blk { size_t in_off, in_len, out_off; uint32_t isize; };
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    if (n > (6ull << 30)) n = 6ull << 30;
    std::vector<uint8_t> raw(n); n = fread(raw.data(), 1, n, f); fclose(f);
    std::vector<Blk> blocks; size_t p = 0, total = 0;
    while (p + 28 <= n) {
        const size_t bsize = (size_t)(raw[p + 16] | (raw[p + 17] << 8)) + 1;
        if (p + bsize > n) break;
        uint32_t isz; memcpy(&isz, &raw[p + bsize - 4], 4);
        blocks.push_back({p + 18, bsize - 26, total, isz}); total += isz; p += bsize;
    }
    printf("%zu blocks, %.2f GB compressed, %.2f GB inflated\n", blocks.size(), p / 1e9, total / 1e9);
    void *h = dlopen("libdeflate.so.0", RTLD_NOW);
    auto alloc = h ? (void *(*)())dlsym(h, "libdeflate_alloc_decompressor") : nullptr;
    auto dec = h ? (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress") : nullptr;
    printf("libdeflate: %s\n", dec ? "yes" : "no (zlib)");
    auto inflate1 = [&](void *d, const Blk &b, uint8_t *dst) {
        if (dec) { size_t got; dec(d, raw.data() + b.in_off, b.in_len, dst, b.isize, &got); return; }
        z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
        zs.next_in = raw.data() + b.in_off; zs.avail_in = b.in_len; zs.next_out = dst; zs.avail_out = b.isize; inflate(&zs, Z_FINISH); inflateEnd(&zs);
    };
    uint8_t *big = (uint8_t *)mmap(nullptr, total + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(big, total, MADV_HUGEPAGE);
    for (int mode : {2, 0, 1}) {
        for (int T : {8, 16, 32, 64, 128, 256}) {
            if (mode == 2 && T != 64) continue;   // first-touch run only once (pages stay afterwards)
            std::atomic<size_t> next{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] {
                void *d = alloc ? alloc() : nullptr;
                std::vector<uint8_t> scratch(65536);
                for (;;) {
                    const size_t i0 = next.fetch_add(16);
                    if (i0 >= blocks.size()) break;
                    for (size_t i = i0; i < std::min(blocks.size(), i0 + 16); i++) inflate1(d, blocks[i], mode == 1 ? scratch.data() : big + blocks[i].out_off);
                }
            });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            printf("mode %s threads %3d: %.3fs = %.1f GB/s inflated\n", mode == 2 ? "first-touch" : mode == 0 ? "pretouched " : "scratch    ", T, dt, total / dt / 1e9);
        }
    }
    // memory write bandwidth reference: memset of the same buffer with T threads
    for (int T : {16, 64, 256}) {
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] { memset(big + total / T * t, t, total / T); });
        for (auto &x : th) x.join();
        printf("memset threads %3d: %.1f GB/s\n", T, total / (now() - t0) / 1e9);
    }
    return 0;
}
