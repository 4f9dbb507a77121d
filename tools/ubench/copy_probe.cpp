// copy_probe: how fast do N threads move a file in the page cache (tmpfs) into page-locked memory — the feed of the device ingest, which the
// end-to-end run is bound by?
//   hipcc -O2 -std=c++17 -mavx2 tools/ubench/copy_probe.cpp -o tools/ubench/copy_probe -lpthread
//   tools/ubench/copy_probe <file> [threads] [GiB to move]
// Ways measured over the same bytes, 512 KiB chunks handed out by an atomic counter, destination = a 1 GiB page-locked ring:
//   pread        pread(fd, pinned + o, chunk, off)                      (what covh_bam_gpu_ingest's reader does; the kernel's copy_to_user)
//   map          memcpy from a MAP_SHARED mapping, first touch faults page by page
//   map+nt       the same with non-temporal 32-byte stores (no read-for-ownership of the destination lines)
//   ...+zap      madvise(MADV_DONTNEED) on the chunk afterwards (so that the mapping's page tables do not pile up until the process ends)
//   map+pop(+nt) madvise(MADV_POPULATE_READ) on the chunk first
// each alone and beside a stream of H2D copies out of the ring (the DMA reads the host memory the threads write).
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static void nt_copy(uint8_t *dst, const uint8_t *src, size_t n) {      // both 32-byte aligned, n a multiple of 128
    for (size_t o = 0; o < n; o += 128) {
        const __m256i a = _mm256_load_si256((const __m256i *)(src + o)), b = _mm256_load_si256((const __m256i *)(src + o + 32));
        const __m256i c = _mm256_load_si256((const __m256i *)(src + o + 64)), d = _mm256_load_si256((const __m256i *)(src + o + 96));
        _mm256_stream_si256((__m256i *)(dst + o), a); _mm256_stream_si256((__m256i *)(dst + o + 32), b);
        _mm256_stream_si256((__m256i *)(dst + o + 64), c); _mm256_stream_si256((__m256i *)(dst + o + 96), d);
    }
    _mm_sfence();
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: copy_probe <file> [threads] [GiB]\n"); return 2; }
    const int threads = argc > 2 ? atoi(argv[2]) : 14;
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror(argv[1]); return 1; }
    struct stat sb; fstat(fd, &sb);
    const size_t chunk = 512u << 10, ring = (size_t)1 << 30;
    const size_t total = std::min<size_t>((size_t)sb.st_size / chunk * chunk, (size_t)(argc > 3 ? atoi(argv[3]) : 8) << 30), n = total / chunk;
    CK(hipSetDevice(0));
    uint8_t *pin = nullptr, *dev = nullptr;
    CK(hipHostMalloc((void **)&pin, ring, hipHostMallocDefault)); memset(pin, 0, ring);
    CK(hipMalloc(&dev, ring));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint8_t *map = (uint8_t *)mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { perror("mmap"); return 1; }
    printf("file %.2f GB, moving %.2f GB in %zu KiB chunks, %d threads\n", sb.st_size / 1e9, total / 1e9, chunk >> 10, threads);
    auto run = [&](const char *name, int mode, bool with_dma) {
        std::atomic<size_t> next{0};
        std::atomic<bool> stop{false};
        std::thread dma;
        double dma_bytes = 0;
        if (with_dma) dma = std::thread([&] { while (!stop.load()) { CK(hipMemcpyAsync(dev, pin, ring / 4, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); dma_bytes += ring / 4; } });
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back([&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) break;
                uint8_t *dst = pin + (i * chunk) % ring;
                const size_t off = i * chunk;
                if (mode == 0) { size_t o = 0; while (o < chunk) { const ssize_t r = pread(fd, dst + o, chunk - o, (off_t)(off + o)); if (r <= 0) break; o += (size_t)r; } }
                else {
                    // modes: 1 map; 2 populate + memcpy + zap; 3 populate + nt + zap; 4 nt; 5 nt + zap; 6 memcpy + zap
                    if ((mode == 2 || mode == 3) && madvise(map + off, chunk, MADV_POPULATE_READ) != 0) { perror("MADV_POPULATE_READ"); exit(1); }
                    if (mode == 3 || mode == 4 || mode == 5) nt_copy(dst, map + off, chunk); else memcpy(dst, map + off, chunk);
                    if (mode == 2 || mode == 3 || mode == 5 || mode == 6) madvise(map + off, chunk, MADV_DONTNEED);
                }
            }
        });
        for (auto &t : th) t.join();
        const double dt = now() - t0;
        if (with_dma) { stop = true; dma.join(); }
        printf("%-12s %s: %.1f GB/s", name, with_dma ? "beside H2D copies" : "alone            ", total / dt / 1e9);
        if (with_dma) printf("   (H2D meanwhile %.1f GB/s)", dma_bytes / dt / 1e9);
        printf("\n");
        fflush(stdout);
    };
    const bool zap_all = true;
    for (int rep = 0; rep < 2; rep++)
        for (int dma = 0; dma < 2; dma++) {
            run("pread", 0, dma);
            run("map", 1, dma);
            if (zap_all) { const double t0 = now(); madvise(map, (size_t)sb.st_size, MADV_DONTNEED); printf("   (zap of the whole mapping afterwards: %.3f s)\n", now() - t0); }
            run("map+nt", 4, dma);
            if (zap_all) { const double t0 = now(); madvise(map, (size_t)sb.st_size, MADV_DONTNEED); printf("   (zap of the whole mapping afterwards: %.3f s)\n", now() - t0); }
            run("map+nt+zap", 5, dma);
            run("map+cpy+zap", 6, dma);
            if (rep == 0) { run("map+pop", 2, dma); run("map+pop+nt", 3, dma); }
        }
    return 0;
}
