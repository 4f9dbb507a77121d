// io_probe: how fast can a file in the page cache / on tmpfs reach HBM?  ("measure both")
//   hipcc -O2 -std=c++17 tools/ubench/io_probe.cpp -o tools/ubench/io_probe -lpthread
//   tools/ubench/io_probe <file> [threads]
// Paths measured over the same file:
//   A  threaded pread into page-locked slots (what covh_bam_gpu_ingest does), slots of 16 / 64 / 256 MiB, then H2D — pipelined
//   B  mmap + hipHostRegister of file pages piece by piece (registration threads ahead of the DMA), H2D straight from the page cache
//   C  hipMemcpy from the plain mmap (the runtime's own staging)
// plus the parts alone: pread rate, memcpy-from-mmap rate, H2D rate from page-locked memory, register / unregister cost per GB,
// hipMalloc / hipHostMalloc cost.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F>
static void par(int threads, size_t n, F fn) {
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back([&] { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) break; fn(i); } });
    for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: io_probe <file> [threads]\n"); return 2; }
    const char *path = argv[1];
    const int threads = argc > 2 ? atoi(argv[2]) : 16;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { perror(path); return 1; }
    struct stat sb; fstat(fd, &sb);
    const size_t size = (size_t)sb.st_size;
    printf("file %s: %.2f GB, %d threads\n", path, size / 1e9, threads);
    double t0 = now();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    printf("hip init %.3fs\n", now() - t0);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

    // ---- allocation costs
    for (size_t gb : {1, 4, 16}) {
        void *d = nullptr; t0 = now(); CK(hipMalloc(&d, gb << 30)); const double ta = now() - t0;
        t0 = now(); CK(hipMemsetAsync(d, 0, gb << 30, st)); CK(hipStreamSynchronize(st)); const double tm = now() - t0;
        t0 = now(); CK(hipFree(d)); printf("hipMalloc %zu GiB %.3fs, first memset %.3fs, hipFree %.3fs\n", gb, ta, tm, now() - t0);
    }
    uint8_t *dev = nullptr; const size_t dev_bytes = std::min<size_t>(size, (size_t)24 << 30);
    CK(hipMalloc(&dev, dev_bytes + (256u << 20)));
    CK(hipMemsetAsync(dev, 0, dev_bytes, st)); CK(hipStreamSynchronize(st));
    for (size_t mb : {64, 256}) {
        void *h = nullptr; t0 = now(); CK(hipHostMalloc(&h, mb << 20, hipHostMallocDefault)); const double ta = now() - t0;
        t0 = now(); memset(h, 1, mb << 20); const double tt = now() - t0;
        t0 = now(); CK(hipHostFree(h)); printf("hipHostMalloc %zu MiB %.4fs, first touch %.4fs, hipHostFree %.4fs\n", mb, ta, tt, now() - t0);
    }

    // ---- parts alone
    const size_t big = (size_t)1 << 30;
    uint8_t *pin = nullptr; CK(hipHostMalloc((void **)&pin, big, hipHostMallocDefault)); memset(pin, 0, big);
    for (int rep = 0; rep < 2; rep++) {
        t0 = now(); CK(hipMemcpyAsync(dev, pin, big, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
        printf("H2D 1 GiB from page-locked: %.1f GB/s\n", big / (now() - t0) / 1e9);
    }
    for (size_t chunk_mb : {1, 4, 16})
        for (int T : {8, threads, 2 * threads}) {
            const size_t chunk = chunk_mb << 20, total = std::min<size_t>(size, (size_t)4 << 30), n = total / chunk;
            t0 = now();
            par(T, n, [&](size_t i) { size_t o = 0; while (o < chunk) { const ssize_t r = pread(fd, pin + (i * chunk) % big + o, chunk - o, (off_t)(i * chunk + o)); if (r <= 0) break; o += (size_t)r; } });
            printf("pread -> page-locked: chunk %2zu MiB, %2d threads: %.1f GB/s\n", chunk_mb, T, n * chunk / (now() - t0) / 1e9);
        }
    t0 = now();
    uint8_t *map = (uint8_t *)mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { perror("mmap"); return 1; }
    printf("mmap %.4fs\n", now() - t0);
    {
        const size_t chunk = 4u << 20, total = std::min<size_t>(size, (size_t)4 << 30), n = total / chunk;
        for (int rep = 0; rep < 2; rep++) {
            t0 = now();
            par(threads, n, [&](size_t i) { memcpy(pin + (i * chunk) % big, map + i * chunk, chunk); });
            printf("memcpy mmap -> page-locked (%s), %d threads: %.1f GB/s\n", rep ? "mapped" : "first touch", threads, n * chunk / (now() - t0) / 1e9);
        }
    }
    // C: pageable copy straight from the mapping
    {
        const size_t total = std::min<size_t>(size - ((size_t)4 << 30 < size ? (size_t)4 << 30 : 0), (size_t)2 << 30), off = (size_t)4 << 30 < size ? (size_t)4 << 30 : 0;
        t0 = now(); CK(hipMemcpy(dev, map + off, total, hipMemcpyHostToDevice));
        printf("C  hipMemcpy from plain mmap (untouched pages): %.1f GB/s\n", total / (now() - t0) / 1e9);
    }
    // register cost
    for (size_t mb : {64, 256, 1024}) {
        const size_t n = mb << 20, off = ((size_t)6 << 30) + n < size ? (size_t)6 << 30 : 0;
        if (off + n > size) continue;
        for (unsigned flags : {0u, (unsigned)hipHostRegisterReadOnly}) {
            t0 = now(); hipError_t e = hipHostRegister(map + off, n, flags); const double tr = now() - t0;
            if (e != hipSuccess) { printf("hipHostRegister(flags %u) %zu MiB failed: %s\n", flags, mb, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            t0 = now(); CK(hipMemcpyAsync(dev, map + off, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); const double tc = now() - t0;
            t0 = now(); CK(hipHostUnregister(map + off)); const double tu = now() - t0;
            printf("hipHostRegister flags %u, %4zu MiB of the mmap: register %.4fs (%.1f GB/s), H2D %.1f GB/s, unregister %.4fs\n", flags, mb, tr, n / tr / 1e9, n / tc / 1e9, tu);
        }
    }

    // ---- A: pread into rotating page-locked slots, H2D behind it
    for (size_t slot_mb : {16, 64, 256}) {
        const int NS = 4; const size_t slot = slot_mb << 20, total = std::min(size, dev_bytes), n = (total + slot - 1) / slot;
        uint8_t *b[NS]; hipEvent_t ev[NS];
        for (int k = 0; k < NS; k++) { CK(hipHostMalloc((void **)&b[k], slot, hipHostMallocDefault)); memset(b[k], 0, slot); CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
        t0 = now();
        double t_read = 0, t_wait = 0;
        for (size_t k = 0; k < n; k++) {
            const int s = (int)(k % NS);
            double a = now();
            if (k >= NS) CK(hipEventSynchronize(ev[s]));
            t_wait += now() - a; a = now();
            const size_t off = k * slot, len = std::min(slot, total - off), chunk = 4u << 20, nc = (len + chunk - 1) / chunk;
            par(threads, nc, [&](size_t i) { size_t o = i * chunk; const size_t e = std::min(len, o + chunk); while (o < e) { const ssize_t r = pread(fd, b[s] + o, e - o, (off_t)(off + o)); if (r <= 0) break; o += (size_t)r; } });
            t_read += now() - a;
            CK(hipMemcpyAsync(dev + off, b[s], len, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(ev[s], st));
        }
        CK(hipStreamSynchronize(st));
        const double dt = now() - t0;
        printf("A  pread -> %d slots of %3zu MiB -> H2D: %.2f GB in %.3fs = %.1f GB/s (read %.3fs, slot waits %.3fs)\n", NS, slot_mb, total / 1e9, dt, total / dt / 1e9, t_read, t_wait);
        for (int k = 0; k < NS; k++) { CK(hipHostFree(b[k])); CK(hipEventDestroy(ev[k])); }
    }
    // A2: two reader groups (reads of slot k+1 overlap the tail of slot k's reads): per-chunk pipelining with a global queue
    {
        const size_t chunk = 8u << 20, total = std::min(size, dev_bytes), n = (total + chunk - 1) / chunk;
        const int NB = 32;                                          // 32 x 8 MiB in flight
        uint8_t *ring = nullptr; CK(hipHostMalloc((void **)&ring, (size_t)NB * chunk, hipHostMallocDefault)); memset(ring, 0, (size_t)NB * chunk);
        std::vector<hipEvent_t> ev(NB); for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        std::mutex mu; std::condition_variable cv; std::vector<char> done(n, 0); size_t issued = 0; std::atomic<size_t> next{0};
        t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back([&] {
            for (;;) {
                const size_t i = next.fetch_add(1); if (i >= n) break;
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return i < issued + NB; }); }      // the ring slot's previous upload has been issued...
                if (i >= (size_t)NB) (void)hipEventSynchronize(ev[i % NB]);                                      // ...and has finished
                const size_t off = i * chunk, len = std::min(chunk, total - off); size_t o = 0;
                while (o < len) { const ssize_t r = pread(fd, ring + (i % NB) * chunk + o, len - o, (off_t)(off + o)); if (r <= 0) break; o += (size_t)r; }
                { std::lock_guard<std::mutex> lk(mu); done[i] = 1; } cv.notify_all();
            }
        });
        for (size_t i = 0; i < n; i++) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done[i] != 0; }); }
            const size_t off = i * chunk, len = std::min(chunk, total - off);
            CK(hipMemcpyAsync(dev + off, ring + (i % NB) * chunk, len, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(ev[i % NB], st));
            { std::lock_guard<std::mutex> lk(mu); issued = i + 1; } cv.notify_all();
        }
        CK(hipStreamSynchronize(st));
        for (auto &t : th) t.join();
        const double dt = now() - t0;
        printf("A2 pread -> ring of %d x 8 MiB, every thread its own chunk, in-order H2D: %.2f GB in %.3fs = %.1f GB/s\n", NB, total / 1e9, dt, total / dt / 1e9);
        CK(hipHostFree(ring));
    }
    // ---- B: register file pages ahead of the DMA
    for (size_t piece_mb : {64, 256}) {
        for (int RT : {1, 4}) {
            const size_t piece = piece_mb << 20, total = std::min(size, dev_bytes) / piece * piece, n = total / piece;
            if (!n) continue;
            const size_t AHEAD = 6;
            std::mutex mu; std::condition_variable cv; std::vector<char> reg(n, 0); size_t copied_upto = 0; std::atomic<size_t> next{0}; bool failed = false;
            t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < RT; t++) th.emplace_back([&] {
                (void)hipSetDevice(0);
                for (;;) {
                    const size_t i = next.fetch_add(1); if (i >= n) break;
                    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return i < copied_upto + AHEAD; }); }
                    const hipError_t e = hipHostRegister(map + i * piece, piece, hipHostRegisterDefault);
                    { std::lock_guard<std::mutex> lk(mu); reg[i] = e == hipSuccess ? 1 : 2; if (e != hipSuccess) failed = true; } cv.notify_all();
                }
            });
            std::vector<hipEvent_t> ev(n);
            double t_unreg = 0;
            for (size_t i = 0; i < n && !failed; i++) {
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return reg[i] != 0; }); }
                if (reg[i] != 1) break;
                CK(hipMemcpyAsync(dev + i * piece, map + i * piece, piece, hipMemcpyHostToDevice, st));
                CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); CK(hipEventRecord(ev[i], st));
                if (i >= 2) { CK(hipEventSynchronize(ev[i - 2])); const double a = now(); CK(hipHostUnregister(map + (i - 2) * piece)); t_unreg += now() - a; }
                { std::lock_guard<std::mutex> lk(mu); copied_upto = i + 1; } cv.notify_all();
            }
            CK(hipStreamSynchronize(st));
            { std::lock_guard<std::mutex> lk(mu); copied_upto = n + AHEAD; } cv.notify_all();
            for (auto &t : th) t.join();
            const double dt = now() - t0;
            if (failed) { printf("B  register-ahead: hipHostRegister failed\n"); (void)hipGetLastError(); break; }
            for (size_t i = n >= 2 ? n - 2 : 0; i < n; i++) (void)hipHostUnregister(map + i * piece);
            printf("B  mmap + hipHostRegister ahead (%d registering threads, pieces of %3zu MiB) -> H2D from the page cache: %.2f GB in %.3fs = %.1f GB/s (unregister %.3fs on the DMA thread)\n",
                   RT, piece_mb, total / 1e9, dt, total / dt / 1e9, t_unreg);
        }
    }
    printf("done\n");
    fflush(stdout);
    _exit(0);
}
