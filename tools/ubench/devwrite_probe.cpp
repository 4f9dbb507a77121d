// devwrite_probe: can the CPU write the file straight into device memory through the PCIe BAR (no staging copy, no DMA)?
//   tools/ubench/devwrite_probe <file> [threads]
// hipMalloc memory (coarse grained) and hipExtMallocWithFlags(fine grained) are tried; a SIGSEGV / SIGBUS on first touch means the
// allocation is not CPU-visible here.  Prints GB/s of threaded pread() straight into the device allocation, then verifies a sample
// of the bytes by copying them back.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static sigjmp_buf jb;
static void on_fault(int) { siglongjmp(jb, 1); }
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const int threads = argc > 2 ? atoi(argv[2]) : 14;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb; fstat(fd, &sb);
    const size_t total = std::min<size_t>((size_t)sb.st_size, (size_t)8 << 30);
    (void)hipSetDevice(0); (void)hipFree(nullptr);
    signal(SIGSEGV, on_fault); signal(SIGBUS, on_fault);
    for (int kind = 0; kind < 2; kind++) {
        uint8_t *d = nullptr;
        hipError_t e = kind == 0 ? hipMalloc((void **)&d, total) : hipExtMallocWithFlags((void **)&d, total, hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("kind %d: allocation failed: %s\n", kind, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        if (sigsetjmp(jb, 1)) { printf("kind %d (%s): not CPU-writable (fault on first touch)\n", kind, kind ? "fine grained" : "hipMalloc"); (void)hipFree(d); continue; }
        volatile uint8_t *t = d; t[0] = 1; t[4096] = 2;      // faults if the BAR does not map it
        for (size_t chunk_kb : {256, 1024, 4096}) {
            const size_t chunk = chunk_kb << 10, n = total / chunk;
            std::atomic<size_t> next{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (int k = 0; k < threads; k++) th.emplace_back([&] { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) break; size_t o = 0; while (o < chunk) { const ssize_t r = pread(fd, d + i * chunk + o, chunk - o, (off_t)(i * chunk + o)); if (r <= 0) break; o += (size_t)r; } } });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            printf("kind %d (%s): pread -> device memory, %d threads, chunk %4zu KiB: %.2f GB in %.3fs = %.1f GB/s\n", kind, kind ? "fine grained" : "hipMalloc", threads, chunk_kb, n * chunk / 1e9, dt, n * chunk / dt / 1e9);
        }
        std::vector<uint8_t> back(1 << 20), want(1 << 20);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(back.data(), d + (total / 2 & ~(size_t)4095), back.size(), hipMemcpyDeviceToHost);
        (void)!pread(fd, want.data(), want.size(), (off_t)(total / 2 & ~(size_t)4095));
        printf("kind %d: bytes read back %s\n", kind, memcmp(back.data(), want.data(), back.size()) ? "DIFFER" : "match");
        (void)hipFree(d);
    }
    fflush(stdout); _exit(0);
}
