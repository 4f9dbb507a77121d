// exit_probe: what does it cost to END a process that used the GPU?  tools/ubench/exit_probe <mode>
//   0 runtime initialised only   1 + 1 GiB page-locked host memory   2 + 32 GiB device memory (touched)   3 + 6 streams, 64 events
//   4 everything   5 everything, freed and destroyed explicitly before exit
// prints the wall clock at _exit so that the parent can split its wait into run time and reaping time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>
#include <vector>
static double wall() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double t0 = wall();
    (void)hipSetDevice(0); (void)hipFree(nullptr);
    const double t1 = wall();
    void *h = nullptr, *d = nullptr; std::vector<hipStream_t> st; std::vector<hipEvent_t> ev;
    if (mode == 1 || mode >= 4) { (void)hipHostMalloc(&h, (size_t)1 << 30, hipHostMallocDefault); }
    if (mode == 2 || mode >= 4) { (void)hipMalloc(&d, (size_t)32 << 30); (void)hipMemset(d, 1, (size_t)32 << 30); (void)hipDeviceSynchronize(); }
    if (mode == 3 || mode >= 4) { st.resize(6); for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); ev.resize(64); for (auto &e : ev) (void)hipEventCreate(&e); }
    if (mode == 5) { for (auto &e : ev) (void)hipEventDestroy(e); for (auto &s : st) (void)hipStreamDestroy(s); (void)hipFree(d); (void)hipHostFree(h); }
    printf("mode %d: init %.3fs, setup %.3fs, exit at %.6f\n", mode, t1 - t0, wall() - t1, wall());
    fflush(stdout);
    _exit(0);
}
