// What does a scattered store cost?  k_inflate_wave's pass 3 is bound by its stores: 64 lanes, each writing front to back inside its own
// ~1 KiB of a 64 KiB block.  This probe issues exactly that address pattern with one store shape at a time:
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/store_probe tools/ubench/store_probe.hip && timeout 60 tools/ubench/store_probe
// Output: ns per wave-store instruction per CU and payload GB/s for each (width, misalignment, advance) — advance < width = stores that
// overlap their predecessor (the padded stores of Sink<2> / Sink<3>).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int WIDTH>
__global__ __launch_bounds__(64) void k_store(uint8_t *__restrict__ out, uint32_t mis, uint32_t advance, uint32_t n_stores) {
    uint8_t *p = out + (size_t)blockIdx.x * 65536u + (threadIdx.x & 63u) * 1024u + mis;
    uint64_t v = 0x0102030405060708ull * (threadIdx.x + 1u);
    for (uint32_t k = 0; k < n_stores; k++) {
        if (WIDTH == 1) *p = (uint8_t)v;
        else if (WIDTH == 2) { const uint16_t x = (uint16_t)v; __builtin_memcpy(p, &x, 2); }
        else if (WIDTH == 4) { const uint32_t x = (uint32_t)v; __builtin_memcpy(p, &x, 4); }
        else if (WIDTH == 8) __builtin_memcpy(p, &v, 8);
        else { const uint64_t x[2] = {v, ~v}; __builtin_memcpy(p, x, 16); }
        p += advance;
        v = v * 6364136223846793005ull + 1442695040888963407ull;      // a few VALU ops between the stores, like a decode step
    }
}

// the same address pattern read instead of written (k_bam_extract and k_lz_resolve read records and matches at any byte offset)
template <int WIDTH>
__global__ __launch_bounds__(64) void k_load(const uint8_t *__restrict__ in, uint32_t mis, uint32_t advance, uint32_t n_loads, uint64_t *__restrict__ sink) {
    const uint8_t *p = in + (size_t)blockIdx.x * 65536u + (threadIdx.x & 63u) * 1024u + mis;
    uint64_t acc = 0;
    for (uint32_t k = 0; k < n_loads; k++) {
        if (WIDTH == 1) acc += *p;
        else if (WIDTH == 2) { uint16_t x; __builtin_memcpy(&x, p, 2); acc += x; }
        else if (WIDTH == 4) { uint32_t x; __builtin_memcpy(&x, p, 4); acc += x; }
        else if (WIDTH == 8) { uint64_t x; __builtin_memcpy(&x, p, 8); acc += x; }
        else { uint64_t x[2]; __builtin_memcpy(x, p, 16); acc += x[0] ^ x[1]; }
        p += advance;
    }
    if (acc == 0x123456789abcdefull) sink[0] = acc;      // never true for this data: keeps the loads alive
}

// Is the cost of a scattered store instruction per INSTRUCTION or per active LANE?  (Round 5: pass 3 of k_inflate_wave stores in ~390 of its
// lock-steps per block with about a third of the lanes active each time; queueing the lanes' segments and flushing them together would cut
// the instructions, not the lane-stores.)  Every iteration a different subset of `active` lanes stores 8 bytes into its own KiB.
__global__ __launch_bounds__(64) void k_store_part(uint8_t *__restrict__ out, uint32_t active, uint32_t n_iter) {
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *p = out + (size_t)blockIdx.x * 65536u + lane * 1024u;
    uint64_t v = 0x0102030405060708ull * (threadIdx.x + 1u);
    for (uint32_t k = 0; k < n_iter; k++) {
        if (((lane * 7u + k * 13u) & 63u) < active) { __builtin_memcpy(p, &v, 8); p += 8; if (p >= out + (size_t)blockIdx.x * 65536u + lane * 1024u + 1016u) p -= 1016u; }
        v = v * 6364136223846793005ull + 1442695040888963407ull;
    }
}

int main() {
    const uint32_t blocks = 81920;                                   // one round of the ingest
    uint8_t *out;
    CHK(hipMalloc(&out, (size_t)blocks * 65536u + 4096));
    CHK(hipMemset(out, 0, (size_t)blocks * 65536u + 4096));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int cus = 0;
    CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    struct Case { int width; uint32_t mis, advance; };
    const Case cases[] = {{1, 0, 1}, {2, 0, 2}, {2, 1, 2}, {4, 0, 4}, {4, 1, 4}, {4, 2, 4}, {8, 0, 8}, {8, 4, 8}, {8, 1, 8}, {8, 2, 8}, {16, 0, 16}, {16, 4, 16}, {16, 1, 16},
                          {8, 0, 5}, {8, 1, 5}, {8, 0, 3}, {4, 0, 3}, {4, 0, 2}};
    printf("%u blocks x 64 lanes, %d CUs; each lane covers 960 bytes of its own KiB\n", blocks, cus);
    for (const Case &c : cases) {
        const uint32_t n = 960u / c.advance;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0, 0));
            switch (c.width) {
                case 1: hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n); break;
                case 2: hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n); break;
                case 4: hipLaunchKernelGGL(k_store<4>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n); break;
                case 8: hipLaunchKernelGGL(k_store<8>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n); break;
                default: hipLaunchKernelGGL(k_store<16>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n); break;
            }
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wave_stores = (double)blocks * n;
        printf("width %2d  misaligned by %u  advance %2u: %8.3f ms  %7.1f ns per wave-store per CU  %7.1f GB/s payload  (%u stores per lane)\n", c.width, c.mis, c.advance,
               best, best * 1e6 / (wave_stores / cus), (double)blocks * 64 * n * c.advance / best / 1e6, n);
    }
    printf("partial stores: 8 bytes, 240 iterations, a different subset of the lanes active in each\n");
    for (uint32_t active : {4u, 8u, 16u, 21u, 32u, 48u, 64u}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_store_part, dim3(blocks), dim3(64), 0, 0, out, active, 240u);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("active lanes %2u of 64: %8.3f ms  %7.1f ns per wave-store instruction per CU  %6.2f ns per lane-store per CU\n", active, best,
               best * 1e6 / ((double)blocks * 240 / cus), best * 1e6 / ((double)blocks * 240 * active / cus));
    }
    uint64_t *sink;
    CHK(hipMalloc(&sink, 8));
    printf("loads, same address pattern:\n");
    const Case lcases[] = {{1, 0, 1}, {2, 0, 2}, {2, 1, 2}, {4, 0, 4}, {4, 1, 4}, {4, 2, 4}, {8, 0, 8}, {8, 4, 8}, {8, 1, 8}, {16, 0, 16}, {16, 4, 16}, {16, 1, 16}};
    for (const Case &c : lcases) {
        const uint32_t n = 960u / c.advance;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0, 0));
            switch (c.width) {
                case 1: hipLaunchKernelGGL(k_load<1>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n, sink); break;
                case 2: hipLaunchKernelGGL(k_load<2>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n, sink); break;
                case 4: hipLaunchKernelGGL(k_load<4>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n, sink); break;
                case 8: hipLaunchKernelGGL(k_load<8>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n, sink); break;
                default: hipLaunchKernelGGL(k_load<16>, dim3(blocks), dim3(64), 0, 0, out, c.mis, c.advance, n, sink); break;
            }
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wave_loads = (double)blocks * n;
        printf("width %2d  misaligned by %u  advance %2u: %8.3f ms  %7.1f ns per wave-load per CU  %7.1f GB/s  (%u loads per lane)\n", c.width, c.mis, c.advance, best,
               best * 1e6 / (wave_loads / cus), (double)blocks * 64 * n * c.advance / best / 1e6, n);
    }
    CHK(hipFree(sink));
    CHK(hipFree(out));
    return 0;
}
