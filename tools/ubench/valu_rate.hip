// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer VALU / SALU / LDS instructions the
// pileup kernels are made of, measured on the GPU at 1, 2, 4 and 8 waves per SIMD.  Decides which pipe bounds k_pileup_fast.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
constexpr int ITER = 2000;

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 ^ 0x55u, a3 = a0 + 7u, a4 = a0 * 5u, a5 = a0 + 11u, a6 = a0 ^ 9u, a7 = a0 + 13u;
    __shared__ unsigned lds[4096];
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1;
    __syncthreads();
    for (int it = 0; it < ITER; it++) {
        if (KIND == 0) {   // 8 independent chains of v_add_u32
            REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                               "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
        } else if (KIND == 1) {   // v_mad_u32_u24
            REP16(asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n"
                               "v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
        } else if (KIND == 2) {   // SDWA add with sign-extended word
            REP16(asm volatile("v_add_u32_sdwa %0, %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_add_u32_sdwa %1, %1, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_add_u32_sdwa %2, %2, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_add_u32_sdwa %3, %3, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_add_u32_sdwa %4, %4, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_add_u32_sdwa %5, %5, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_add_u32_sdwa %6, %6, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_add_u32_sdwa %7, %7, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
        } else if (KIND == 3) {   // min3 / max3 / add3
            REP16(asm volatile("v_min3_u32 %0, %0, %8, %1\n v_max3_u32 %1, %1, %8, %2\n v_add3_u32 %2, %2, %8, %3\n v_min3_u32 %3, %3, %8, %4\n"
                               "v_max3_u32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_min3_u32 %6, %6, %8, %7\n v_max3_u32 %7, %7, %8, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
        } else if (KIND == 4) {   // DPP adds (row_shr:1)
            REP16(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 5) {   // v_cmp + v_cndmask pairs
            REP16(asm volatile("v_cmp_ne_u32 vcc, 0, %0\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_ne_u32 vcc, 0, %2\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cmp_ne_u32 vcc, 0, %4\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_ne_u32 vcc, 0, %6\n v_cndmask_b32 %7, %7, %8, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed) : "vcc");)
        } else if (KIND == 6) {   // v_fma_f32 (the guide's 2-cycle reference)
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                               "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
        } else if (KIND == 7) {   // predicated block: v_cmp -> s_and_saveexec -> 2 VALU -> s_or exec   (the histogram pattern)
            REP16(asm volatile("v_cmp_ne_u32 vcc, 0, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n s_or_b64 exec, exec, s[20:21]\n"
                               "v_cmp_ne_u32 vcc, 0, %3\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n s_or_b64 exec, exec, s[20:21]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(a6), "v"(a7), "v"(seed) : "vcc", "s20", "s21");)
        } else if (KIND == 8) {   // ds_add_u32, random-ish addresses (no return)
            unsigned ad = ((a0 * 2654435761u) >> 20) & 0xffcu;
            REP16(asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:4096\n ds_add_u32 %0, %1 offset:8192\n ds_add_u32 %0, %1 offset:12288\n" :: "v"(ad), "v"(seed) : "memory");)
            a0 += 77u;
        } else if (KIND == 9) {   // s_bcnt1 + s_add chain on the scalar unit
            REP16(asm volatile("s_bcnt1_i32_b64 s20, vcc\n s_add_u32 s21, s21, s20\n s_bcnt1_i32_b64 s20, vcc\n s_add_u32 s21, s21, s20\n"
                               "s_bcnt1_i32_b64 s20, vcc\n s_add_u32 s21, s21, s20\n s_bcnt1_i32_b64 s20, vcc\n s_add_u32 s21, s21, s20\n" ::: "s20", "s21");)
        }
    }
    if (KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[threadIdx.x];
}

template <int KIND>
void run(const char *name, int per_iter, unsigned *d, int cus, double ghz_hint) {
    for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: a 256-thread block puts one wave on each SIMD
        const int grid = cus * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(256), 0, 0, d, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(256), 0, 0, d, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)wps * ITER * per_iter;
        printf("%-28s waves/SIMD %d : %.3f ms, %.2f ns per wave-instruction per SIMD = %.2f cycles at %.2f GHz\n", name, wps, ms,
               ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * ghz_hint, ghz_hint);
    }
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate / 1e6;
    printf("device %s, %d CUs, clock %.2f GHz\n", p.name, cus, ghz);
    unsigned *d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    run<6>("v_fma_f32", 128, d, cus, ghz);
    run<0>("v_add_u32", 128, d, cus, ghz);
    run<1>("v_mad_u32_u24", 128, d, cus, ghz);
    run<2>("v_add_u32_sdwa sext", 128, d, cus, ghz);
    run<3>("min3/max3/add3", 128, d, cus, ghz);
    run<4>("v_add_u32_dpp row_shr", 128, d, cus, ghz);
    run<5>("v_cmp + v_cndmask", 128, d, cus, ghz);
    run<7>("cmp/saveexec/2 valu/or (x5)", 160, d, cus, ghz);
    run<8>("ds_add_u32 random", 64, d, cus, ghz);
    run<9>("s_bcnt1 + s_add", 128, d, cus, ghz);
    return 0;
}
