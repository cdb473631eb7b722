// micro-benchmark: issue rate of integer VALU ops on gfx950 at a given occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t s = seed | 1;
    for (int i = 0; i < iters; ++i) {
#define OP8(INS) asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        if (KIND == 0) { OP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8") }
        if (KIND == 1) { OP8("v_alignbit_b32 %0, %0, %8, 7\n v_alignbit_b32 %1, %1, %8, 7\n v_alignbit_b32 %2, %2, %8, 7\n v_alignbit_b32 %3, %3, %8, 7\n v_alignbit_b32 %4, %4, %8, 7\n v_alignbit_b32 %5, %5, %8, 7\n v_alignbit_b32 %6, %6, %8, 7\n v_alignbit_b32 %7, %7, %8, 7") }
        if (KIND == 2) { OP8("v_med3_i32 %0, %0, %8, 64\n v_med3_i32 %1, %1, %8, 64\n v_med3_i32 %2, %2, %8, 64\n v_med3_i32 %3, %3, %8, 64\n v_med3_i32 %4, %4, %8, 64\n v_med3_i32 %5, %5, %8, 64\n v_med3_i32 %6, %6, %8, 64\n v_med3_i32 %7, %7, %8, 64") }
        if (KIND == 3) { OP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8") }
        if (KIND == 4) { OP8("v_max_i32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max_i32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max_i32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max_i32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max_i32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max_i32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max_i32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_max_i32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2") }
        if (KIND == 5) { OP8("v_bfe_i32 %0, %0, %8, 2\n v_bfe_i32 %1, %1, %8, 2\n v_bfe_i32 %2, %2, %8, 2\n v_bfe_i32 %3, %3, %8, 2\n v_bfe_i32 %4, %4, %8, 2\n v_bfe_i32 %5, %5, %8, 2\n v_bfe_i32 %6, %6, %8, 2\n v_bfe_i32 %7, %7, %8, 2") }
        if (KIND == 6) { OP8("v_mad_i32_i24 %0, %0, %8, %8\n v_mad_i32_i24 %1, %1, %8, %8\n v_mad_i32_i24 %2, %2, %8, %8\n v_mad_i32_i24 %3, %3, %8, %8\n v_mad_i32_i24 %4, %4, %8, %8\n v_mad_i32_i24 %5, %5, %8, %8\n v_mad_i32_i24 %6, %6, %8, %8\n v_mad_i32_i24 %7, %7, %8, %8") }
        if (KIND == 7) { OP8("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8") }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int KIND> void run(const char *name, uint32_t *d, int wavesPerSimd)
{
    // 256 CUs x 4 SIMDs; block = 256 threads = 4 waves (one per SIMD); blocks per CU = wavesPerSimd
    const int blocks = 256 * wavesPerSimd, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instrPerSimd = (double)iters * 8 * wavesPerSimd;          // wave-instructions per SIMD
    printf("%-16s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wavesPerSimd, ms,
           ms * 1e-3 * 2.4e9 / instrPerSimd);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_u32", d, w); run<1>("v_alignbit", d, w); run<2>("v_med3_i32", d, w); run<3>("v_fma_f32", d, w);
        run<4>("v_max_i32_sdwa", d, w); run<5>("v_bfe_i32", d, w); run<6>("v_mad_i32_i24", d, w); run<7>("v_and_b32", d, w);
    }
    return 0;
}
