// Issue cost (SIMD cycles per wave64 instruction) of the instruction types used by the solve
// loop, measured with independent instructions and 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 4000
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define R8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float y = 1.0001f, z = 0.5f;
    float s0, s1, s2, s3, s4, s5, s6, s7;
    unsigned long long t0 = now();
    for (int i = 0; i < N_IT; i++) {
        if (MODE == 0)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "v"(z));
        else if (MODE == 1)
            asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_f32_dpp %6, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %7, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (MODE == 2)
            asm volatile("v_readlane_b32 %8, %0, 31\n\tv_readlane_b32 %9, %1, 31\n\tv_readlane_b32 %10, %2, 31\n\tv_readlane_b32 %11, %3, 31\n\t"
                         "v_readlane_b32 %12, %4, 31\n\tv_readlane_b32 %13, %5, 31\n\tv_readlane_b32 %14, %6, 31\n\tv_readlane_b32 %15, %7, 31"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                           "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7));
        else if (MODE == 3)
            asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (MODE == 4)
            asm volatile("v_rsq_f32 %0, %0\n\tv_rsq_f32 %1, %1\n\tv_rsq_f32 %2, %2\n\tv_rsq_f32 %3, %3\n\tv_rsq_f32 %4, %4\n\tv_rsq_f32 %5, %5\n\tv_rsq_f32 %6, %6\n\tv_rsq_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (MODE == 5)
            asm volatile("v_med3_f32 %0, %0, %8, %9\n\tv_med3_f32 %1, %1, %8, %9\n\tv_med3_f32 %2, %2, %8, %9\n\tv_med3_f32 %3, %3, %8, %9\n\t"
                         "v_med3_f32 %4, %4, %8, %9\n\tv_med3_f32 %5, %5, %8, %9\n\tv_med3_f32 %6, %6, %8, %9\n\tv_med3_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "v"(z));
        else if (MODE == 6)
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                         "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
        else if (MODE == 7)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4\n\t"
                         "v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4"
                         : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&y));
    }
    unsigned long long t1 = now();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name) {
    for (int blocks : {1024, 2048, 4096}) {
        float* d; unsigned long long* c;
        (void)hipMalloc(&d, blocks * 64 * 4); (void)hipMalloc(&c, blocks * 8);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, 0.3f);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, 0.3f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), c, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        int wps = blocks / 1024;
        printf("%-22s waves/SIMD=%d  %6.2f ticks per instr per wave  -> %5.2f SIMD ticks per instr\n", name, wps, avg / N_IT / 8, avg / N_IT / 8 / wps);
        (void)hipFree(d); (void)hipFree(c);
    }
}
int main() {
    run<0>("v_fma_f32"); run<6>("v_mul_f32"); run<5>("v_med3_f32"); run<1>("v_add_f32_dpp"); run<2>("v_readlane_b32");
    run<3>("v_permlane32_swap"); run<4>("v_rsq_f32"); run<7>("v_pk_fma_f32");
    return 0;
}
