// Micro-benchmark of the wave primitives used by the solve loop (cycles per dependent step).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_prims.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 2000
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float seed) {
    float x = seed + threadIdx.x * 1e-3f, y = 1.0001f, z = 0.5f;
    float s;
    unsigned long long t0 = now();
    for (int i = 0; i < N_IT; i++) {
        if (MODE == 0) {   // 8 dependent FMAs
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                         "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
        } else if (MODE == 1) {   // 8 independent-ish FMAs (two chains)
            float w = x + 1.f;
            asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\t"
                         "v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(w) : "v"(y), "v"(z));
            x += w * 1e-9f;
        } else if (MODE == 2) {   // 5-step DPP reduce chain (with the s_nop 1 hazards) x1
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(z));
        } else if (MODE == 3) {   // readlane -> VALU use -> readlane ... 4 round trips
            asm volatile("v_readlane_b32 %1, %0, 31\n\ts_nop 1\n\tv_fma_f32 %0, %0, %2, %1\n\t"
                         "v_readlane_b32 %1, %0, 31\n\ts_nop 1\n\tv_fma_f32 %0, %0, %2, %1\n\t"
                         "v_readlane_b32 %1, %0, 31\n\ts_nop 1\n\tv_fma_f32 %0, %0, %2, %1\n\t"
                         "v_readlane_b32 %1, %0, 31\n\ts_nop 1\n\tv_fma_f32 %0, %0, %2, %1" : "+v"(x), "=&s"(s) : "v"(z));
        } else if (MODE == 4) {   // permlane32_swap dependent x4
            float w = x;
            asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1\n\t"
                         "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1\n\t"
                         "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1\n\t"
                         "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "+v"(x), "+v"(w));
            x *= 0.24f;
        } else if (MODE == 5) {   // a whole single-row step as in the solve loop
            float t, xx, dI, s0, s1, RJ = x, RM = y, dv = z, E = 0.f;
            asm volatile(
                "v_mul_f32 %[t], %[RJ], %[dv]\n\tv_readlane_b32 %[s1], %[RJ], 23\n\ts_nop 0\n\t"
                "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                "v_add_f32_dpp %[t], %[t], %[t] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                "v_add_f32_dpp %[t], %[t], %[t] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
                "v_add_f32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                "v_readlane_b32 %[s0], %[t], 31\n\ts_nop 1\n\t"
                "v_med3_f32 %[x], -%[s0], %[LO], %[HI]\n\tv_subrev_f32 %[dI], %[s1], %[x]\n\ts_nop 0\n\t"
                "v_fmac_f32_dpp %[dv], %[RM], %[dI] quad_perm:[0,1,2,3] row_mask:0x3 bank_mask:0xf\n\t"
                "v_fmac_f32 %[RJ], %[E], %[dI]\n\ts_nop 1"
                : [t] "=&v"(t), [x] "=&v"(xx), [dI] "=&v"(dI), [s0] "=&s"(s0), [s1] "=&s"(s1), [RJ] "+v"(RJ), [dv] "+v"(dv)
                : [RM] "v"(RM), [LO] "v"(0.f), [HI] "v"(1e10f), [E] "v"(E));
            z = dv; x = RJ;
        }
    }
    unsigned long long t1 = now();
    out[blockIdx.x * 64 + threadIdx.x] = x + z;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int steps_per_iter, int blocks) {
    float* d; unsigned long long* c;
    hipMalloc(&d, blocks * 64 * 4); hipMalloc(&c, blocks * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, 0.3f);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, 0.3f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), c, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("%-34s blocks=%5d  %8.1f s_memtime ticks / iteration  = %6.2f per step (%d steps)\n", name, blocks, avg / N_IT, avg / N_IT / steps_per_iter, steps_per_iter);
    hipFree(d); hipFree(c);
}
int main() {
    for (int blocks : {256, 2048}) {   // 1 wave per CU ; 8 waves per CU (2 per SIMD)
        run<0>("8 dependent v_fma", 8, blocks);
        run<1>("8 v_fma in two chains", 8, blocks);
        run<2>("5-step DPP reduce + mul", 6, blocks);
        run<3>("4x readlane->fma round trip", 4, blocks);
        run<4>("4x mov+permlane32_swap+add", 4, blocks);
        run<5>("one single-row step (12 VALU)", 1, blocks);
    }
    return 0;
}
