// Dependent-chain latency (ticks per step, one wave per SIMD) of the instruction patterns on
// the critical path of the solve loop.  hipcc --offload-arch=gfx950 -O3 tools/ubench_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 2000
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define R8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float seed) {
    float a = seed + threadIdx.x * 1e-3f, y = 1.0001f, z = 1e-6f, x = 0.f, c2 = 0.f;
    float s = 0.f;
    unsigned long long m = 0x00000000013FFFFFull;
    unsigned long long t0 = now();
    for (int i = 0; i < N_IT; i++) {
        if (MODE == 0) asm volatile(R8("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(y), "v"(z));
        else if (MODE == 1) asm volatile(R8("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t") : "+v"(a));
        else if (MODE == 2) asm volatile(R8("v_readlane_b32 %1, %0, 31\n\ts_nop 1\n\tv_fmac_f32 %0, %1, %2\n\ts_nop 0\n\t") : "+v"(a), "=&s"(s) : "v"(z));
        else if (MODE == 3) asm volatile(R8("s_mov_b64 exec, %2\n\tv_mul_f32 %1, %0, %3\n\ts_mov_b64 exec, -1\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(a), "+v"(x) : "s"(m), "v"(z));
        else if (MODE == 4) asm volatile(R8("v_mov_b32 %1, %0\n\ts_nop 0\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(a), "+v"(c2));
        else if (MODE == 5) asm volatile(R8("v_med3_f32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(z), "v"(y));
        else if (MODE == 6) asm volatile(R8("v_rsq_f32 %0, %0\n\ts_nop 1\n\t") : "+v"(a));
        else if (MODE == 7) asm volatile(R8("v_mul_f32 %1, %0, %2\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(a), "+v"(x) : "v"(z));
        else if (MODE == 8) asm volatile(R8("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %1, %2\n\t") : "+v"(a), "+v"(x) : "v"(z));
        else if (MODE == 9) asm volatile(R8("v_readlane_b32 %1, %0, 31\n\tv_mov_b32 %3, %2\n\tv_mov_b32 %3, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_mov_b32 %3, %2\n\t") : "+v"(a), "=&s"(s), "+v"(z), "+v"(x));
        else if (MODE == 10) asm volatile(R8("v_add_f32 %0, %0, %1\n\t") : "+v"(a) : "v"(z));
        else if (MODE == 11) asm volatile(R8("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2\n\t") : "+v"(a) : "v"(y), "v"(z), "v"(x));
    }
    unsigned long long t1 = now();
    out[blockIdx.x * 64 + threadIdx.x] = a + x + c2 + s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int per) {
    for (int blocks : {1024, 2048}) {
        float* d; unsigned long long* c;
        (void)hipMalloc(&d, blocks * 64 * 4); (void)hipMalloc(&c, blocks * 8);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, 0.3f);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, 0.3f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), c, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        printf("%-44s waves/SIMD=%d  %7.2f ticks per step (%d instr/step)\n", name, blocks / 1024, avg / N_IT / 8, per);
        (void)hipFree(d); (void)hipFree(c);
    }
}
int main() {
    // wall-clock calibration of the tick
    {
        float* d; unsigned long long* c;
        (void)hipMalloc(&d, 1024 * 64 * 4); (void)hipMalloc(&c, 1024 * 8);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, d, c, 0.3f);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, d, c, 0.3f);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("tick calibration: %llu ticks in %.3f ms kernel -> >= %.1f MHz\n", h, ms, h / ms / 1e3);
    }
    run<0>("v_fma dependent", 1);
    run<10>("v_add dependent", 1);
    run<11>("2 independent v_fma chains (per pair)", 2);
    run<5>("v_med3 dependent", 1);
    run<1>("v_add_dpp + s_nop 1 dependent", 1);
    run<8>("v_add_dpp + 2 independent v_mov", 3);
    run<2>("readlane, s_nop1, fmac(sgpr), s_nop0", 2);
    run<9>("readlane, 2 mov, fmac(sgpr), mov", 5);
    run<3>("exec=m, v_mul, exec=-1, v_add", 2);
    run<7>("v_mul, v_add (no exec)", 2);
    run<4>("mov, swap32, add (+nops)", 3);
    run<6>("v_rsq + s_nop 1 dependent", 1);
    return 0;
}
