// Mock of the register-resident solve's Gauss-Seidel loop -- the REAL row steps of snk_pgs_v2.hpp (motor_step, quad_step,
// cone2_step) on register-resident rows -- at 1, 2, 3 and 4 waves per SIMD.  What it answers: how
// much more solve throughput a SIMD delivers with a third wave, i.e. what a <= 168-register variant of the 16-link
// kernel could gain.  Occupancy is set with the dynamic LDS size (one wave per workgroup, as in the step kernels), the
// code object is the same for every occupancy it is run at.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/ubench_solve.hip -o tools/ubench_solve
//   tools/ubench_solve [iterations per wave, default 2000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/snk.h"
#include "../bullet-envs_amd/csrc/snk_device.hpp"

struct MockLds {
    float MmS[16][4];
};

// NC contacts (a multiple of 4): NC / 2 normal slots, NC friction slots, 16 motor columns
template <int NC, int WAVES, int PIN>
__global__ __launch_bounds__(64, WAVES) void solve_mock(const float* __restrict__ in, float* __restrict__ out, int n_iter,
                                                        long long* __restrict__ ticks) {
    extern __shared__ float4 smem_raw[];
    MockLds& L = *reinterpret_cast<MockLds*>(smem_raw);
    constexpr int NS_N = NC / 2, NS_F = NC;
    const int lane = threadIdx.x;
    const int d = lane & 31;
    float RJ[NS_N + NS_F], RM[NS_N + NS_F], RMm[16];
#pragma unroll
    for (int s = 0; s < NS_N + NS_F; s++) {
        const float a = in[(2 * s) * 64 + lane], b = in[(2 * s + 1) * 64 + lane];
        RJ[s] = d < 22 ? 0.05f * a : (d == 22 ? a : 0.f);
        RM[s] = d < 22 ? 0.05f * b : (d == 24 ? 1.0f : 0.f);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) RMm[j] = d < 22 ? 0.02f * in[(400 + j) * 64 + lane] : 0.f;
    float TARGV = (d >= 6 && d < 22) ? 0.1f * in[420 * 64 + lane] : 0.f;
    float ACCV = 0.f;
    float dv = d == 22 ? 1.0f : (d == 31 ? -1.0f : 0.0f);
    float lsq = 0.f;
    const float E3163 = d == 31 ? 1.0f : 0.0f;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    __builtin_amdgcn_s_setprio(3);
    for (int it = 0; it < n_iter; it++) {
#pragma unroll
        for (int s = 0; s < NS_N + NS_F; s++) asm volatile("" : "+v"(RJ[s]), "+v"(RM[s]));
        float mres = (it & 1) ? snk::motors16<true, false, true>(L, RMm, dv, TARGV, ACCV, 0.f)
                              : snk::motors16<false, false, true>(L, RMm, dv, TARGV, ACCV, 0.f);
        lsq = fmaxf(lsq, mres);
#pragma unroll
        for (int q = 0; q < NC / 4; q++)
            snk::quad_step<true>(RJ[2 * q], RM[2 * q], RJ[2 * q + 1], RM[2 * q + 1], dv, E3163, snk::kLowMask, lsq);
#pragma unroll
        for (int c = 0; c < NC / 2; c++)
            snk::cone2_step<true, PIN>(RJ[NS_N + 2 * c], RM[NS_N + 2 * c], RJ[NS_N + 2 * c + 1], RM[NS_N + 2 * c + 1], RJ[c],
                                         dv, 1e-30f, E3163, snk::kLowMask, lsq);
    }
    __builtin_amdgcn_s_setprio(0);
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float acc = dv + lsq + ACCV;
#pragma unroll
    for (int s = 0; s < NS_N + NS_F; s++) acc += RJ[s];
    out[(size_t)blockIdx.x * 64 + lane] = acc;
    if (lane == 0) ticks[blockIdx.x] = (long long)(t1 - t0);
}

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

template <int NC, int WAVES, int PIN>
void run(const float* d_in, float* d_out, long long* d_ticks, int n_iter, int n_cu) {
    auto kern = solve_mock<NC, WAVES, PIN>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
    // VALU instructions of one iteration (hand count of the asm blocks): 16 motors x 4, quads 44 + 6, cone2 52 + 1
    const double valu_per_iter = 16 * 4 + (NC / 4) * 50.0 + (NC / 2) * 53.0;
    for (int w = 1; w <= WAVES; w++) {
        // 4 w one-wave workgroups per CU: an LDS size that lets 4 w of them in and not 4 w + 4 (one KB of slack under the
        // even share: at exactly 160 KB / 12 a CU held only 11 and the launch ran in two rounds)
        const size_t lds = (size_t)(160 * 1024 / (4 * w) / 1024 - (w == 3 ? 1 : 0)) * 1024;
        const int grid = n_cu * 4 * w;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, nullptr, d_in, d_out, 50, d_ticks);   // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, nullptr, d_in, d_out, n_iter, d_ticks);
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> t(grid);
        CHECK(hipMemcpy(t.data(), d_ticks, grid * sizeof(long long), hipMemcpyDeviceToHost));
        double mean = 0;
        long long mx = 0;
        for (long long x : t) { mean += (double)x; mx = x > mx ? x : mx; }
        mean /= grid;
        // s_memtime runs at 100 MHz; launch time is what counts
        const double it_per_s_simd = (double)w * n_iter / (ms * 1e-3);
        printf("contacts %2d  vgprs %3d  waves/SIMD %d  launch %8.3f ms  per-wave mean %9.0f max %9lld ticks  "
               "iterations/s/SIMD %9.0f  VALU/clk/SIMD %.3f (at 2.4 GHz)\n",
               NC, fa.numRegs, w, ms, mean, mx, it_per_s_simd, it_per_s_simd * valu_per_iter / 2.4e9);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const int n_iter = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s: %d CUs, %d iterations per wave\n", prop.name, n_cu, n_iter);
    std::vector<float> h(512 * 64);
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    float *d_in, *d_out;
    long long* d_ticks;
    CHECK(hipMalloc(&d_in, h.size() * sizeof(float)));
    CHECK(hipMemcpy(d_in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 16 * 64 * sizeof(float)));
    CHECK(hipMalloc(&d_ticks, (size_t)n_cu * 16 * sizeof(long long)));
    run<24, 4, 2>(d_in, d_out, d_ticks, n_iter, n_cu);      // fits 128 registers: up to four waves
    run<36, 3, 1>(d_in, d_out, d_ticks, n_iter, n_cu);      // the gait's mean, 168 registers: up to three
    run<40, 3, 1>(d_in, d_out, d_ticks, n_iter, n_cu);
    run<44, 3, 1>(d_in, d_out, d_ticks, n_iter, n_cu);
    run<48, 3, 1>(d_in, d_out, d_ticks, n_iter, n_cu);
    run<64, 2, 0>(d_in, d_out, d_ticks, n_iter, n_cu);     // the shipped kernel's slot count
    return 0;
}
