// the one form of the 16-link solve not yet priced -- lane = ROW for the contact rows, sequential
// Gauss-Seidel inside blocks of 16 rows through a block-diagonal Delassus tile.  A MOCK with the instruction mix and the
// dependency chains of that design (numbers bounded, not physics), run like tools/ubench_solve.hip at 1 .. 3 waves per
// SIMD, next to the shipped row steps' rate measured by that tool on the same box.
//
// The design (per wave, NC contacts, R = 3 NC rows: the NC normals first, then the NC friction pairs (A, B) in adjacent
// lanes -- Bullet's row order; row r lives in lane r % 64 of register set r / 64):
//   JS[set][22]   J / den, lane = row, one register per velocity component
//   AT[set][16]   the Delassus tile of each block of 16 rows: register i, lane (block b, row j) = (J_j M^-1 J_i^T) / den_j
//   RHS, ACC      per-row scalars, lane = row
//   RM[R / 2]     M^-1 J^T, lane = velocity component, rows 2p / 2p + 1 in the two halves of a register (today's layout)
//   dv            delta-v, lane = velocity component, both halves alike (today's layout)
// Per block of 16 rows:
//   prologue   22 v_readlane (dv -> SGPRs), 22 v_fmac with an SGPR operand: s = J.dv / den for the 64 rows of the set
//              (16 of them used), one v_sub for the rhs
//   per normal row i   x = max(a - s, 0); d = x - a; dI = v_readlane(d, lane i); s += AT[i] dI (SGPR operand);
//                      a = x in lane i (v_cndmask on a constant mask); per two rows: dvec = {dI_even | dI_odd} by halves,
//                      dvp += dvec * RM[pair]
//   per friction pair  x = a - s; r2 = x^2 + partner's (one DPP add); scale = min(lim * rsq(r2), 1); x *= scale; d = x - a;
//                      two v_readlane, two v_fmac on s, one v_cndmask on a, the same dv update
//   epilogue   the halves of dvp exchanged (v_permlane32_swap) and added into dv
// Friction limits lim = mu * a_normal reach the friction rows' lanes through one ds_bpermute per register set and
// iteration.  The motor rows are today's (snk::motors16).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/ubench_block.hip -o tools/ubench_block
//   tools/ubench_block [iterations per wave, default 2000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/snk.h"
#include "../bullet-envs_amd/csrc/snk_device.hpp"

struct MockLds {
    float MmS[16][4];
};

__device__ __forceinline__ float rdl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <int NC, int WAVES>
__global__ __launch_bounds__(64, WAVES) void block_mock(const float* __restrict__ in, float* __restrict__ out, int n_iter,
                                                        long long* __restrict__ ticks) {
    extern __shared__ float4 smem_raw[];
    MockLds& L = *reinterpret_cast<MockLds*>(smem_raw);
    constexpr int R = 3 * NC, G = (R + 63) / 64, NP = (R + 1) / 2;
    static_assert(NC % 2 == 0 && NC <= 64, "the normals fit one register set; friction pairs start on an even lane");
    const int lane = threadIdx.x;
    const int d = lane & 31;
    float JS[G][22], AT[G][16], RHS[G], ACC[G], LIM[G], RM[NP], RMm[16];
    int k = 0;
    auto nxt = [&]() { const float v = in[((k++) % 480) * 64 + lane]; return v; };
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int c = 0; c < 22; c++) JS[g][c] = 0.05f * nxt();
#pragma unroll
        for (int i = 0; i < 16; i++) AT[g][i] = 0.01f * nxt();
        RHS[g] = 0.1f * nxt();
        ACC[g] = 0.f;
        LIM[g] = 0.f;
    }
#pragma unroll
    for (int p = 0; p < NP; p++) RM[p] = d < 22 ? 0.05f * nxt() : 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) RMm[j] = d < 22 ? 0.02f * nxt() : 0.f;
    float TARGV = (d >= 6 && d < 22) ? 0.1f * nxt() : 0.f;
    float ACCV = 0.f;
    float dv = d == 22 ? 1.0f : (d == 31 ? -1.0f : 0.0f);
    float lsq = 0.f;
    const int nidx = 4 * (lane >= NC ? ((lane - NC) >> 1) : lane);          // set 0: friction row in lane l belongs to contact (l - NC) / 2
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    __builtin_amdgcn_s_setprio(3);
    for (int it = 0; it < n_iter; it++) {
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int c = 0; c < 22; c++) asm volatile("" : "+v"(JS[g][c]));
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("" : "+v"(AT[g][i]));
        }
#pragma unroll
        for (int p = 0; p < NP; p++) asm volatile("" : "+v"(RM[p]));
        float mres = (it & 1) ? snk::motors16<true, false, true>(L, RMm, dv, TARGV, ACCV, 0.f)
                              : snk::motors16<false, false, true>(L, RMm, dv, TARGV, ACCV, 0.f);
        lsq = fmaxf(lsq, mres);
        float s = 0.f, dvp = 0.f, dI_even = 0.f;
        auto prologue = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            s = -RHS[g];
#pragma unroll
            for (int c = 0; c < 22; c++) s = fmaf(JS[g][c], rdl(dv, c), s);
        };
        auto epilogue = [&]() {
            float x = dvp, c2 = dvp;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(c2));
            dv += x;
            dv += c2;
            dvp = 0.f;
        };
        auto row = [&](auto r_c) {
            constexpr int r = decltype(r_c)::value;
            constexpr int g = r / 64, l = r % 64, i = l % 16;
            if constexpr (i == 0) prologue(std::integral_constant<int, g>{});
            if constexpr (r == NC) {
                // the friction limits: every friction row's lane takes its contact's normal impulse (set 0 holds the normals)
#pragma unroll
                for (int gg = 0; gg < G; gg++) {
                    const int idx = gg == 0 ? nidx : 4 * ((64 * gg + lane - NC) >> 1);
                    LIM[gg] = 0.8f * __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(ACC[0])));
                }
            }
            if constexpr (r < NC) {
                const float a = ACC[g];
                const float x = fmaxf(a - s, 0.f);
                const float dd = x - a;
                const float dI = rdl(dd, l);
                s = fmaf(AT[g][i], dI, s);
                ACC[g] = (lane == l) ? x : a;
                lsq = fmaxf(lsq, fabsf(dI));
                if constexpr ((r & 1) == 0) dI_even = dI;
                else {
                    const float t = lane < 32 ? dI_even : dI;
                    dvp = fmaf(t, RM[r / 2], dvp);
                }
            } else if constexpr (((r - NC) & 1) == 0) {
                const float a = ACC[g];
                float x = a - s;
                float q = fmaf(x, x, 1e-30f);
                q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), 0xB1, 0xf, 0xf, true));      // the partner's
                const float sc = fminf(LIM[g] * __builtin_amdgcn_rsqf(q), 1.0f);
                x *= sc;
                const float dd = x - a;
                const float dA = rdl(dd, l), dB = rdl(dd, l + 1);
                s = fmaf(AT[g][i], dA, s);
                s = fmaf(AT[g][i + 1], dB, s);
                ACC[g] = (lane == l || lane == l + 1) ? x : a;
                lsq = fmaxf(lsq, fmaxf(fabsf(dA), fabsf(dB)));
                const float t = lane < 32 ? dA : dB;
                dvp = fmaf(t, RM[r / 2], dvp);
            }
            if constexpr (i == 15 || r == R - 1) epilogue();
        };
        [&]<int... RR>(std::integer_sequence<int, RR...>) { (row(std::integral_constant<int, RR>{}), ...); }
        (std::make_integer_sequence<int, R>{});
    }
    __builtin_amdgcn_s_setprio(0);
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float acc = dv + lsq + ACCV;
#pragma unroll
    for (int g = 0; g < G; g++) acc += ACC[g];
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

template <int NC, int WAVES>
void run(const float* d_in, float* d_out, long long* d_ticks, int n_iter, int n_cu) {
    auto kern = block_mock<NC, WAVES>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
    const int R = 3 * NC;
    for (int w = 1; w <= WAVES; w++) {
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
        const double it_per_s_simd = (double)w * n_iter / (ms * 1e-3);
        printf("block form: contacts %2d (%3d rows + 16 motors)  vgprs %3d  scratch %4zu B  waves/SIMD %d  launch %8.3f ms  "
               "iterations/s/SIMD %9.0f  clocks per row and SIMD %.1f (at 2.4 GHz)\n",
               NC, R, fa.numRegs, (size_t)fa.localSizeBytes, w, ms, it_per_s_simd, 2.4e9 / it_per_s_simd / (R + 16));
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
    run<24, 3>(d_in, d_out, d_ticks, n_iter, n_cu);
    run<36, 3>(d_in, d_out, d_ticks, n_iter, n_cu);      // the gait's mean
    run<40, 3>(d_in, d_out, d_ticks, n_iter, n_cu);      // its 90th percentile
    run<36, 2>(d_in, d_out, d_ticks, n_iter, n_cu);      // the same rows with 256 registers to spend
    run<64, 2>(d_in, d_out, d_ticks, n_iter, n_cu);      // the shipped kernel's slot count
    return 0;
}
