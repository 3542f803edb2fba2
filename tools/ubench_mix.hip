// What one SIMD sustains on the INSTRUCTION CLASSES of the register-resident solve's row steps, alone and in their
// dependency patterns, at 1, 2 and 4 waves per SIMD (round 4: the real steps saturate at ~0.25 VALU / clock / SIMD whatever
// the occupancy -- tools/ubench_solve.hip -- although a SIMD-32 issues a plain v_fma every 2 clocks; which class costs
// what?).  One wave per workgroup, occupancy set through the dynamic LDS size, 128-register kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mix.hip -o tools/ubench_mix && tools/ubench_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define DECL_KERNEL(NAME, NINSTR, BODY)                                                                                 \
    constexpr int kN_##NAME = NINSTR;                                                                                   \
    __global__ __launch_bounds__(64, 4) void k_##NAME(const float* __restrict__ in, float* __restrict__ out, int n_iter) { \
        const int lane = threadIdx.x;                                                                                   \
        float a = in[lane], b = in[64 + lane], c = in[128 + lane], d = in[192 + lane], e = in[256 + lane],              \
              f = in[320 + lane], g = in[384 + lane], h = in[448 + lane];                                               \
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                                                                   \
        typedef float v2f __attribute__((ext_vector_type(2)));                                                          \
        v2f pa = {a, b}, pc = {c, d}, pe = {e, f};                                                                      \
        const unsigned long long lowmask = 0xFFFFFFFFull;                                                               \
        const float sconst = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(in[7])));                     \
        for (int it = 0; it < n_iter; it++) {                                                                           \
            asm volatile(REP16(BODY)                                                                                    \
                         : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d), [e] "+v"(e), [f] "+v"(f), [g] "+v"(g),   \
                           [h] "+v"(h), [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3), [pa] "+v"(pa),    \
                           [pc] "+v"(pc), [pe] "+v"(pe)                                                                 \
                         : [lowmask] "s"(lowmask), [sc] "s"(sconst) : "vcc");                                           \
        }                                                                                                               \
        out[(size_t)blockIdx.x * 64 + lane] = a + b + c + d + e + f + g + h + pa.x + pa.y + pc.x + pc.y + s0 + s1 + s2 + s3;                                            \
    }

// 4 independent plain FMAs
DECL_KERNEL(fma, 4, "v_fma_f32 %[a], %[a], %[e], %[f]\n\tv_fma_f32 %[b], %[b], %[e], %[f]\n\tv_fma_f32 %[c], %[c], %[e], %[f]\n\tv_fma_f32 %[d], %[d], %[e], %[f]\n\t")
// 4 independent VOP2 adds (4-byte encodings)
DECL_KERNEL(add, 4, "v_add_f32 %[a], %[a], %[e]\n\tv_add_f32 %[b], %[b], %[e]\n\tv_add_f32 %[c], %[c], %[e]\n\tv_add_f32 %[d], %[d], %[e]\n\t")
// 4 independent DPP adds (two interleaved pairs: no wait states needed between different registers)
DECL_KERNEL(dpp, 4, "v_add_f32_dpp %[a], %[e], %[a] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %[b], %[e], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %[c], %[e], %[c] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %[d], %[e], %[d] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
// the reduction pattern of quad_step: two interleaved dependent DPP chains
DECL_KERNEL(dppchain, 4, "v_add_f32_dpp %[a], %[a], %[a] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %[b], %[b], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32 %[c], %[e]\n\tv_add_f32_dpp %[a], %[a], %[a] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
// readlanes whose results nothing reads
DECL_KERNEL(readlane, 4, "v_readlane_b32 %[s0], %[a], 31\n\tv_readlane_b32 %[s1], %[b], 63\n\tv_readlane_b32 %[s2], %[c], 31\n\tv_readlane_b32 %[s3], %[d], 63\n\t")
// readlane -> VALU that reads the SGPR (the scalar round trip of every row step)
DECL_KERNEL(rl_use, 4, "v_readlane_b32 %[s0], %[a], 31\n\tv_readlane_b32 %[s1], %[b], 63\n\ts_nop 1\n\tv_max_f32_e64 %[c], -%[s0], 0\n\tv_max_f32_e64 %[d], -%[s1], 0\n\t")
// VALU with an SGPR operand that was NOT just written
DECL_KERNEL(sgpr_src, 4, "v_max_f32_e64 %[a], -%[sc], 0\n\tv_max_f32_e64 %[b], -%[sc], 0\n\tv_max_f32_e64 %[c], -%[sc], 0\n\tv_max_f32_e64 %[d], -%[sc], 0\n\t")
// cndmask on an SGPR-pair mask
DECL_KERNEL(cndmask, 4, "v_cndmask_b32_e64 %[a], %[e], %[f], %[lowmask]\n\tv_cndmask_b32_e64 %[b], %[e], %[f], %[lowmask]\n\tv_cndmask_b32_e64 %[c], %[e], %[f], %[lowmask]\n\tv_cndmask_b32_e64 %[d], %[e], %[f], %[lowmask]\n\t")
// fmac with DPP source
DECL_KERNEL(fmac_dpp, 4, "v_fmac_f32_dpp %[a], %[e], %[f] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f32_dpp %[b], %[e], %[f] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f32_dpp %[c], %[e], %[f] row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fmac_f32_dpp %[d], %[e], %[f] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
// permlane32 swap + the two adds that follow it in every step
DECL_KERNEL(swap, 3, "v_permlane32_swap_b32 %[a], %[b]\n\tv_add_f32 %[c], %[c], %[a]\n\tv_add_f32 %[c], %[c], %[b]\n\t")
// s_nop only (does a wave's s_nop take the SIMD's issue slot away from the other waves?)
DECL_KERNEL(fma_nop, 4, "v_fma_f32 %[a], %[a], %[e], %[f]\n\ts_nop 1\n\tv_fma_f32 %[b], %[b], %[e], %[f]\n\ts_nop 1\n\tv_fma_f32 %[c], %[c], %[e], %[f]\n\ts_nop 1\n\tv_fma_f32 %[d], %[d], %[e], %[f]\n\ts_nop 1\n\t")
// max3 with abs modifiers (the residual)
DECL_KERNEL(max3, 4, "v_max3_f32 %[a], %[a], |%[e]|, |%[f]|\n\tv_max3_f32 %[b], %[b], |%[e]|, |%[f]|\n\tv_max3_f32 %[c], %[c], |%[e]|, |%[f]|\n\tv_max3_f32 %[d], %[d], |%[e]|, |%[f]|\n\t")
// rsq + the clamp multiply (cone projection)
DECL_KERNEL(rsq, 2, "v_rsq_f32 %[a], %[e]\n\tv_mul_f32_e64 %[b], %[sc], %[f] clamp\n\t")


// ---- which property makes an instruction expensive: the encoding (VOP2 4 bytes / VOP3, DPP 8 bytes), an SGPR operand, modifiers?
DECL_KERNEL(add_e64, 4, "v_add_f32_e64 %[a], %[a], %[e]\n\tv_add_f32_e64 %[b], %[b], %[e]\n\tv_add_f32_e64 %[c], %[c], %[e]\n\tv_add_f32_e64 %[d], %[d], %[e]\n\t")
DECL_KERNEL(mul, 4, "v_mul_f32 %[a], %[a], %[e]\n\tv_mul_f32 %[b], %[b], %[e]\n\tv_mul_f32 %[c], %[c], %[e]\n\tv_mul_f32 %[d], %[d], %[e]\n\t")
DECL_KERNEL(fmac, 4, "v_fmac_f32 %[a], %[e], %[f]\n\tv_fmac_f32 %[b], %[e], %[f]\n\tv_fmac_f32 %[c], %[e], %[f]\n\tv_fmac_f32 %[d], %[e], %[f]\n\t")
DECL_KERNEL(max3_plain, 4, "v_max3_f32 %[a], %[a], %[e], %[f]\n\tv_max3_f32 %[b], %[b], %[e], %[f]\n\tv_max3_f32 %[c], %[c], %[e], %[f]\n\tv_max3_f32 %[d], %[d], %[e], %[f]\n\t")
DECL_KERNEL(max_vop2_s, 4, "v_max_f32 %[a], %[sc], %[e]\n\tv_max_f32 %[b], %[sc], %[e]\n\tv_max_f32 %[c], %[sc], %[e]\n\tv_max_f32 %[d], %[sc], %[e]\n\t")
DECL_KERNEL(sub_vop2_s, 4, "v_subrev_f32 %[a], %[sc], %[a]\n\tv_subrev_f32 %[b], %[sc], %[b]\n\tv_subrev_f32 %[c], %[sc], %[c]\n\tv_subrev_f32 %[d], %[sc], %[d]\n\t")
DECL_KERNEL(fmac_s, 4, "v_fmac_f32 %[a], %[sc], %[f]\n\tv_fmac_f32 %[b], %[sc], %[f]\n\tv_fmac_f32 %[c], %[sc], %[f]\n\tv_fmac_f32 %[d], %[sc], %[f]\n\t")
DECL_KERNEL(cnd_vcc, 4, "v_cndmask_b32 %[a], %[e], %[f], vcc\n\tv_cndmask_b32 %[b], %[e], %[f], vcc\n\tv_cndmask_b32 %[c], %[e], %[f], vcc\n\tv_cndmask_b32 %[d], %[e], %[f], vcc\n\t")
DECL_KERNEL(mov, 4, "v_mov_b32 %[a], %[e]\n\tv_mov_b32 %[b], %[f]\n\tv_mov_b32 %[c], %[e]\n\tv_mov_b32 %[d], %[f]\n\t")
DECL_KERNEL(mov_dpp, 4, "v_mov_b32_dpp %[a], %[e] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %[b], %[f] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %[c], %[e] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %[d], %[f] row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
DECL_KERNEL(pk_fma, 2, "v_pk_fma_f32 %[pa], %[pe], %[pe], %[pa]\n\tv_pk_fma_f32 %[pc], %[pe], %[pe], %[pc]\n\t")
DECL_KERNEL(pk_mul, 2, "v_pk_mul_f32 %[pa], %[pe], %[pa]\n\tv_pk_mul_f32 %[pc], %[pe], %[pc]\n\t")
DECL_KERNEL(pk_add, 2, "v_pk_add_f32 %[pa], %[pe], %[pa]\n\tv_pk_add_f32 %[pc], %[pe], %[pc]\n\t")
// (a kernel mixing SALU ops into the stream hung in round 4: s_and / s_or write SCC, which an asm block here does not
//  declare, and the loop branch read it -- an infinite loop, not a GPU fault.  Left out.)

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

template <class K>
void run(const char* name, K kern, int ninstr, const float* d_in, float* d_out, int n_iter, int n_cu) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    printf("%-10s", name);
    for (int w : {1, 2, 4}) {
        const size_t lds = (size_t)(160 * 1024 / (4 * w) / 1024) * 1024;
        const int grid = n_cu * 4 * w;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, nullptr, d_in, d_out, 20);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, nullptr, d_in, d_out, n_iter);
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_s_simd = (double)w * n_iter * 16.0 * ninstr / (ms * 1e-3);
        printf("  w=%d: %6.3f VALU/clk/SIMD (%5.2f clk each)", w, instr_per_s_simd / 2.4e9, 2.4e9 / instr_per_s_simd);
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int n_iter = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s: %d CUs; clocks at a nominal 2.4 GHz; occupancy by LDS (placement of 1 / 2 waves over the SIMDs is the "
           "dispatcher's; w=4 is forced even by the 128-register bound)\n", prop.name, n_cu);
    std::vector<float> hbuf(512);
    for (int i = 0; i < 512; i++) hbuf[i] = 0.001f * (float)(i % 37);
    float *d_in, *d_out;
    CHECK(hipMalloc(&d_in, hbuf.size() * sizeof(float)));
    CHECK(hipMemcpy(d_in, hbuf.data(), hbuf.size() * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 16 * 64 * sizeof(float)));
#define RUN(NAME) run(#NAME, k_##NAME, kN_##NAME, d_in, d_out, n_iter, n_cu)
    RUN(fma); RUN(add); RUN(dpp); RUN(dppchain); RUN(readlane); RUN(rl_use); RUN(sgpr_src); RUN(cndmask); RUN(fmac_dpp);
    RUN(swap); RUN(fma_nop); RUN(max3); RUN(rsq);
    RUN(add_e64); RUN(mul); RUN(fmac); RUN(max3_plain); RUN(max_vop2_s); RUN(sub_vop2_s); RUN(fmac_s); RUN(cnd_vcc); RUN(mov);
    RUN(mov_dpp); RUN(pk_fma); RUN(pk_mul); RUN(pk_add);
    return 0;
}
