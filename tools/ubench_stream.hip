// The streamed-row solve's contact steps on REGISTER-HELD rows (no loads), priced before building anything into pgs_v1
// (VERDICT r5 item 2).  Today's steps -- row_step_normal2 (two normals per step, one coupling scalar) and row_step_cone (one
// contact's friction pair) of snk_device.hpp, the real ones -- against the four-row forms:
//   row_step_normal4   four normals' dots from ONE delta-v, their four reductions interleaved (every DPP add is the other
//                      three's wait state: no s_nop left), rows 2..4 corrected by six coupling scalars
//                      c_ij = (J_i / den_i) . (M^-1 J_j^T)  -- exactly the sequential sweep, as the 16-link quad_step
//   row_step_cone2     two contacts' friction pairs: four dots from one delta-v, the second pair corrected by the first
//                      pair's impulse changes through a 2 x 2 coupling block
// Lanes 0..39 take part, as in pgs_v1 (38 velocity components + two scalar columns); impulses and couplings come from LDS
// one step ahead and go back to it, as in pgs_v1.  One wave per workgroup, 1 and 2 waves per SIMD (occupancy through the
// dynamic LDS size).  MOTORS: the 32 motor rows of an iteration (pgs_v1's motor_rows: one v_readlane chain per motor).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/ubench_stream.hip -o tools/ubench_stream
//   tools/ubench_stream [iterations per wave, default 2000]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/snk.h"
#include "../bullet-envs_amd/csrc/snk_device.hpp"

#define SNK_RED64x4_STEP(A, B, C, D, MODE)               \
    "v_add_f32_dpp " A ", " A ", " A " " MODE "\n\t"      \
    "v_add_f32_dpp " B ", " B ", " B " " MODE "\n\t"      \
    "v_add_f32_dpp " C ", " C ", " C " " MODE "\n\t"      \
    "v_add_f32_dpp " D ", " D ", " D " " MODE "\n\t"

// four consecutive contact normals.  cp = {c21, c31, c32, c41}, cq = {c42, c43}.  56 VALU, no s_nop: 14 issue slots per
// row against row_step_normal2's 18.
template <int SUM_LANE>
__device__ __forceinline__ void row_step_normal4(float jA, float mA, float jB, float mB, float jC, float mC, float jD, float mD,
                                                 float& accA, float& accB, float& accC, float& accD, float c21, float c31,
                                                 float c32, float c41, float c42, float c43, float& dv, float& lsq) {
    float tA, tB, tC, tD, xA, xB, xC, xD, dA, dB, dC, dD, sA, sB, sC, sD;
    asm volatile(
        "v_mul_f32 %[tA], %[jA], %[dv]\n\t"
        "v_mul_f32 %[tB], %[jB], %[dv]\n\t"
        "v_mul_f32 %[tC], %[jC], %[dv]\n\t"
        "v_mul_f32 %[tD], %[jD], %[dv]\n\t"
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "row_bcast:31 row_mask:0xc bank_mask:0xf")
        "v_readlane_b32 %[sA], %[tA], %[SL]\n\t"
        "v_readlane_b32 %[sB], %[tB], %[SL]\n\t"
        "v_readlane_b32 %[sC], %[tC], %[SL]\n\t"
        "v_readlane_b32 %[sD], %[tD], %[SL]\n\t"
        "v_subrev_f32 %[xA], %[sA], %[accA]\n\t"
        "v_subrev_f32 %[xB], %[sB], %[accB]\n\t"
        "v_subrev_f32 %[xC], %[sC], %[accC]\n\t"
        "v_subrev_f32 %[xD], %[sD], %[accD]\n\t"
        "v_max_f32 %[xA], 0, %[xA]\n\t"
        "v_sub_f32 %[dA], %[xA], %[accA]\n\t"
        "v_fma_f32 %[xB], -%[c21], %[dA], %[xB]\n\t"
        "v_fma_f32 %[xC], -%[c31], %[dA], %[xC]\n\t"
        "v_fma_f32 %[xD], -%[c41], %[dA], %[xD]\n\t"
        "v_max_f32 %[xB], 0, %[xB]\n\t"
        "v_mul_f32 %[tA], %[dA], %[mA]\n\t"
        "v_sub_f32 %[dB], %[xB], %[accB]\n\t"
        "v_add_f32 %[dv], %[dv], %[tA]\n\t"
        "v_fma_f32 %[xC], -%[c32], %[dB], %[xC]\n\t"
        "v_fma_f32 %[xD], -%[c42], %[dB], %[xD]\n\t"
        "v_mul_f32 %[tB], %[dB], %[mB]\n\t"
        "v_max_f32 %[xC], 0, %[xC]\n\t"
        "v_add_f32 %[dv], %[dv], %[tB]\n\t"
        "v_sub_f32 %[dC], %[xC], %[accC]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[tA]|, |%[tB]|\n\t"
        "v_fma_f32 %[xD], -%[c43], %[dC], %[xD]\n\t"
        "v_mul_f32 %[tC], %[dC], %[mC]\n\t"
        "v_max_f32 %[xD], 0, %[xD]\n\t"
        "v_add_f32 %[dv], %[dv], %[tC]\n\t"
        "v_sub_f32 %[dD], %[xD], %[accD]\n\t"
        "v_mul_f32 %[tD], %[dD], %[mD]\n\t"
        "v_add_f32 %[dv], %[dv], %[tD]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[tC]|, |%[tD]|\n\t"
        : [tA] "=&v"(tA), [tB] "=&v"(tB), [tC] "=&v"(tC), [tD] "=&v"(tD), [xA] "=&v"(xA), [xB] "=&v"(xB), [xC] "=&v"(xC),
          [xD] "=&v"(xD), [dA] "=&v"(dA), [dB] "=&v"(dB), [dC] "=&v"(dC), [dD] "=&v"(dD), [sA] "=&s"(sA), [sB] "=&s"(sB),
          [sC] "=&s"(sC), [sD] "=&s"(sD), [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [jA] "v"(jA), [mA] "v"(mA), [jB] "v"(jB), [mB] "v"(mB), [jC] "v"(jC), [mC] "v"(mC), [jD] "v"(jD), [mD] "v"(mD),
          [accA] "v"(accA), [accB] "v"(accB), [accC] "v"(accC), [accD] "v"(accD), [c21] "v"(c21), [c31] "v"(c31),
          [c32] "v"(c32), [c41] "v"(c41), [c42] "v"(c42), [c43] "v"(c43), [SL] "n"(SUM_LANE));
    accA = xA; accB = xB; accC = xC; accD = xD;
}

// The same four normals with a TRANSPOSED reduction (round 6, counted in profiles/r06_ubench_stream.txt, then measured):
// all 64 lanes enabled, lanes 40..63 of the rows exact zeros.  v_permlane32_swap folds two 40-lane product vectors into
// the halves of one register ([A | B], [C | D]); after the first DPP stage every lane pair holds its sum twice, so one
// v_cndmask merges the two registers (even lanes AB, odd lanes CD); three more DPP stages leave the row sums by class in
// lanes 12..15 of every row of 16; a v_permlane16_swap of the register with its copy adds rows 0+1 and 2+3.  A's dot ends in
// lane 14, C's in lane 15, B's in lane 46, D's in lane 47.  21 instructions for four dots instead of 32; 45 VALU in all.
__device__ __forceinline__ void row_step_normal4t(float jA, float mA, float jB, float mB, float jC, float mC, float jD, float mD,
                                                  float& accA, float& accB, float& accC, float& accD, float c21, float c31,
                                                  float c32, float c41, float c42, float c43, unsigned long long odd,
                                                  float& dv, float& lsq) {
    float tA, tB, tC, tD, xA, xB, xC, xD, dA, dB, dC, dD, sA, sB, sC, sD;
    asm volatile(
        "v_mul_f32 %[tA], %[jA], %[dv]\n\t"
        "v_mul_f32 %[tB], %[jB], %[dv]\n\t"
        "v_mul_f32 %[tC], %[jC], %[dv]\n\t"
        "v_mul_f32 %[tD], %[jD], %[dv]\n\t"
        "v_permlane32_swap_b32 %[tA], %[tB]\n\t"
        "s_nop 0\n\t"
        "v_permlane32_swap_b32 %[tC], %[tD]\n\t"
        "v_add_f32 %[tA], %[tA], %[tB]\n\t"
        "v_add_f32 %[tC], %[tC], %[tD]\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[tA], %[tA], %[tA] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %[tC], %[tC], %[tC] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32_e64 %[tA], %[tA], %[tC], %[odd]\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[tA], %[tA], %[tA] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[tA], %[tA], %[tA] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[tA], %[tA], %[tA] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32 %[tB], %[tA]\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %[tA], %[tB]\n\t"
        "v_add_f32 %[tA], %[tA], %[tB]\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[sA], %[tA], 14\n\t"
        "v_readlane_b32 %[sC], %[tA], 15\n\t"
        "v_readlane_b32 %[sB], %[tA], 46\n\t"
        "v_readlane_b32 %[sD], %[tA], 47\n\t"
        "v_subrev_f32 %[xA], %[sA], %[accA]\n\t"
        "v_subrev_f32 %[xB], %[sB], %[accB]\n\t"
        "v_subrev_f32 %[xC], %[sC], %[accC]\n\t"
        "v_subrev_f32 %[xD], %[sD], %[accD]\n\t"
        "v_max_f32 %[xA], 0, %[xA]\n\t"
        "v_sub_f32 %[dA], %[xA], %[accA]\n\t"
        "v_fma_f32 %[xB], -%[c21], %[dA], %[xB]\n\t"
        "v_fma_f32 %[xC], -%[c31], %[dA], %[xC]\n\t"
        "v_fma_f32 %[xD], -%[c41], %[dA], %[xD]\n\t"
        "v_max_f32 %[xB], 0, %[xB]\n\t"
        "v_mul_f32 %[tA], %[dA], %[mA]\n\t"
        "v_sub_f32 %[dB], %[xB], %[accB]\n\t"
        "v_add_f32 %[dv], %[dv], %[tA]\n\t"
        "v_fma_f32 %[xC], -%[c32], %[dB], %[xC]\n\t"
        "v_fma_f32 %[xD], -%[c42], %[dB], %[xD]\n\t"
        "v_mul_f32 %[tB], %[dB], %[mB]\n\t"
        "v_max_f32 %[xC], 0, %[xC]\n\t"
        "v_add_f32 %[dv], %[dv], %[tB]\n\t"
        "v_sub_f32 %[dC], %[xC], %[accC]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[tA]|, |%[tB]|\n\t"
        "v_fma_f32 %[xD], -%[c43], %[dC], %[xD]\n\t"
        "v_mul_f32 %[tC], %[dC], %[mC]\n\t"
        "v_max_f32 %[xD], 0, %[xD]\n\t"
        "v_add_f32 %[dv], %[dv], %[tC]\n\t"
        "v_sub_f32 %[dD], %[xD], %[accD]\n\t"
        "v_mul_f32 %[tD], %[dD], %[mD]\n\t"
        "v_add_f32 %[dv], %[dv], %[tD]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[tC]|, |%[tD]|\n\t"
        : [tA] "=&v"(tA), [tB] "=&v"(tB), [tC] "=&v"(tC), [tD] "=&v"(tD), [xA] "=&v"(xA), [xB] "=&v"(xB), [xC] "=&v"(xC),
          [xD] "=&v"(xD), [dA] "=&v"(dA), [dB] "=&v"(dB), [dC] "=&v"(dC), [dD] "=&v"(dD), [sA] "=&s"(sA), [sB] "=&s"(sB),
          [sC] "=&s"(sC), [sD] "=&s"(sD), [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [jA] "v"(jA), [mA] "v"(mA), [jB] "v"(jB), [mB] "v"(mB), [jC] "v"(jC), [mC] "v"(mC), [jD] "v"(jD), [mD] "v"(mD),
          [accA] "v"(accA), [accB] "v"(accB), [accC] "v"(accC), [accD] "v"(accD), [c21] "v"(c21), [c31] "v"(c31),
          [c32] "v"(c32), [c41] "v"(c41), [c42] "v"(c42), [c43] "v"(c43), [odd] "s"(odd));
    accA = xA; accB = xB; accC = xC; accD = xD;
}

// two contacts' cone-friction pairs (X: rows A, B; Y: rows C, D).  Y's dots are corrected by X's impulse changes:
// yC -= cCA dXA + cCB dXB,  yD -= cDA dXA + cDB dXB.  70 VALU, no s_nop: 35 issue slots per contact against row_step_cone's 39.
template <int SUM_LANE>
__device__ __forceinline__ void row_step_cone2(float jA, float mA, float jB, float mB, float jC, float mC, float jD, float mD,
                                               float& accA, float& accB, float& accC, float& accD, float limX, float limY,
                                               float cCA, float cCB, float cDA, float cDB, float EPS, float& dv, float& lsq) {
    float tA, tB, tC, tD, xA, xB, xC, xD, r2, q2, sA, sB, sC, sD;
    asm volatile(
        "v_mul_f32 %[tA], %[jA], %[dv]\n\t"
        "v_mul_f32 %[tB], %[jB], %[dv]\n\t"
        "v_mul_f32 %[tC], %[jC], %[dv]\n\t"
        "v_mul_f32 %[tD], %[jD], %[dv]\n\t"
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        SNK_RED64x4_STEP("%[tA]", "%[tB]", "%[tC]", "%[tD]", "row_bcast:31 row_mask:0xc bank_mask:0xf")
        "v_readlane_b32 %[sA], %[tA], %[SL]\n\t"
        "v_readlane_b32 %[sB], %[tB], %[SL]\n\t"
        "v_readlane_b32 %[sC], %[tC], %[SL]\n\t"
        "v_readlane_b32 %[sD], %[tD], %[SL]\n\t"
        "v_subrev_f32 %[xA], %[sA], %[accA]\n\t"
        "v_subrev_f32 %[xB], %[sB], %[accB]\n\t"
        "v_subrev_f32 %[xC], %[sC], %[accC]\n\t"
        "v_subrev_f32 %[xD], %[sD], %[accD]\n\t"
        "v_fma_f32 %[r2], %[xA], %[xA], %[EPS]\n\t"
        "v_fma_f32 %[r2], %[xB], %[xB], %[r2]\n\t"
        "v_rsq_f32 %[r2], %[r2]\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_e64 %[r2], %[limX], %[r2] clamp\n\t"
        "v_mul_f32 %[xA], %[xA], %[r2]\n\t"
        "v_mul_f32 %[xB], %[xB], %[r2]\n\t"
        "v_sub_f32 %[tA], %[xA], %[accA]\n\t"
        "v_sub_f32 %[tB], %[xB], %[accB]\n\t"
        "v_fma_f32 %[xC], -%[cCA], %[tA], %[xC]\n\t"
        "v_fma_f32 %[xD], -%[cDA], %[tA], %[xD]\n\t"
        "v_mul_f32 %[tA], %[tA], %[mA]\n\t"
        "v_fma_f32 %[xC], -%[cCB], %[tB], %[xC]\n\t"
        "v_fma_f32 %[xD], -%[cDB], %[tB], %[xD]\n\t"
        "v_mul_f32 %[tB], %[tB], %[mB]\n\t"
        "v_fma_f32 %[q2], %[xC], %[xC], %[EPS]\n\t"
        "v_add_f32 %[dv], %[dv], %[tA]\n\t"
        "v_fma_f32 %[q2], %[xD], %[xD], %[q2]\n\t"
        "v_add_f32 %[dv], %[dv], %[tB]\n\t"
        "v_rsq_f32 %[q2], %[q2]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[tA]|, |%[tB]|\n\t"
        "v_mul_f32_e64 %[q2], %[limY], %[q2] clamp\n\t"
        "v_mul_f32 %[xC], %[xC], %[q2]\n\t"
        "v_mul_f32 %[xD], %[xD], %[q2]\n\t"
        "v_sub_f32 %[tC], %[xC], %[accC]\n\t"
        "v_sub_f32 %[tD], %[xD], %[accD]\n\t"
        "v_mul_f32 %[tC], %[tC], %[mC]\n\t"
        "v_mul_f32 %[tD], %[tD], %[mD]\n\t"
        "v_add_f32 %[dv], %[dv], %[tC]\n\t"
        "v_add_f32 %[dv], %[dv], %[tD]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[tC]|, |%[tD]|\n\t"
        : [tA] "=&v"(tA), [tB] "=&v"(tB), [tC] "=&v"(tC), [tD] "=&v"(tD), [xA] "=&v"(xA), [xB] "=&v"(xB), [xC] "=&v"(xC),
          [xD] "=&v"(xD), [r2] "=&v"(r2), [q2] "=&v"(q2), [sA] "=&s"(sA), [sB] "=&s"(sB), [sC] "=&s"(sC), [sD] "=&s"(sD),
          [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [jA] "v"(jA), [mA] "v"(mA), [jB] "v"(jB), [mB] "v"(mB), [jC] "v"(jC), [mC] "v"(mC), [jD] "v"(jD), [mD] "v"(mD),
          [accA] "v"(accA), [accB] "v"(accB), [accC] "v"(accC), [accD] "v"(accD), [limX] "v"(limX), [limY] "v"(limY),
          [cCA] "v"(cCA), [cCB] "v"(cCB), [cDA] "v"(cDA), [cDB] "v"(cDB), [EPS] "v"(EPS), [SL] "n"(SUM_LANE));
    accA = xA; accB = xB; accC = xC; accD = xD;
}

struct MockLds {
    float acc[64][4];        // {normal impulse, friction A, friction B, coupling with the previous normal}: pgs_v1's L.acc
    float cn[16][8];         // per quad of normals: c21 c31 c32 c41 c42 c43 (two spare)
    float cf[32][4];         // per pair of contacts: the 2 x 2 block
};

constexpr int kMO = 40;

// MODE bit 0: normals four at a time; bit 1: friction two contacts at a time; bit 2: the 32 motor rows as well
// WHAT: 0 normals + friction, 1 normals only, 2 friction only
template <int NC, int MODE, int WHAT>
__global__ __launch_bounds__(64, 2) void stream_mock(const float* __restrict__ in, float* __restrict__ out, int n_iter,
                                                     long long* __restrict__ ticks) {
    extern __shared__ float4 smem_raw[];
    MockLds& L = *reinterpret_cast<MockLds*>(smem_raw);
    const int lane = threadIdx.x;
    float NJ[NC], NM[NC], FJA[NC], FJB[NC], FMA[NC], FMB[NC], RMm[32];
    const bool col = lane < 38;
#pragma unroll
    for (int s = 0; s < NC; s++) {
        const float* p = in + (size_t)(6 * s) * 64 + lane;
        NJ[s] = col ? 0.05f * p[0] : (lane == 38 ? 0.01f * p[0] : 0.f);
        NM[s] = col ? 0.05f * p[64] : (lane == 39 ? 1.0f : 0.f);
        FJA[s] = col ? 0.05f * p[128] : 0.f;
        FJB[s] = col ? 0.05f * p[192] : 0.f;
        FMA[s] = col ? 0.05f * p[256] : (lane == 39 ? 1.0f : 0.f);
        FMB[s] = col ? 0.05f * p[320] : (lane == 39 ? 1.0f : 0.f);
    }
#pragma unroll
    for (int j = 0; j < 32; j++) RMm[j] = col ? 0.02f * in[(size_t)(400 + j) * 64 + lane] : 0.f;
    const float TARGV = (lane >= 6 && lane < 38) ? 0.1f * in[440 * 64 + lane] : 0.f;
    const float DINVV = (lane >= 6 && lane < 38) ? 0.7f : 0.f;
    float ACCV = 0.f;
    for (int i = lane; i < 64 * 4; i += 64) (&L.acc[0][0])[i] = (i & 3) == 3 ? 0.01f * in[i] : 0.f;
    for (int i = lane; i < 16 * 8; i += 64) (&L.cn[0][0])[i] = 0.01f * in[300 + i];
    for (int i = lane; i < 32 * 4; i += 64) (&L.cf[0][0])[i] = 0.01f * in[500 + i];
    __syncthreads();
    float dv = lane == 38 ? 1.0f : 0.f;
    float lsq = 0.f;
    const float EPS = 1e-30f, mu = 2.0f;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    __builtin_amdgcn_s_setprio(3);
    const unsigned long long kOdd = 0xAAAAAAAAAAAAAAAAull;
    if (lane < kMO || (MODE & 8)) {
        for (int it = 0; it < n_iter; it++) {
#pragma unroll
            for (int s = 0; s < NC; s++)
                asm volatile("" : "+v"(NJ[s]), "+v"(NM[s]), "+v"(FJA[s]), "+v"(FJB[s]), "+v"(FMA[s]), "+v"(FMB[s]));
            if (MODE & 4) {
                float Uv = 0.f;
#pragma unroll
                for (int j = 0; j < 32; j++) {
                    float u = (TARGV - dv) * DINVV;
                    const float sdI = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), 6 + j));
                    Uv = (lane == 6 + j) ? u : Uv;
                    dv += sdI * RMm[j];
                }
                ACCV += Uv;
            }
            if (WHAT != 2) {
                if (MODE & 9) {
                    float4 an = make_float4(L.acc[0][0], L.acc[1][0], L.acc[2][0], L.acc[3][0]);
                    float4 cp = *reinterpret_cast<const float4*>(&L.cn[0][0]);
                    float2 cq = *reinterpret_cast<const float2*>(&L.cn[0][4]);
#pragma unroll
                    for (int k = 0; k < NC; k += 4) {
                        float a0 = an.x, a1 = an.y, a2 = an.z, a3 = an.w;
                        const float4 c = cp;
                        const float2 e = cq;
                        const int kn = (k + 4) & 63;
                        an = make_float4(L.acc[kn][0], L.acc[kn + 1][0], L.acc[kn + 2][0], L.acc[kn + 3][0]);
                        cp = *reinterpret_cast<const float4*>(&L.cn[kn / 4][0]);
                        cq = *reinterpret_cast<const float2*>(&L.cn[kn / 4][4]);
                        if (MODE & 8)
                            row_step_normal4t(NJ[k], NM[k], NJ[k + 1], NM[k + 1], NJ[k + 2], NM[k + 2], NJ[k + 3], NM[k + 3], a0, a1, a2,
                                              a3, c.x, c.y, c.z, c.w, e.x, e.y, kOdd, dv, lsq);
                        else
                            row_step_normal4<kMO - 1>(NJ[k], NM[k], NJ[k + 1], NM[k + 1], NJ[k + 2], NM[k + 2], NJ[k + 3], NM[k + 3], a0,
                                                      a1, a2, a3, c.x, c.y, c.z, c.w, e.x, e.y, dv, lsq);
                        L.acc[k][0] = a0; L.acc[k + 1][0] = a1; L.acc[k + 2][0] = a2; L.acc[k + 3][0] = a3;
                    }
                } else {
                    float a0n = L.acc[0][0];
                    float2 a1n = make_float2(L.acc[1][0], L.acc[1][3]);
#pragma unroll
                    for (int k = 0; k < NC; k += 2) {
                        float a0 = a0n, a1 = a1n.x;
                        const float c1 = a1n.y;
                        a0n = L.acc[(k + 2) & 63][0];
                        a1n = make_float2(L.acc[(k + 3) & 63][0], L.acc[(k + 3) & 63][3]);
                        snk::row_step_normal2<kMO - 1>(NJ[k], NM[k], NJ[k + 1], NM[k + 1], a0, a1, c1, dv, lsq);
                        L.acc[k][0] = a0;
                        L.acc[k + 1][0] = a1;
                    }
                }
            }
            if (WHAT != 1) {
                if (MODE & 2) {
                    float4 fa = *reinterpret_cast<const float4*>(L.acc[0]), fb = *reinterpret_cast<const float4*>(L.acc[1]);
                    float4 cc = *reinterpret_cast<const float4*>(L.cf[0]);
#pragma unroll
                    for (int k = 0; k < NC; k += 2) {
                        const float4 x = fa, y = fb, c = cc;
                        fa = *reinterpret_cast<const float4*>(L.acc[(k + 2) & 63]);
                        fb = *reinterpret_cast<const float4*>(L.acc[(k + 3) & 63]);
                        cc = *reinterpret_cast<const float4*>(L.cf[((k + 2) & 63) / 2]);
                        float aA = x.y, aB = x.z, aC = y.y, aD = y.z;
                        row_step_cone2<kMO - 1>(FJA[k], FMA[k], FJB[k], FMB[k], FJA[k + 1], FMA[k + 1], FJB[k + 1], FMB[k + 1], aA, aB,
                                                aC, aD, mu * x.x, mu * y.x, c.x, c.y, c.z, c.w, EPS, dv, lsq);
                        *reinterpret_cast<float2*>(&L.acc[k][1]) = make_float2(aA, aB);
                        *reinterpret_cast<float2*>(&L.acc[k + 1][1]) = make_float2(aC, aD);
                    }
                } else {
                    float4 fn = *reinterpret_cast<const float4*>(L.acc[0]);
#pragma unroll
                    for (int k = 0; k < NC; k++) {
                        const float4 c = fn;
                        fn = *reinterpret_cast<const float4*>(L.acc[(k + 1) & 63]);
                        float aA = c.y, aB = c.z;
                        snk::row_step_cone<kMO - 1>(FJA[k], FMA[k], FJB[k], FMB[k], aA, aB, mu * c.x, EPS, dv, lsq);
                        *reinterpret_cast<float2*>(&L.acc[k][1]) = make_float2(aA, aB);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float acc = dv + lsq + ACCV;
#pragma unroll
    for (int s = 0; s < NC; s++) acc += NJ[s] + FJA[s];
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

template <int NC, int MODE, int WHAT>
void run(const char* name, const float* d_in, float* d_out, long long* d_ticks, int n_iter, int n_cu) {
    auto kern = stream_mock<NC, MODE, WHAT>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
    // issue slots (VALU + s_nop) of one iteration, hand count of the asm blocks
    const double nrm = WHAT == 2 ? 0.0 : ((MODE & 8) ? (NC / 4) * 52.0 : (MODE & 1) ? (NC / 4) * 56.0 : (NC / 2) * 36.0);
    const double nrm_valu = WHAT == 2 ? 0.0 : ((MODE & 8) ? (NC / 4) * 45.0 : (MODE & 1) ? (NC / 4) * 56.0 : (NC / 2) * 28.0);
    const double frc = WHAT == 1 ? 0.0 : ((MODE & 2) ? (NC / 2) * 71.0 : NC * 39.0);
    const double frc_valu = WHAT == 1 ? 0.0 : ((MODE & 2) ? (NC / 2) * 70.0 : NC * 31.0);
    const double mot = (MODE & 4) ? 32 * 5.0 : 0.0;
    const int rows = (WHAT == 2 ? 0 : NC) + (WHAT == 1 ? 0 : 2 * NC);
    for (int w = 1; w <= 2; w++) {
        const size_t lds = (size_t)(160 * 1024 / (4 * w) / 1024) * 1024;
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
        printf("%-34s contacts %2d  vgprs %3d scratch %3d B  waves/SIMD %d  launch %8.3f ms  iterations/s/SIMD %9.0f  "
               "clocks per contact row and SIMD %6.1f  VALU/clk/SIMD %.3f  slots/clk/SIMD %.3f (at 2.4 GHz)\n",
               name, NC, fa.numRegs, (int)fa.localSizeBytes, w, ms, it_per_s_simd, 2.4e9 / (it_per_s_simd * (rows ? rows : 1)),
               it_per_s_simd * (nrm_valu + frc_valu + mot) / 2.4e9, it_per_s_simd * (nrm + frc + mot) / 2.4e9);
        fflush(stdout);
    }
}

// ---- self-check of the transposed reduction: the four sums against a plain per-row sum on the host
__global__ void check_kernel(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x;
    float j[4], m[4];
    for (int r = 0; r < 4; r++) { j[r] = lane < 40 ? in[r * 64 + lane] : 0.f; m[r] = 0.f; }
    float dv = lane < 40 ? in[4 * 64 + lane] : 0.f;
    // acc = 1e30 and couplings 0: x = max(acc - s, 0) = acc - s exactly representable?  no: read the sums back through acc = 0
    // and the sign: x_r = max(0 - s_r, 0); run twice, with the rows as they are and negated
    for (int sign = 0; sign < 2; sign++) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, lsq = 0.f, d = dv;
        const float sg = sign ? -1.f : 1.f;
        row_step_normal4t(sg * j[0], m[0], sg * j[1], m[1], sg * j[2], m[2], sg * j[3], m[3], a0, a1, a2, a3, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                          0xAAAAAAAAAAAAAAAAull, d, lsq);
        if (lane == 0) { out[4 * sign + 0] = a0; out[4 * sign + 1] = a1; out[4 * sign + 2] = a2; out[4 * sign + 3] = a3; }
    }
}

static void check_transposed(const float* d_in) {
    float* d_out;
    CHECK(hipMalloc(&d_out, 8 * sizeof(float)));
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, nullptr, d_in, d_out);
    float got[8];
    CHECK(hipMemcpy(got, d_out, sizeof(got), hipMemcpyDeviceToHost));
    std::vector<float> h(5 * 64);
    CHECK(hipMemcpy(h.data(), d_in, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int r = 0; r < 4; r++) {
        double s = 0;
        for (int l = 0; l < 40; l++) s += (double)h[r * 64 + l] * (double)h[4 * 64 + l];
        // x = max(-s, 0) for the rows as they are, max(s, 0) for the negated ones
        const double want0 = s < 0 ? -s : 0, want1 = s > 0 ? s : 0;
        worst = fmax(worst, fmax(fabs(got[r] - want0), fabs(got[4 + r] - want1)));
        printf("transposed reduction check: row %d  dot %+.6f  got %+.6f / %+.6f\n", r, s, got[r], got[4 + r]);
    }
    printf("transposed reduction check: worst difference %.2e %s\n", worst, worst < 1e-5 ? "(ok)" : "(WRONG)");
}

int main(int argc, char** argv) {
    const int n_iter = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s: %d CUs, %d iterations per wave\n", prop.name, n_cu, n_iter);
    std::vector<float> h(1024 * 64);
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    float *d_in, *d_out;
    long long* d_ticks;
    CHECK(hipMalloc(&d_in, h.size() * sizeof(float)));
    CHECK(hipMemcpy(d_in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 16 * 64 * sizeof(float)));
    CHECK(hipMalloc(&d_ticks, (size_t)n_cu * 16 * sizeof(long long)));
    constexpr int NC = 24;
    run<NC, 0, 1>("normals: normal2 (shipped)", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 1, 1>("normals: normal4", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 8, 1>("normals: normal4t (transposed)", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 0, 2>("friction: cone (shipped)", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 2, 2>("friction: cone2", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 0, 0>("iteration: normal2 + cone", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 3, 0>("iteration: normal4 + cone2", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 1, 0>("iteration: normal4 + cone", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 4, 0>("iteration + 32 motors: shipped", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 7, 0>("iteration + 32 motors: 4-row", d_in, d_out, d_ticks, n_iter, n_cu);
    run<NC, 8, 0>("iteration: normal4t + cone", d_in, d_out, d_ticks, n_iter, n_cu);
    check_transposed(d_in);
    return 0;
}
