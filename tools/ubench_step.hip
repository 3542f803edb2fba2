// Isolated timing of the solve loop's row steps (ticks per step, 1 and 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 1000
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
namespace snk {
// Two consecutive single rows living in the two halves of one slot (hand-written, 25 VALU).
// ORDER 0: lower-half row first, then the upper-half row (coupling scalar in lane 57);
// ORDER 1: upper first, then lower (coupling in lane 25) -- motors are walked backwards on
// even iterations.  For each row:  a' = med3(-dot, LO, HI), dI = a' - a; the second row's dot
// first receives  c * dI_first.  Both contributions are exchanged between the halves, and the
// accumulated impulses (lane 23 / 55) are updated through E2355.
// RES: also accumulate max |dI * den| (lane 24 / 56) into lsq.
#define OLD_DUO_ASM(SEL)                                                                          \
    "v_mul_f32 %[t], %[RJ], %[dv]\n\t"                                                             \
    "v_readlane_b32 %[s2], %[RJ], %[AF]\n\t"                                                       \
    "v_readlane_b32 %[s3], %[RJ], %[AS]\n\t"                                                       \
    "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_readlane_b32 %[s4], %[RJ], %[CL]\n\t"                                                       \
    "s_nop 0\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                   \
    "s_nop 1\n\t"                                                                                  \
    "v_readlane_b32 %[s0], %[t], %[DF]\n\t"                                                        \
    "v_readlane_b32 %[s1], %[t], %[DS]\n\t"                                                        \
    "s_nop 0\n\t"                                                                                  \
    "v_med3_f32 %[dF], -%[s0], %[LO], %[HI]\n\t"                                                   \
    "v_mov_b32 %[x], %[s1]\n\t"                                                                    \
    "v_subrev_f32 %[dF], %[s2], %[dF]\n\t"                                                         \
    "v_fmac_f32 %[x], %[s4], %[dF]\n\t"                                                            \
    "v_med3_f32 %[x], -%[x], %[LO], %[HI]\n\t"                                                     \
    "v_subrev_f32 %[dS], %[s3], %[x]\n\t"                                                          \
    SEL                                                                                            \
    "v_mul_f32 %[x], %[RM], %[t]\n\t"                                                              \
    "v_mul_f32 %[c2], %[RM], %[t]\n\t"                                                             \
    "v_fmac_f32 %[RJ], %[E], %[t]\n\t"                                                             \
    "s_nop 0\n\t"                                                                                  \
    "v_permlane32_swap_b32 %[x], %[c2]\n\t"                                                        \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32 %[dv], %[dv], %[x]\n\t"                                                             \
    "v_add_f32 %[dv], %[dv], %[c2]\n\t"
template <int ORDER, bool RES>
__device__ __forceinline__ void old_duo_step(float& RJ, const float RM, float& dv, float LO, float HI, float E2355,
                                         unsigned long long lowmask, float& lsq) {
    float t, x, dF, dS, c2;
    float s0, s1, s2, s3, s4;
    // F = first row, S = second row.  Dots land in lane 31 (lower row) / 63 (upper row), the
    // accumulated impulses sit in lanes 23 / 55; v_cndmask gives every lane its own half's dI
    // (mask = lower half -> takes the lower row's value).
    if (ORDER == 0) {
        asm volatile(OLD_DUO_ASM("v_cndmask_b32_e64 %[t], %[dS], %[dF], %[lowmask]\n\t")
                     : [t] "=&v"(t), [x] "=&v"(x), [dF] "=&v"(dF), [dS] "=&v"(dS), [c2] "=&v"(c2), [s0] "=&s"(s0),
                       [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [RJ] "+v"(RJ), [dv] "+v"(dv)
                     : [RM] "v"(RM), [LO] "v"(LO), [HI] "v"(HI), [E] "v"(E2355), [lowmask] "s"(lowmask), [AF] "n"(23),
                       [AS] "n"(55), [CL] "n"(57), [DF] "n"(31), [DS] "n"(63));
    } else {
        asm volatile(OLD_DUO_ASM("v_cndmask_b32_e64 %[t], %[dF], %[dS], %[lowmask]\n\t")
                     : [t] "=&v"(t), [x] "=&v"(x), [dF] "=&v"(dF), [dS] "=&v"(dS), [c2] "=&v"(c2), [s0] "=&s"(s0),
                       [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [RJ] "+v"(RJ), [dv] "+v"(dv)
                     : [RM] "v"(RM), [LO] "v"(LO), [HI] "v"(HI), [E] "v"(E2355), [lowmask] "s"(lowmask), [AF] "n"(55),
                       [AS] "n"(23), [CL] "n"(25), [DF] "n"(63), [DS] "n"(31));
    }
    if (RES) {
        asm volatile("v_mul_f32 %[x], %[dI], %[RJ]\n\t"
                     "v_max_f32 %[lsq], %[lsq], |%[x]|\n\t"
                     "s_nop 1"
                     : [x] "=&v"(x), [lsq] "+v"(lsq)
                     : [dI] "v"(t), [RJ] "v"(RJ));
    } else {
        asm volatile("s_nop 1");
    }
}

// Bullet's cone-friction pair (hand-written, 28 VALU): direction A in the lower half and B in
// the upper half of one register, so one reduction yields both dots; the accumulated pair is
// projected radially onto the disc of radius mu * lambda_n (the normal's accumulated impulse:
// kept pre-multiplied in lane NL+4 of RJnorm); the two contributions to delta-v cross halves with v_permlane32_swap.
template <int NL, bool RES>
__device__ __forceinline__ void old_cone_step(float& RJ, const float RM, const float RJnorm, float& dv, float EPS,
                                          float E2355, unsigned long long lowmask, float& lsq) {
    float t, xA, xB, r2, c2;
    float s0, s1, s2, s3, s4;
    asm volatile(
        "v_mul_f32 %[t], %[RJ], %[dv]\n\t"
        "v_readlane_b32 %[s4], %[RJnorm], %[NLn]\n\t"     // mu * lambda_n, kept in lane 27 / 59 of the normal
        "v_readlane_b32 %[s0], %[RJ], 23\n\t"
        "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_readlane_b32 %[s2], %[RJ], 55\n\t"
        "v_mov_b32 %[xA], %[s0]\n\t"
        "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32 %[xB], %[s2]\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[t], %[t], %[t] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[t], %[t], %[t] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 %[s1], %[t], 31\n\t"
        "v_readlane_b32 %[s3], %[t], 63\n\t"
        "s_nop 1\n\t"
        "v_fma_f32 %[r2], %[s1], %[s1], %[EPS]\n\t"
        "v_fma_f32 %[r2], %[s3], %[s3], %[r2]\n\t"
        "v_rsq_f32 %[r2], %[r2]\n\t"
        "s_nop 1\n\t"
        "v_mul_f32_e64 %[r2], %[s4], %[r2] clamp\n\t"
        "v_fma_f32 %[xA], %[r2], -%[s1], -%[xA]\n\t"
        "v_fma_f32 %[xB], %[r2], -%[s3], -%[xB]\n\t"
        "v_cndmask_b32_e64 %[t], %[xB], %[xA], %[lowmask]\n\t"
        "v_mul_f32 %[r2], %[RM], %[t]\n\t"
        "v_mul_f32 %[c2], %[RM], %[t]\n\t"
        "v_fmac_f32 %[RJ], %[E], %[t]\n\t"
        "s_nop 0\n\t"
        "v_permlane32_swap_b32 %[r2], %[c2]\n\t"
        "s_nop 1\n\t"
        "v_add_f32 %[dv], %[dv], %[r2]\n\t"
        "v_add_f32 %[dv], %[dv], %[c2]\n\t"
        : [t] "=&v"(t), [xA] "=&v"(xA), [xB] "=&v"(xB), [r2] "=&v"(r2), [c2] "=&v"(c2),
          [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [RJ] "+v"(RJ), [dv] "+v"(dv)
        : [RM] "v"(RM), [RJnorm] "v"(RJnorm), [EPS] "v"(EPS), [E] "v"(E2355), [lowmask] "s"(lowmask),
          [NLn] "n"(NL + 4));
    if (RES) {
        asm volatile("v_mul_f32 %[x], %[dI], %[RJ]\n\t"
                     "v_max_f32 %[lsq], %[lsq], |%[x]|\n\t"
                     "s_nop 1"
                     : [x] "=&v"(xA), [lsq] "+v"(lsq)
                     : [dI] "v"(t), [RJ] "v"(RJ));
    } else {
        asm volatile("s_nop 1");
    }
}


// DPP sum over each 32-lane half of %[t]; results in lane 31 / 63.  A dependent DPP read needs
// 2 wait states after the VALU write (s_nop 1); the other wave of the SIMD issues into them.
#define SNK_REDUCE5                                                                               \
    "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
// delta-v += RM_lower * sL + RM_upper * sU in both halves.  X is a register that is only ever
// written under the DOF masks (lanes 0..21 and 24 of each half), so its other lanes stay zero
// and the folded lanes of delta-v (+1 / -1) are never touched.  Lane 24 / 56 of RM holds the
// row's denominator, so after the exchange X[24] / c2[24] hold dI * den of the lower / upper
// row for Bullet's residual test.
#define SNK_SCATTER(SL, SU)                                                                       \
    "s_mov_b64 exec, %[ldof]\n\t"                                                                  \
    "v_mul_f32 %[X], " SL ", %[RM]\n\t"                                                            \
    "s_mov_b64 exec, %[udof]\n\t"                                                                  \
    "v_mul_f32 %[X], " SU ", %[RM]\n\t"                                                            \
    "s_mov_b64 exec, -1\n\t"                                                                       \
    "v_mov_b32 %[c2], %[X]\n\t"                                                                    \
    "s_nop 0\n\t"                                                                                  \
    "v_permlane32_swap_b32 %[X], %[c2]\n\t"                                                        \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32 %[dv], %[dv], %[X]\n\t"                                                             \
    "v_add_f32 %[dv], %[dv], %[c2]\n\t"

// Two consecutive single rows living in the two halves of one slot (hand-written, 20 VALU):
// lower-half row first, then the upper-half row.  The reduction leaves each row's
// -(rhs - J.dv/den + a) in lane 31 / 63, exactly where RJ keeps that row's accumulated
// impulse a, so  a' = med3(-sum, LO, HI), dI = a' - a  are lane-local.  The upper row's sum
// first receives c * dI_lower (c in lane 63 of RM; lane 31 of RM is zero, so the lower lane
// recomputes the same dI).  gfx940-family hazards: VALU write -> v_readlane 1 wait state,
// v_readlane SGPR -> VALU read 2 wait states.
template <bool RES>
__device__ __forceinline__ void duo_step(float& RJ, const float RM, float& dv, float& X, float LO, float HI, float E3163,
                                         unsigned long long ldof, unsigned long long udof, float& lsq) {
    float t, dI, c2;
    float sL, sU;
    asm volatile(
        "v_mul_f32 %[t], %[RJ], %[dv]\n\t"
        "s_nop 1\n\t"
        SNK_REDUCE5
        "v_med3_f32 %[dI], -%[t], %[LO], %[HI]\n\t"
        "v_sub_f32 %[dI], %[dI], %[RJ]\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[sL], %[dI], 31\n\t"
        "s_nop 1\n\t"
        "v_fmac_f32 %[t], %[sL], %[RM]\n\t"
        "v_med3_f32 %[dI], -%[t], %[LO], %[HI]\n\t"
        "v_sub_f32 %[dI], %[dI], %[RJ]\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[sU], %[dI], 63\n\t"
        "v_fmac_f32 %[RJ], %[E], %[dI]\n\t"
        SNK_SCATTER("%[sL]", "%[sU]")
        : [t] "=&v"(t), [dI] "=&v"(dI), [c2] "=&v"(c2), [sL] "=&s"(sL), [sU] "=&s"(sU), [RJ] "+v"(RJ), [dv] "+v"(dv),
          [X] "+v"(X)
        : [RM] "v"(RM), [LO] "v"(LO), [HI] "v"(HI), [E] "v"(E3163), [ldof] "s"(ldof), [udof] "s"(udof));
    if (RES) {
        asm volatile("v_max3_f32 %[lsq], %[lsq], |%[X]|, |%[c2]|\n\t"
                     "s_nop 1"
                     : [lsq] "+v"(lsq)
                     : [X] "v"(X), [c2] "v"(c2));
    } else {
        asm volatile("s_nop 1");
    }
}

// Bullet's cone-friction pair (hand-written, 23 VALU): direction A in the lower half and B in
// the upper half of one register, so one reduction yields both sums; the pair of new
// accumulated impulses is projected radially onto the disc of radius lambda_n (the friction
// rows are built in units of mu, see load_slot) read from lane NL of the normal's RJ; dI is
// lane-local in lanes 31 / 63.
template <int NL, bool RES>
__device__ __forceinline__ void cone_step(float& RJ, const float RM, const float RJnorm, float& dv, float& X, float EPS,
                                          float E3163, unsigned long long ldof, unsigned long long udof, float& lsq) {
    float t, dI, r2, c2;
    float s1, s3, s4, sA, sB;
    asm volatile(
        "v_mul_f32 %[t], %[RJ], %[dv]\n\t"
        "v_readlane_b32 %[s4], %[RJnorm], %[NLn]\n\t"
        "s_nop 0\n\t"
        SNK_REDUCE5
        "s_nop 0\n\t"
        "v_readlane_b32 %[s1], %[t], 31\n\t"
        "v_readlane_b32 %[s3], %[t], 63\n\t"
        "s_nop 0\n\t"
        "v_fma_f32 %[r2], %[s1], %[s1], %[EPS]\n\t"
        "v_fma_f32 %[r2], %[s3], %[s3], %[r2]\n\t"
        "v_rsq_f32 %[r2], %[r2]\n\t"
        "s_nop 1\n\t"
        "v_mul_f32_e64 %[r2], %[s4], %[r2] clamp\n\t"
        "v_fma_f32 %[dI], -%[t], %[r2], -%[RJ]\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[sA], %[dI], 31\n\t"
        "v_readlane_b32 %[sB], %[dI], 63\n\t"
        "v_fmac_f32 %[RJ], %[E], %[dI]\n\t"
        SNK_SCATTER("%[sA]", "%[sB]")
        : [t] "=&v"(t), [dI] "=&v"(dI), [r2] "=&v"(r2), [c2] "=&v"(c2), [s1] "=&s"(s1), [s3] "=&s"(s3), [s4] "=&s"(s4),
          [sA] "=&s"(sA), [sB] "=&s"(sB), [RJ] "+v"(RJ), [dv] "+v"(dv), [X] "+v"(X)
        : [RM] "v"(RM), [RJnorm] "v"(RJnorm), [EPS] "v"(EPS), [E] "v"(E3163), [ldof] "s"(ldof), [udof] "s"(udof),
          [NLn] "n"(NL));
    if (RES) {
        asm volatile("v_max3_f32 %[lsq], %[lsq], |%[X]|, |%[c2]|\n\t"
                     "s_nop 1"
                     : [lsq] "+v"(lsq)
                     : [X] "v"(X), [c2] "v"(c2));
    } else {
        asm volatile("s_nop 1");
    }
}

// A motor row has a unit Jacobian (J = e_{6+j}), so its dot is just delta-v of that joint:
// every lane 6+j evaluates its own motor's candidate  dI = rhs - dv/den  (Bullet's
// deltaImpulse with cfm = 0) lane-locally, the row being resolved is picked with one
// v_readlane, and delta-v += M^-1[:, 6+j] * dI is one FMA with that scalar.  4 VALU (7 with a
// finite impulse clamp) instead of half a duo (12.5).  The accumulated impulse lives in lane
// 6+j of ACCV (updated under a one-lane exec mask).  Returns dI (wave-uniform).
template <int J, bool CLAMP>
__device__ __forceinline__ float motor_step(const float RMj, float& dv, const float RHSV, const float DINVV, float& ACCV,
                                            float NMI, float PMI) {
    float u, xs, s;
    if (CLAMP) {
        asm volatile(
            "v_fma_f32 %[u], -%[DINVV], %[dv], %[RHSV]\n\t"
            "v_add_f32 %[xs], %[ACCV], %[u]\n\t"
            "v_med3_f32 %[xs], %[xs], %[NMI], %[PMI]\n\t"
            "v_sub_f32 %[u], %[xs], %[ACCV]\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %[s], %[u], %[LN]\n\t"
            "s_mov_b64 exec, %[MASK]\n\t"
            "v_add_f32 %[ACCV], %[ACCV], %[u]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "v_fmac_f32 %[dv], %[s], %[RMj]\n\t"
            : [u] "=&v"(u), [xs] "=&v"(xs), [s] "=&s"(s), [ACCV] "+v"(ACCV), [dv] "+v"(dv)
            : [RMj] "v"(RMj), [RHSV] "v"(RHSV), [DINVV] "v"(DINVV), [NMI] "v"(NMI), [PMI] "v"(PMI), [LN] "n"(6 + J),
              [MASK] "n"(1 << (6 + J)));
    } else {
        asm volatile(
            "v_fma_f32 %[u], -%[DINVV], %[dv], %[RHSV]\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %[s], %[u], %[LN]\n\t"
            "s_mov_b64 exec, %[MASK]\n\t"
            "v_add_f32 %[ACCV], %[ACCV], %[u]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "v_fmac_f32 %[dv], %[s], %[RMj]\n\t"
            : [u] "=&v"(u), [s] "=&s"(s), [ACCV] "+v"(ACCV), [dv] "+v"(dv)
            : [RMj] "v"(RMj), [RHSV] "v"(RHSV), [DINVV] "v"(DINVV), [LN] "n"(6 + J), [MASK] "n"(1 << (6 + J)));
        (void)xs;
    }
    return s;
}


}
using namespace snk;
constexpr unsigned long long kLDof = 0x00000000013FFFFFull, kUDof = 0x013FFFFF00000000ull;
template <int MODE>
__global__ __launch_bounds__(64, 2) void k(float* out, unsigned long long* cyc, float seed) {
    const int lane = threadIdx.x, d = lane & 31;
    float RJ[8], RM[8];
#pragma unroll
    for (int s = 0; s < 8; s++) { RJ[s] = d < 22 ? seed * (s + 1) * 1e-3f * (d + 1) : 0.f; RM[s] = d < 22 ? 1e-4f * (d - s) : 0.f; }
    float dv = d == 22 ? 1.f : 0.f, X = 0.f, lsq = 0.f, ACCV = 0.f;
    const float E = (d == 23 || d == 31) ? 1.f : 0.f, ZERO = 0.f, BIG = 1e10f, EPS = 1e-30f;
    const unsigned long long LOWMASK = 0x00000000FFFFFFFFull;
    unsigned long long t0 = now();
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            if (MODE == 0) old_duo_step<0, false>(RJ[s], RM[s], dv, ZERO, BIG, E, LOWMASK, lsq);
            else if (MODE == 1) old_cone_step<23, false>(RJ[s], RM[s], RJ[(s + 1) & 7], dv, EPS, E, LOWMASK, lsq);
            else if (MODE == 2) duo_step<false>(RJ[s], RM[s], dv, X, ZERO, BIG, E, kLDof, kUDof, lsq);
            else if (MODE == 3) cone_step<31, false>(RJ[s], RM[s], RJ[(s + 1) & 7], dv, X, EPS, E, kLDof, kUDof, lsq);
            else if (MODE == 4) { motor_step<3, false>(RM[s], dv, RJ[s], RJ[(s + 1) & 7], ACCV, ZERO, BIG); }
#ifdef EXTRA_MODES
            EXTRA_MODES
#endif
        }
    }
    unsigned long long t1 = now();
    float acc = dv + X + lsq + ACCV;
#pragma unroll
    for (int s = 0; s < 8; s++) acc += RJ[s];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name) {
    for (int blocks : {1024, 2048}) {
        float* dd; unsigned long long* c;
        (void)hipMalloc(&dd, blocks * 64 * 4); (void)hipMalloc(&c, blocks * 8);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, dd, c, 0.3f);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, dd, c, 0.3f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), c, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        printf("%-28s waves/SIMD=%d  %7.1f ticks per step\n", name, blocks / 1024, avg / N_IT / 8);
        (void)hipFree(dd); (void)hipFree(c);
    }
}
int main() {
    run<0>("old duo"); run<1>("old cone"); run<2>("new duo"); run<3>("new cone"); run<4>("motor direct");
#ifdef EXTRA_RUNS
    EXTRA_RUNS
#endif
    return 0;
}
