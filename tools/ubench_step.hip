// Isolated timing of the solve loop's row steps (ticks per step, 1 and 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define N_IT 1000
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
namespace snk {
// DPP sum over each 32-lane half of %[t]; results in lane 31 / 63.  A dependent DPP read needs
// 2 wait states after the VALU write (s_nop 1); the other wave of the SIMD issues into them.
// Measured on MI355X (tools/ubench_lat.hip): a dependent v_add_f32_dpp step costs 12.6 clocks,
// a plain dependent VALU 5, v_readlane -> VALU use ~20, so the five reduction steps are the
// longest part of a row step; the scalar v_readlanes are placed in their shadow.
#define SNK_REDUCE_12 \
    "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define SNK_REDUCE_22 \
    "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define SNK_REDUCE_345                                                                            \
    "v_add_f32_dpp %[t], %[t], %[t] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
    "s_nop 1\n\t"                                                                                  \
    "v_add_f32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                   \
    "s_nop 1\n\t"

// Two consecutive single rows living in the two halves of one slot (hand-written, 25 VALU):
// lower-half row first, then the upper-half row.  One multiply and one reduction give both
// sums  -(rhs - J.dv/den + a)  in lanes 31 / 63; they, the accumulated impulses (lanes 31 / 63
// of RJ) and the coupling scalar c (lane 57 of RJ; delta-v is 0 there) are read into SGPRs and the clamps run on
// wave-uniform values:  a' = med3(-sum, LO, HI), dI = a' - a;  the upper row's sum first
// receives c * dI_lower.  v_cndmask gives every lane its own half's dI; the two contributions
// to delta-v cross halves with v_permlane32_swap.  RES: also max |dI * den| (den: lane 24 / 56
// of RM) into lsq.
#define SNK_DUO_HEAD                                     \
    "v_mul_f32 %[t], %[RJ], %[dv]\n\t"                    \
    "v_readlane_b32 %[s2], %[RJ], 31\n\t"                 \
    "v_readlane_b32 %[s3], %[RJ], 63\n\t"                 \
    SNK_REDUCE_12                                          \
    "v_readlane_b32 %[s4], %[RJ], 57\n\t"                 \
    "s_nop 0\n\t"                                         \
    SNK_REDUCE_22                                          \
    "s_nop 1\n\t"                                         \
    SNK_REDUCE_345                                         \
    "v_readlane_b32 %[s0], %[t], 31\n\t"                  \
    "v_readlane_b32 %[s1], %[t], 63\n\t"                  \
    "s_nop 0\n\t"
#define SNK_DUO_TAIL                                     \
    "v_cndmask_b32_e64 %[t], %[dS], %[dF], %[lowmask]\n\t" \
    "v_mul_f32 %[x], %[RM], %[t]\n\t"                     \
    "v_mul_f32 %[c2], %[RM], %[t]\n\t"                    \
    "v_fmac_f32 %[RJ], %[E], %[t]\n\t"                    \
    "s_nop 0\n\t"                                         \
    "v_permlane32_swap_b32 %[x], %[c2]\n\t"               \
    "s_nop 1\n\t"                                         \
    "v_add_f32 %[dv], %[dv], %[x]\n\t"                    \
    "v_add_f32 %[dv], %[dv], %[c2]\n\t"
// BOX = false: contact normals, bounds [0, inf) -> a single v_max; BOX = true: [-HI, HI]
template <bool RES, bool BOX>
__device__ __forceinline__ void duo_step(float& RJ, const float RM, float& dv, float HI, float E3163,
                                         unsigned long long lowmask, float& lsq) {
    float t, x, dF, dS, c2;
    float s0, s1, s2, s3, s4;
    if (!BOX) {
        asm volatile(
            SNK_DUO_HEAD
            "v_max_f32_e64 %[dF], -%[s0], 0\n\t"
            "v_mov_b32 %[x], %[s1]\n\t"
            "v_subrev_f32 %[dF], %[s2], %[dF]\n\t"
            "v_fmac_f32 %[x], %[s4], %[dF]\n\t"
            "v_max_f32_e64 %[x], -%[x], 0\n\t"
            "v_subrev_f32 %[dS], %[s3], %[x]\n\t"
            SNK_DUO_TAIL
            : [t] "=&v"(t), [x] "=&v"(x), [dF] "=&v"(dF), [dS] "=&v"(dS), [c2] "=&v"(c2), [s0] "=&s"(s0), [s1] "=&s"(s1),
              [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [RJ] "+v"(RJ), [dv] "+v"(dv)
            : [RM] "v"(RM), [E] "v"(E3163), [lowmask] "s"(lowmask));
    } else
    asm volatile(
        SNK_DUO_HEAD
        "v_med3_f32 %[dF], -%[s0], -%[HI], %[HI]\n\t"
        "v_mov_b32 %[x], %[s1]\n\t"
        "v_subrev_f32 %[dF], %[s2], %[dF]\n\t"
        "v_fmac_f32 %[x], %[s4], %[dF]\n\t"
        "v_med3_f32 %[x], -%[x], -%[HI], %[HI]\n\t"
        "v_subrev_f32 %[dS], %[s3], %[x]\n\t"
        SNK_DUO_TAIL
        : [t] "=&v"(t), [x] "=&v"(x), [dF] "=&v"(dF), [dS] "=&v"(dS), [c2] "=&v"(c2), [s0] "=&s"(s0), [s1] "=&s"(s1),
          [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [RJ] "+v"(RJ), [dv] "+v"(dv)
        : [RM] "v"(RM), [HI] "v"(HI), [E] "v"(E3163), [lowmask] "s"(lowmask));
    if (RES) {
        asm volatile("v_max3_f32 %[lsq], %[lsq], |%[x]|, |%[c2]|\n\t"
                     "s_nop 1"
                     : [lsq] "+v"(lsq)
                     : [x] "v"(x), [c2] "v"(c2));
    } else {
        asm volatile("s_nop 1");
    }
}

// Bullet's cone-friction pair (hand-written, 27 VALU): direction A in the lower half and B in
// the upper half of one register, so one reduction yields both sums; the pair of new
// accumulated impulses is projected radially onto the disc of radius lambda_n (the friction
// rows are built in units of mu, see load_slot), read from lane NL of the normal's RJ.
template <int NL, bool RES>
__device__ __forceinline__ void cone_step(float& RJ, const float RM, const float RJnorm, float& dv, float EPS,
                                          float E3163, unsigned long long lowmask, float& lsq) {
    float t, xA, xB, r2, c2;
    float s0, s1, s2, s3, s4;
    asm volatile(
        "v_mul_f32 %[t], %[RJ], %[dv]\n\t"
        "v_readlane_b32 %[s4], %[RJnorm], %[NLn]\n\t"
        "v_readlane_b32 %[s0], %[RJ], 31\n\t"
        SNK_REDUCE_12
        "v_readlane_b32 %[s2], %[RJ], 63\n\t"
        "v_mov_b32 %[xA], %[s0]\n\t"
        SNK_REDUCE_22
        "v_mov_b32 %[xB], %[s2]\n\t"
        "s_nop 0\n\t"
        SNK_REDUCE_345
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
        : [RM] "v"(RM), [RJnorm] "v"(RJnorm), [EPS] "v"(EPS), [E] "v"(E3163), [lowmask] "s"(lowmask),
          [NLn] "n"(NL));
    if (RES) {
        asm volatile("v_max3_f32 %[lsq], %[lsq], |%[r2]|, |%[c2]|\n\t"
                     "s_nop 1"
                     : [lsq] "+v"(lsq)
                     : [r2] "v"(r2), [c2] "v"(c2));
    } else {
        asm volatile("s_nop 1");
    }
}

// A motor row has a unit Jacobian (J = e_{6+j}), so its dot is just delta-v of that joint, and
// with the row written in units of 1/den (impulse variable y = dI * den; RMm = M^-1[:, 6+j] / den)
// every lane 6+j evaluates its own motor's candidate  y = target - dv  (Bullet's
// deltaImpulse = rhs - dv/den with cfm = 0, times den) lane-locally; the row being resolved
// is picked with one v_readlane and delta-v += RMm * y is one FMA with that scalar.  4 VALU
// (7 with a finite impulse clamp, bound max_impulse * den per lane in PMIV) instead of half a
// duo (12.5).  The accumulated y lives in lane 6+j of ACCV (updated under a one-lane exec
// mask).  Returns y (wave-uniform).
template <int J, bool CLAMP>
__device__ __forceinline__ float motor_step(const float RMj, float& dv, const float TARGV, float& ACCV, float PMIV) {
    float u, xs, s;
    if (CLAMP) {
        asm volatile(
            "v_sub_f32 %[u], %[TARGV], %[dv]\n\t"
            "v_add_f32 %[xs], %[ACCV], %[u]\n\t"
            "v_med3_f32 %[xs], %[xs], -%[PMIV], %[PMIV]\n\t"
            "v_sub_f32 %[u], %[xs], %[ACCV]\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %[s], %[u], %[LN]\n\t"
            "s_mov_b64 exec, %[MASK]\n\t"
            "v_add_f32 %[ACCV], %[ACCV], %[u]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "v_fmac_f32 %[dv], %[s], %[RMj]\n\t"
            : [u] "=&v"(u), [xs] "=&v"(xs), [s] "=&s"(s), [ACCV] "+v"(ACCV), [dv] "+v"(dv)
            : [RMj] "v"(RMj), [TARGV] "v"(TARGV), [PMIV] "v"(PMIV), [LN] "n"(6 + J), [MASK] "n"(1 << (6 + J)));
    } else {
        asm volatile(
            "v_sub_f32 %[u], %[TARGV], %[dv]\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %[s], %[u], %[LN]\n\t"
            "s_mov_b64 exec, %[MASK]\n\t"
            "v_add_f32 %[ACCV], %[ACCV], %[u]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "v_fmac_f32 %[dv], %[s], %[RMj]\n\t"
            : [u] "=&v"(u), [s] "=&s"(s), [ACCV] "+v"(ACCV), [dv] "+v"(dv)
            : [RMj] "v"(RMj), [TARGV] "v"(TARGV), [LN] "n"(6 + J), [MASK] "n"(1 << (6 + J)));
        (void)xs;
    }
    return s;
}


// ---- experimental variants ----
// normal duo that also records "new accumulated impulse == 0" of its two rows in bit BIT / BIT+1 of z
template <int BIT>
__device__ __forceinline__ void duo_z(float& RJ, const float RM, float& dv, float E3163, unsigned long long lowmask, float& lsq,
                                      unsigned& z) {
    float t, x, dF, dS, c2;
    float s0, s1, s2, s3, s4;
    unsigned tz;
    asm volatile(
        SNK_DUO_HEAD
        "v_max_f32_e64 %[dF], -%[s0], 0\n\t"
        "v_mov_b32 %[x], %[s1]\n\t"
        "v_cmp_eq_f32 vcc, 0, %[dF]\n\t"
        "v_subrev_f32 %[dF], %[s2], %[dF]\n\t"
        "v_fmac_f32 %[x], %[s4], %[dF]\n\t"
        "s_and_b32 %[tz], vcc_lo, %[BL]\n\t"
        "v_max_f32_e64 %[x], -%[x], 0\n\t"
        "s_or_b32 %[z], %[z], %[tz]\n\t"
        "v_cmp_eq_f32 vcc, 0, %[x]\n\t"
        "v_subrev_f32 %[dS], %[s3], %[x]\n\t"
        "v_cndmask_b32_e64 %[t], %[dS], %[dF], %[lowmask]\n\t"
        "v_mul_f32 %[x], %[RM], %[t]\n\t"
        "s_and_b32 %[tz], vcc_lo, %[BU]\n\t"
        "v_mul_f32 %[c2], %[RM], %[t]\n\t"
        "s_or_b32 %[z], %[z], %[tz]\n\t"
        "v_fmac_f32 %[RJ], %[E], %[t]\n\t"
        "s_nop 0\n\t"
        "v_permlane32_swap_b32 %[x], %[c2]\n\t"
        "v_add_f32 %[dv], %[dv], %[x]\n\t"
        "v_add_f32 %[dv], %[dv], %[c2]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[x]|, |%[c2]|\n\t"
        : [t] "=&v"(t), [x] "=&v"(x), [dF] "=&v"(dF), [dS] "=&v"(dS), [c2] "=&v"(c2), [s0] "=&s"(s0), [s1] "=&s"(s1),
          [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [tz] "=&s"(tz), [z] "+s"(z), [RJ] "+v"(RJ), [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [RM] "v"(RM), [E] "v"(E3163), [lowmask] "s"(lowmask), [BL] "n"(1u << BIT), [BU] "n"(1u << (BIT + 1))
        : "vcc", "scc");
}
// cone that is skipped when bit BIT of zb is set
template <int NL, int BIT>
__device__ __forceinline__ void cone_z(float& RJ, const float RM, const float RJnorm, float& dv, float EPS,
                                       float E3163, unsigned long long lowmask, float& lsq, unsigned zb) {
    float t, xA, xB, r2, c2;
    float s0, s1, s2, s3, s4;
    asm volatile(
        "s_bitcmp1_b32 %[zb], %[BIT]\n\t"
        "s_cbranch_scc1 .Lskip%=\n\t"
        "v_mul_f32 %[t], %[RJ], %[dv]\n\t"
        "v_readlane_b32 %[s4], %[RJnorm], %[NLn]\n\t"
        "v_readlane_b32 %[s0], %[RJ], 31\n\t"
        SNK_REDUCE_12
        "v_readlane_b32 %[s2], %[RJ], 63\n\t"
        "v_mov_b32 %[xA], %[s0]\n\t"
        SNK_REDUCE_22
        "v_mov_b32 %[xB], %[s2]\n\t"
        "s_nop 0\n\t"
        SNK_REDUCE_345
        "v_readlane_b32 %[s1], %[t], 31\n\t"
        "v_readlane_b32 %[s3], %[t], 63\n\t"
        "s_nop 0\n\t"
        "v_fma_f32 %[r2], %[s1], %[s1], %[EPS]\n\t"
        "v_fma_f32 %[r2], %[s3], %[s3], %[r2]\n\t"
        "v_rsq_f32 %[r2], %[r2]\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_e64 %[r2], %[s4], %[r2] clamp\n\t"
        "v_fma_f32 %[xA], %[r2], -%[s1], -%[xA]\n\t"
        "v_fma_f32 %[xB], %[r2], -%[s3], -%[xB]\n\t"
        "v_cndmask_b32_e64 %[t], %[xB], %[xA], %[lowmask]\n\t"
        "v_mul_f32 %[r2], %[RM], %[t]\n\t"
        "v_mul_f32 %[c2], %[RM], %[t]\n\t"
        "v_fmac_f32 %[RJ], %[E], %[t]\n\t"
        "s_nop 0\n\t"
        "v_permlane32_swap_b32 %[r2], %[c2]\n\t"
        "v_add_f32 %[dv], %[dv], %[r2]\n\t"
        "v_add_f32 %[dv], %[dv], %[c2]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[r2]|, |%[c2]|\n\t"
        ".Lskip%=:\n\t"
        : [t] "=&v"(t), [xA] "=&v"(xA), [xB] "=&v"(xB), [r2] "=&v"(r2), [c2] "=&v"(c2),
          [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [RJ] "+v"(RJ), [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [RM] "v"(RM), [RJnorm] "v"(RJnorm), [EPS] "v"(EPS), [E] "v"(E3163), [lowmask] "s"(lowmask), [zb] "s"(zb),
          [NLn] "n"(NL), [BIT] "n"(BIT)
        : "scc");
}
#define SNK_RED64(T)                                                                                   \
    "v_add_f32_dpp " T ", " T ", " T " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  \
    "s_nop 1\n\t"                                                                                      \
    "v_add_f32_dpp " T ", " T ", " T " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  \
    "s_nop 1\n\t"                                                                                      \
    "v_add_f32_dpp " T ", " T ", " T " row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "s_nop 1\n\t"                                                                                      \
    "v_add_f32_dpp " T ", " T ", " T " row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "s_nop 1\n\t"                                                                                      \
    "v_add_f32_dpp " T ", " T ", " T " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                    \
    "s_nop 1\n\t"                                                                                      \
    "v_add_f32_dpp " T ", " T ", " T " row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
// one DPP step of two independent reductions: each instruction is the other's wait state
#define SNK_RED64x2_STEP(A, B, MODE)                    \
    "v_add_f32_dpp " A ", " A ", " A " " MODE "\n\t"      \
    "v_add_f32_dpp " B ", " B ", " B " " MODE "\n\t"      \
    "s_nop 0\n\t"

// a contact-normal row: a' = max(a + rhs - (J/den).dv, 0); dv += M^-1 J^T (a' - a).  Returns a'.
__device__ __forceinline__ float row_step_normal(float jv, float mv, float rhs, float acc, float den, float& dv, float& lsq) {
    float t, x, dI, P, s;
    asm volatile(
        "v_mul_f32 %[t], %[jv], %[dv]\n\t"
        "v_add_f32 %[x], %[acc], %[rhs]\n\t"
        "s_nop 0\n\t"
        SNK_RED64("%[t]")
        "s_nop 0\n\t"
        "v_readlane_b32 %[s], %[t], 63\n\t"
        "s_nop 1\n\t"
        "v_subrev_f32 %[x], %[s], %[x]\n\t"
        "v_max_f32 %[x], 0, %[x]\n\t"
        "v_sub_f32 %[dI], %[x], %[acc]\n\t"
        "v_mul_f32 %[P], %[dI], %[mv]\n\t"
        "v_mul_f32 %[t], %[dI], %[den]\n\t"
        "v_add_f32 %[dv], %[dv], %[P]\n\t"
        "v_max_f32 %[lsq], %[lsq], |%[t]|\n\t"
        : [t] "=&v"(t), [x] "=&v"(x), [dI] "=&v"(dI), [P] "=&v"(P), [s] "=&s"(s), [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [jv] "v"(jv), [mv] "v"(mv), [rhs] "v"(rhs), [acc] "v"(acc), [den] "v"(den));
    return x;
}

// Bullet's cone-friction pair of one contact: both dots from the same delta-v, the new pair
// (a + rhs - dot) projected radially onto the disc of radius lim.
__device__ __forceinline__ void row_step_cone(float jA, float mA, float jB, float mB, float rhsA, float rhsB, float& accA,
                                              float& accB, float denA, float denB, float lim, float EPS, float& dv,
                                              float& lsq) {
    float tA, tB, xA, xB, r2, P, sA, sB;
    asm volatile(
        "v_mul_f32 %[tA], %[jA], %[dv]\n\t"
        "v_mul_f32 %[tB], %[jB], %[dv]\n\t"
        "v_add_f32 %[xA], %[accA], %[rhsA]\n\t"
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        "v_add_f32_dpp %[tA], %[tA], %[tA] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %[tB], %[tB], %[tB] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32 %[xB], %[accB], %[rhsB]\n\t"
        "v_readlane_b32 %[sA], %[tA], 63\n\t"
        "v_readlane_b32 %[sB], %[tB], 63\n\t"
        "s_nop 0\n\t"
        "v_subrev_f32 %[xA], %[sA], %[xA]\n\t"
        "v_subrev_f32 %[xB], %[sB], %[xB]\n\t"
        "v_fma_f32 %[r2], %[xA], %[xA], %[EPS]\n\t"
        "v_fma_f32 %[r2], %[xB], %[xB], %[r2]\n\t"
        "v_rsq_f32 %[r2], %[r2]\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_e64 %[r2], %[lim], %[r2] clamp\n\t"
        "v_mul_f32 %[xA], %[xA], %[r2]\n\t"
        "v_mul_f32 %[xB], %[xB], %[r2]\n\t"
        "v_sub_f32 %[tA], %[xA], %[accA]\n\t"
        "v_sub_f32 %[tB], %[xB], %[accB]\n\t"
        "v_mul_f32 %[P], %[tA], %[mA]\n\t"
        "v_mul_f32 %[r2], %[tA], %[denA]\n\t"
        "v_fmac_f32 %[P], %[tB], %[mB]\n\t"
        "v_mul_f32 %[tB], %[tB], %[denB]\n\t"
        "v_add_f32 %[dv], %[dv], %[P]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[r2]|, |%[tB]|\n\t"
        : [tA] "=&v"(tA), [tB] "=&v"(tB), [xA] "=&v"(xA), [xB] "=&v"(xB), [r2] "=&v"(r2), [P] "=&v"(P), [sA] "=&s"(sA),
          [sB] "=&s"(sB), [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [jA] "v"(jA), [mA] "v"(mA), [jB] "v"(jB), [mB] "v"(mB), [rhsA] "v"(rhsA), [rhsB] "v"(rhsB), [accA] "v"(accA),
          [accB] "v"(accB), [denA] "v"(denA), [denB] "v"(denB), [lim] "v"(lim), [EPS] "v"(EPS));
    accA = xA;
    accB = xB;
}

// ---- prototype: block-wise Gauss-Seidel with in-block residual updates ----
// One friction pair (rows 2P, 2P+1 of a 16-row block whose running sums sit in lanes 0..15 of `sv`, their
// accumulated impulses in lanes 0..15 of `av`): no cross-lane reduction -- the sums are read with v_readlane,
// the pair is projected onto its friction disc, and the block's other sums are brought up to date with one FMA per
// row (coupling columns AcA / AcB, lane = row; the diagonal carries the -1 of the accumulated impulse).  delta-v
// is only accumulated (dv2 += RM * dI, A in the lower half, B in the upper; combined once per block).
template <int P>
__device__ __forceinline__ void cone_block_step(float& sv, float& av, const float AcA, const float AcB, const float RM,
                                                float& dv2, float lim, float EPS, unsigned long long lowmask, float& lsq) {
    float r2, xA, xB, t;
    float s1, s3, a1, a3;
    asm volatile(
        "s_nop 0\n\t"
        "v_readlane_b32 %[s1], %[sv], %[LA]\n\t"
        "v_readlane_b32 %[s3], %[sv], %[LB]\n\t"
        "v_readlane_b32 %[a1], %[sv], %[LA2]\n\t"
        "v_readlane_b32 %[a3], %[sv], %[LB2]\n\t"
        "s_nop 0\n\t"
        "v_mov_b32 %[xA], %[a1]\n\t"
        "v_mov_b32 %[xB], %[a3]\n\t"
        "v_fma_f32 %[r2], %[s1], %[s1], %[EPS]\n\t"
        "v_fma_f32 %[r2], %[s3], %[s3], %[r2]\n\t"
        "v_rsq_f32 %[r2], %[r2]\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_e64 %[r2], %[lim], %[r2] clamp\n\t"
        "v_fma_f32 %[xA], %[r2], -%[s1], -%[xA]\n\t"      // dI_A = new - old
        "v_fma_f32 %[xB], %[r2], -%[s3], -%[xB]\n\t"
        "v_fmac_f32 %[sv], %[AcA], %[xA]\n\t"             // the block's remaining sums (and this row's own, through the diagonal)
        "v_fmac_f32 %[sv], %[AcB], %[xB]\n\t"
        "v_cndmask_b32_e64 %[t], %[xB], %[xA], %[lowmask]\n\t"
        "v_fmac_f32 %[dv2], %[RM], %[t]\n\t"
        : [r2] "=&v"(r2), [xA] "=&v"(xA), [xB] "=&v"(xB), [t] "=&v"(t), [s1] "=&s"(s1), [s3] "=&s"(s3), [a1] "=&s"(a1),
          [a3] "=&s"(a3), [sv] "+v"(sv), [dv2] "+v"(dv2)
        : [av] "v"(av), [AcA] "v"(AcA), [AcB] "v"(AcB), [RM] "v"(RM), [lim] "v"(lim), [EPS] "v"(EPS), [lowmask] "s"(lowmask),
          [LA] "n"(2 * P), [LB] "n"(2 * P + 1), [LA2] "n"(16 + 2 * P), [LB2] "n"(17 + 2 * P));
    // the accumulated impulses live in lanes 16..31 of `sv` (the coupling columns carry a 1 there), so the FMAs
    // above have already updated them; only the residual is left to track
    asm volatile("v_max3_f32 %[lsq], %[lsq], |%[xA]|, |%[xB]|" : [lsq] "+v"(lsq) : [xA] "v"(xA), [xB] "v"(xB));
}
// per block: the 16 running sums from scratch (lane = row): s = sum_k Jc[k] * dv[k], dv[k] broadcast by v_readlane
template <int K>
__device__ __forceinline__ void block_dots(float& sv, const float (&Jc)[22], const float dv) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) {
        float d;
        asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(d) : "v"(dv), "n"(k));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(d), "v"(Jc[k]));
    }
    sv = acc;
}


}
using namespace snk;
constexpr unsigned long long kLowMask = 0x00000000FFFFFFFFull;
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float seed, float accinit) {
    const int lane = threadIdx.x, d = lane & 31;
    float RJ[8], RM[8];
#pragma unroll
    for (int s = 0; s < 8; s++) { RJ[s] = d < 22 ? seed * (s + 1) * 1e-3f * (d + 1) : (d == 31 ? accinit : 0.f); RM[s] = d < 22 ? 1e-4f * (d - s) : 0.f; }
    float dv = d == 22 ? 1.f : 0.f, lsq = 0.f, ACCV = 0.f;
    float sv = 0.f, av = 0.f, dv2 = 0.f, Jc[22];
#pragma unroll
    for (int k = 0; k < 22; k++) Jc[k] = 1e-3f * (k + 1) * (lane + 1) * seed;
    unsigned zz = 0, zb = __builtin_amdgcn_readfirstlane(accinit == 0.f ? 0xffffffffu : 0u);
    const float E = (d == 31) ? 1.f : 0.f, EPS = 1e-30f;
    unsigned long long t0 = now();
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            if (MODE == 0) duo_step<true, false>(RJ[s], RM[s], dv, 0.f, E, kLowMask, lsq);
            else if (MODE == 1) cone_step<31, true>(RJ[s], RM[s], RJ[(s + 1) & 7], dv, EPS, E, kLowMask, lsq);
            else if (MODE == 2) duo_z<4>(RJ[s], RM[s], dv, E, kLowMask, lsq, zz);
            else if (MODE == 3) cone_z<31, 5>(RJ[s], RM[s], RJ[(s + 1) & 7], dv, EPS, E, kLowMask, lsq, zb);
            else if (MODE == 4) { motor_step<3, false>(RM[s], dv, RJ[s], ACCV, 0.f); }
            else if (MODE == 6) { RJ[s] = row_step_normal(RM[s], RM[(s + 1) & 7], 0.1f, RJ[s], 1.0f, dv, lsq); }
            else if (MODE == 8) {
                // one 16-row block = 8 friction pairs: dots once, then 8 in-block steps (s == 0 starts a block)
                if (s == 0) {
                    float tmp = dv2, sw = dv2;
                    asm volatile("s_nop 0\n\tv_permlane32_swap_b32 %0, %1\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %2, %2, %1" : "+v"(tmp), "+v"(sw), "+v"(dv));
                    dv2 = 0.f;
                    block_dots<22>(sv, Jc, dv);
                }
                switch (s) {
                    case 0: cone_block_step<0>(sv, av, RJ[0], RJ[1], RM[0], dv2, 0.5f, EPS, kLowMask, lsq); break;
                    case 1: cone_block_step<1>(sv, av, RJ[2], RJ[3], RM[1], dv2, 0.5f, EPS, kLowMask, lsq); break;
                    case 2: cone_block_step<2>(sv, av, RJ[4], RJ[5], RM[2], dv2, 0.5f, EPS, kLowMask, lsq); break;
                    case 3: cone_block_step<3>(sv, av, RJ[6], RJ[7], RM[3], dv2, 0.5f, EPS, kLowMask, lsq); break;
                    case 4: cone_block_step<4>(sv, av, RJ[1], RJ[0], RM[4], dv2, 0.5f, EPS, kLowMask, lsq); break;
                    case 5: cone_block_step<5>(sv, av, RJ[3], RJ[2], RM[5], dv2, 0.5f, EPS, kLowMask, lsq); break;
                    case 6: cone_block_step<6>(sv, av, RJ[5], RJ[4], RM[6], dv2, 0.5f, EPS, kLowMask, lsq); break;
                    default: cone_block_step<7>(sv, av, RJ[7], RJ[6], RM[7], dv2, 0.5f, EPS, kLowMask, lsq); break;
                }
            }
            else if (MODE == 7) { float aA = RJ[s], aB = RJ[(s + 1) & 7]; row_step_cone(RM[s], RM[(s+2)&7], RM[(s+1)&7], RM[(s+3)&7], 0.1f, 0.2f, aA, aB, 1.f, 1.f, 0.5f, EPS, dv, lsq); RJ[s] = aA; RJ[(s + 1) & 7] = aB; }
        }
    }
    unsigned long long t1 = now();
    float acc = dv + lsq + ACCV + (float)zz + sv + av + dv2;
#pragma unroll
    for (int s = 0; s < 8; s++) acc += RJ[s];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, float accinit = 0.01f) {
    for (int blocks : {1024, 2048}) {
        float* dd; unsigned long long* c;
        (void)hipMalloc(&dd, blocks * 64 * 4); (void)hipMalloc(&c, blocks * 8);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, dd, c, 0.3f, accinit);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, dd, c, 0.3f, accinit);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), c, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        printf("%-34s waves/SIMD=%d  %7.1f ticks per step\n", name, blocks / 1024, avg / N_IT / 8);
        (void)hipFree(dd); (void)hipFree(c);
    }
}
int main(int argc, char** argv) {
    int m = argc > 1 ? atoi(argv[1]) : -1;
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (m < 0 || m == 0) run<0>("duo (normals)");
    if (m < 0 || m == 2) run<2>("duo + zero flags");
    if (m < 0 || m == 1) run<1>("cone");
    if (m < 0 || m == 3) run<3>("cone + skip bit (active)");
    if (m < 0 || m == 5) run<3>("cone + skip bit (all inert)", 0.0f);
    if (m < 0 || m == 4) run<4>("motor direct");
    if (m < 0 || m == 6) run<6>("32-link normal row (64-lane)");
    if (m < 0 || m == 7) run<7>("32-link cone pair (64-lane)");
    if (m < 0 || m == 8) run<8>("block-wise cone pair (prototype)");
    return 0;
}
