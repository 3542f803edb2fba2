// snk_pgs_v2.hpp -- register-resident constraint solve for the 16-link chain (ND = 22).
//
// The projected Gauss-Seidel sweep is a strictly sequential chain over ~208 rows x 50
// iterations per physics substep; in v1 every row step paid several LDS round trips
// (~500 cycles).  Here ALL rows stay in VGPRs for the whole solve:
//
//   * one 64-lane register holds TWO rows, one per 32-lane half; lane = (half, d), d = lane&31
//       d <  22 : component d of the row vector
//       d == 22 : -rhs impulse   (RJ only)          delta-v register holds +1 there
//       d == 23 : accumulated impulse a (RJ only)   delta-v register holds -1 there
//       d == 24 : row denominator J M^-1 J^T (RJ only, for the residual)
//     RJ = J / den (pre-scaled), RM = M^-1 J^T, both halves of `dv` carry the same delta-v.
//     Then   sum_d RJ[d] dv[d] = (J.dv)/den - rhs - a   and the new accumulated impulse of
//     the row is simply clamp(-sum): no per-row scalar is fetched from memory.
//   * slots: 0..7 motors (motors 2s, 2s+1 in the lower/upper half), 8..39 contact normals
//     (contacts 2s, 2s+1), 40..103 friction pairs (direction A lower, B upper)
//   * a half's dot: 4 DPP row steps + row_bcast:15 -> lane 31 / 63 -> v_readlane; one
//     reduction therefore yields the dots of BOTH rows of a slot
//   * friction pair: Bullet resolves both directions from the same velocity, so the two
//     dots are used as they are.  Two consecutive single rows (a "duo"): the second row must
//     see the first row's update; its dot is corrected exactly by  c * dI_first  with the
//     coupling scalar c = RJ_second . RM_first, precomputed once per substep (lane 25 / 57).
//     This halves the number of sequential steps of the Gauss-Seidel chain at equal work.
//   * contributions to delta-v cross halves with v_permlane32_swap, so both halves always
//     hold the complete delta-v
//   * the solve is VALU-issue-bound (PMC: VALU active 48 % of wave cycles per wave, two waves
//     per SIMD; I-cache hit rate 99.95 %), so the row updates are hand-written with the fewest
//     VALU instructions: 12 per single row, 28 per friction pair; s_nops for the DPP hazards
//     cost nothing because the other wave of the SIMD issues into them.  (A software-
//     pipelined variant with look-ahead reductions was measured: +25 % instructions, slower.)
//   * the early-exit residual (max over rows of |dI * den|) is evaluated only until the first
//     row exceeds the threshold in an iteration; after that the residual-free code runs
//
// Row construction (M^-1 J^T by ABA delta sweeps, one row per lane) goes through a 64-row
// LDS staging buffer, one batch per row kind; J itself is evaluated directly in the
// (half, d) layout from the contact point and the per-lane joint axis/origin.
//
// Restates the same Bullet steps as v1's build_rows_v1/pgs_v1 (snk_device.hpp); the order
// of row updates is identical, so both versions track the oracle.
#pragma once

namespace snk {

struct swap2 {
    float a, b;
};
// v_permlane32_swap: returns a = [x.lo, y.lo], b = [x.hi, y.hi] (32-lane halves)
__device__ __forceinline__ swap2 half_swap(float x, float y) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    swap2 o;
    o.a = __uint_as_float(r[0]);
    o.b = __uint_as_float(r[1]);
    return o;
}
// sum over each 32-lane half; result valid in lane 31 (lower half) and lane 63 (upper half)
__device__ __forceinline__ float half_reduce(float t) {
    t = dpp_add<0xB1, 0xf>(t);
    t = dpp_add<0x4E, 0xf>(t);
    t = dpp_add<0x114, 0xf>(t);
    t = dpp_add<0x118, 0xf>(t);
    t = dpp_add<0x142, 0xa>(t);
    return t;
}
__device__ __forceinline__ float rdlane(float x, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
// Hand-scheduled half reductions for the solve loop.  hipcc splits the first step (after a
// multiply) and the masked row_bcast step into v_mov_dpp + v_add; written out, each is one
// v_add_f32_dpp.  A dependent DPP read needs 2 wait states after the VALU write (s_nop 1);
// with two waves per SIMD the other wave issues into those slots.
#define SNK_HALF_REDUCE_ASM                                                                  \
    "s_nop 1\n\t"                                                                            \
    "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "s_nop 1\n\t"                                                                            \
    "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "s_nop 1\n\t"                                                                            \
    "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
    "s_nop 1\n\t"                                                                            \
    "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
    "s_nop 1\n\t"                                                                            \
    "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                    \
    "s_nop 1\n\t"
// sum of the 32-lane half H of x, as a wave-uniform (SGPR) value
template <int H>
__device__ __forceinline__ float half_dot1(float x) {
    float s;
    if (H)
        asm volatile(SNK_HALF_REDUCE_ASM "v_readlane_b32 %1, %0, 63" : "+v"(x), "=s"(s));
    else
        asm volatile(SNK_HALF_REDUCE_ASM "v_readlane_b32 %1, %0, 31" : "+v"(x), "=s"(s));
    return s;
}
// both half sums at once
__device__ __forceinline__ void half_dot2(float x, float& sA, float& sB) {
    asm volatile(SNK_HALF_REDUCE_ASM "v_readlane_b32 %1, %0, 31\n\tv_readlane_b32 %2, %0, 63"
                 : "+v"(x), "=s"(sA), "=s"(sB));
}

// ------------------------------------------------------------------------------------
// contacts of the current pose, written at their COMPACT index (same geometry as v1)
// ------------------------------------------------------------------------------------
template <class LT>
__device__ int find_contacts_v2(LT& L, const DevModel& M, int lane) {
    constexpr int N = LT::kN;
    static_assert(4 * N == 64, "one contact slot per lane");
    const int slot = lane;
    const int c = slot >> 1;
    const int b = (c + 1) >> 1;
    const float* Rb = L.R[b];
    const float* Rc = M.cyl_R[c];
    float Rw[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            Rw[3 * i + j] = Rb[3 * i] * Rc[j] + Rb[3 * i + 1] * Rc[3 + j] + Rb[3 * i + 2] * Rc[6 + j];
    f3 dl = mk3(-Rw[6], -Rw[7], -Rw[8]);
    float rr = sqrtf(dl.x * dl.x + dl.y * dl.y);
    float lx = 0.f, ly = 0.f;
    if (rr > 1e-12f) { lx = M.cyl_r * dl.x / rr; ly = M.cyl_r * dl.y / rr; }
    float lz = (slot & 1) ? M.cyl_hl : -M.cyl_hl;
    f3 loc = mk3(lx + M.margin * dl.x, ly + M.margin * dl.y, lz + M.margin * dl.z);
    f3 P = ld3(L.o[b]) + mulRv(Rb, ld3(M.cyl_c[c])) + mulRv(Rw, loc);
    const float dist = P.z;
    const bool active = dist < M.break_thr;
    unsigned long long bal = __ballot(active);
    if (active) {
        const int idx = __popcll(bal & ((1ull << lane) - 1ull));
        st3(L.ccP[idx], P);
        L.ccdist[idx] = dist;
        L.ccbody[idx] = b;
        f3 a = mk3(M.aniso[0], M.aniso[1], M.aniso[2]);
        f3 l1 = mulRtv(Rw, mk3(0.f, -1.f, 0.f));
        f3 l2 = mulRtv(Rw, mk3(1.f, 0.f, 0.f));
        st3(L.ccdir[idx][0], mulRv(Rw, mk3(l1.x * a.x, l1.y * a.y, l1.z * a.z)));
        st3(L.ccdir[idx][1], mulRv(Rw, mk3(l2.x * a.x, l2.y * a.y, l2.z * a.z)));
    }
    return __popcll(bal);
}

// ------------------------------------------------------------------------------------
// one batch of rows, lane = row: M^-1 J^T by the ABA delta sweeps
// (btMultiBody::calcAccelerationDeltasMultiDof [U]) plus the row's denominator and
// right-hand side.  KIND 0 motor (-> L.Mm, kept for the limit rows), 1 normal,
// 2 friction A, 3 friction B (-> the 64-row staging buffer).
// Staging row layout: [0..21] M^-1 J^T, [22] rhs impulse, [23] denominator, [24] 1/denominator.
// ------------------------------------------------------------------------------------
template <class LT, int KIND>
__device__ void build_batch_v2(LT& L, const DevModel& M, int lane, int nc) {
    constexpr int N = LT::kN;
    const int count = KIND == 0 ? N : nc;
    if (lane < count) {
        const bool motor = KIND == 0;
        int k;
        f3 P = mk3(0, 0, 0), d = mk3(0, 0, 0);
        float* Mrow = motor ? L.Mm[lane] : L.stM[lane];
        if (motor) {
            k = lane + 1;
        } else {
            k = L.ccbody[lane];
            P = ld3(L.ccP[lane]);
            d = KIND == 1 ? mk3(0.f, 0.f, 1.f) : ld3(L.ccdir[lane][KIND == 2 ? 0 : 1]);
        }
        f3 pN = mk3(0, 0, 0), pF = mk3(0, 0, 0);
        for (int b = N; b >= 1; b--) {
            f3 ax = ld3(L.ax[b]);
            if (!motor && b == k) {
                pN = pN - cross(P - ld3(L.o[b]), d);
                pF = pF - d;
            }
            float u = -dot(ax, pN);
            if (motor && b == k) u += 1.0f;
            Mrow[6 + b - 1] = u;
            float t = u * L.Dinv[b];
            f3 paN = pN + ld3(L.Ua[b]) * t, paF = pF + ld3(L.Ub[b]) * t;
            pN = paN + cross(ld3(L.r[b]), paF);
            pF = paF;
        }
        f3 J0 = mk3(0, 0, 0), J1 = mk3(0, 0, 0);
        if (!motor) {
            J0 = cross(P - ld3(L.o[0]), d);
            J1 = d;
            if (k == 0) { pN = pN - J0; pF = pF - d; }
        }
        float p0[6] = {pN.x, pN.y, pN.z, pF.x, pF.y, pF.z}, a0[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) sum -= L.Inv0[6 * i + j] * p0[j];
            a0[i] = sum;
            Mrow[i] = sum;
        }
        f3 al = mk3(a0[0], a0[1], a0[2]), a = mk3(a0[3], a0[4], a0[5]);
        const float* gb = L.base() + 7;
        float den = dot(J0, al) + dot(J1, a);
        float rv = dot(J0, ld3(gb)) + dot(J1, ld3(gb + 3));
        for (int b = 1; b <= N; b++) {
            a = a + cross(al, ld3(L.r[b]));
            float u = Mrow[6 + b - 1];
            float qdd = (u - (dot(ld3(L.Ua[b]), al) + dot(ld3(L.Ub[b]), a))) * L.Dinv[b];
            f3 ax = ld3(L.ax[b]);
            al = al + ax * qdd;
            Mrow[6 + b - 1] = qdd;
            if (!motor) {
                float Jb = (b <= k) ? dot(ax, cross(P - ld3(L.o[b]), d)) : 0.f;
                den += Jb * qdd;
                rv += Jb * L.qd()[b - 1];
            } else if (b == k) {
                den = qdd;
            }
        }
        float dinv = den > 1.1920929e-7f ? 1.0f / den : 0.f;
        float target;
        if (KIND == 0) {
            float cur = L.qd()[lane];
            float want = M.kp * (L.targets[lane] - L.q()[lane]) * M.inv_dt + cur + M.kd * (0.f - cur);
            target = want - cur;
        } else if (KIND == 1) {
            float pen = L.ccdist[lane] + M.slop;
            target = -rv + (pen > 0.f ? -pen * M.inv_dt : -pen * M.contact_erp * M.inv_dt);
        } else {
            target = -rv;
        }
        float* S = motor ? L.MmS[lane] : &L.stM[lane][22];
        S[0] = target * dinv;
        S[1] = den;
        S[2] = dinv;
    }
    lds_sync();
}

// per-lane constants of the (half, d) layout
struct LaneK {
    int h, d;
    bool isdof;
    float m22, m24;  // -1 at d==22 / +1 at d==24, else 0
    int spoff;       // staging column that feeds this lane's special value (22 rhs, 23 den)
    f3 oL, aL;       // for d < 3 and d >= 6:  J[d] = aL . ((P - oL) x dir);  3 <= d < 6: dir[d-3]
    int bL;          // joint index (body) of lane d, 0 for the base components
};

// Fill the halves of one register slot from the rows staged by the last batch.
// KIND 0 motor (row = 2s + h, from L.Mm), 1 normal (contact 2s + h; both halves),
// 2 friction A (contact s -> lower half only), 3 friction B (contact s -> upper half only).
template <class LT, int KIND>
__device__ __forceinline__ void load_slot(LT& L, const LaneK& K, int s, int count, float& RJ, float& RM) {
    const int row = (KIND == 0 || KIND == 1) ? 2 * s + K.h : s;
    const bool valid = row < count;
    const int rs = valid ? row : 0;
    // staged normals: rows 0..31 of the batch sit in staging rows 0..31, rows 32..63 after them
    const float* st = KIND == 0 ? L.Mm[rs] : L.stM[rs];
    const float* sc = KIND == 0 ? L.MmS[rs] : &L.stM[rs][22];
    const int dd = K.isdof ? K.d : 0;
    float mval = st[dd];
    float dinv = sc[2];
    float sp = sc[K.spoff];
    float jd;
    if (KIND == 0) {
        jd = (K.d == 6 + rs) ? 1.0f : 0.0f;
    } else {
        f3 P = ld3(L.ccP[rs]);
        f3 dir = KIND == 1 ? mk3(0.f, 0.f, 1.f) : ld3(L.ccdir[rs][KIND == 2 ? 0 : 1]);
        const int k = L.ccbody[rs];
        float v = dot(K.aL, cross(P - K.oL, dir));
        v = K.d == 3 ? dir.x : (K.d == 4 ? dir.y : (K.d == 5 ? dir.z : v));
        jd = (K.bL <= k) ? v : 0.0f;
    }
    float rj = K.isdof ? jd * dinv : sp * K.m22 + sp * K.m24;
    float rm = K.isdof ? mval : 0.0f;
    if (!valid) { rj = 0.f; rm = 0.f; }
    const bool mine = KIND == 2 ? (K.h == 0) : (KIND == 3 ? (K.h == 1) : true);
    RJ = mine ? rj : RJ;
    RM = mine ? rm : RM;
}

// old with lane l replaced by a wave-uniform value (once per substep: a select is fine)
__device__ __forceinline__ float wrlane(float old, float v_uniform, int l) {
    return ((int)threadIdx.x == l) ? v_uniform : old;
}

// Two consecutive single rows living in the two halves of one slot (hand-written, 25 VALU).
// ORDER 0: lower-half row first, then the upper-half row (coupling scalar in lane 57);
// ORDER 1: upper first, then lower (coupling in lane 25) -- motors are walked backwards on
// even iterations.  For each row:  a' = med3(-dot, LO, HI), dI = a' - a; the second row's dot
// first receives  c * dI_first.  Both contributions are exchanged between the halves, and the
// accumulated impulses (lane 23 / 55) are updated through E2355.
// RES: also accumulate max |dI * den| (lane 24 / 56) into lsq.
#define SNK_DUO_ASM(SEL)                                                                          \
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
__device__ __forceinline__ void duo_step(float& RJ, const float RM, float& dv, float LO, float HI, float E2355,
                                         unsigned long long lowmask, float& lsq) {
    float t, x, dF, dS, c2;
    float s0, s1, s2, s3, s4;
    // F = first row, S = second row.  Dots land in lane 31 (lower row) / 63 (upper row), the
    // accumulated impulses sit in lanes 23 / 55; v_cndmask gives every lane its own half's dI
    // (mask = lower half -> takes the lower row's value).
    if (ORDER == 0) {
        asm volatile(SNK_DUO_ASM("v_cndmask_b32_e64 %[t], %[dS], %[dF], %[lowmask]\n\t")
                     : [t] "=&v"(t), [x] "=&v"(x), [dF] "=&v"(dF), [dS] "=&v"(dS), [c2] "=&v"(c2), [s0] "=&s"(s0),
                       [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [RJ] "+v"(RJ), [dv] "+v"(dv)
                     : [RM] "v"(RM), [LO] "v"(LO), [HI] "v"(HI), [E] "v"(E2355), [lowmask] "s"(lowmask), [AF] "n"(23),
                       [AS] "n"(55), [CL] "n"(57), [DF] "n"(31), [DS] "n"(63));
    } else {
        asm volatile(SNK_DUO_ASM("v_cndmask_b32_e64 %[t], %[dF], %[dS], %[lowmask]\n\t")
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
__device__ __forceinline__ void cone_step(float& RJ, const float RM, const float RJnorm, float& dv, float EPS,
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

// slot map
constexpr int kSlotMotor = 0;     // 8 slots
constexpr int kSlotNormal = 8;    // 32 slots
constexpr int kSlotFric = 40;     // 64 slots
constexpr int kSlots = 104;

// four consecutive duo slots (8 rows): slots BASE..BASE+3, backwards if ORDER == 1
template <int ORDER, bool RES, int BASE>
__device__ __forceinline__ void duos4(float (&RJ)[kSlots], float (&RM)[kSlots], float& dv, float LO, float HI, float E2355,
                                      unsigned long long lowmask, float& lsq) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int s = BASE + (ORDER ? 3 - i : i);
        duo_step<ORDER, RES>(RJ[s], RM[s], dv, LO, HI, E2355, lowmask, lsq);
    }
}
// eight consecutive friction pairs (contacts 8G..8G+7); the normal impulse of contact ci sits
// in slot kSlotNormal + ci/2, lane 23 (even ci) or 55 (odd ci)
template <bool RES, int G>
__device__ __forceinline__ void cones8(float (&RJ)[kSlots], float (&RM)[kSlots], float& dv, float EPS, float E2355,
                                       unsigned long long lowmask, float& lsq) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c0 = 8 * G + 2 * i;
        cone_step<23, RES>(RJ[kSlotFric + c0], RM[kSlotFric + c0], RJ[kSlotNormal + (c0 >> 1)], dv, EPS, E2355, lowmask, lsq);
        cone_step<55, RES>(RJ[kSlotFric + c0 + 1], RM[kSlotFric + c0 + 1], RJ[kSlotNormal + (c0 >> 1)], dv, EPS, E2355, lowmask, lsq);
    }
}

template <class LT>
__device__ __forceinline__ void substep_v2(LT& L, const DevModel& M, int lane, float mu, int& iters, int& ncontacts) {
    constexpr int N = LT::kN;
    constexpr int ND = N + 6;
    static_assert(N == 16, "v2 is laid out for the 16-link chain");
    const float dt = M.dt;
    // (1) contacts of the current pose, (2) bias forces with gravity, joint damping torque
    const int nc = __builtin_amdgcn_readfirstlane(find_contacts_v2(L, M, lane));
    ncontacts = nc;
    if (lane < N) {
        float qd = L.qd()[lane];
        L.qd_old[lane] = qd;
        L.tauj[lane] = -M.joint_damp * qd;   // PyBullet adds URDF joint damping as a torque [U]
    }
    body_bias<LT, true>(L, M, lane);
    lds_sync();
    aba_main<LT, true>(L, M, lane);
    // joint-0 force sensor, first pass [U] (parked in LDS: nothing but rows may live across the solve)
    {
        f3 zb = mulRv(L.R[0], ld3(M.zbase));
        f3 v_old = ld3(L.base() + 10);
        float nv0 = sqrtf(dot(v_old, v_old));
        f3 a1 = ld3(&L.acc0[3]);
        float fz1 = -dot(zb, (a1 - mk3(0.f, 0.f, M.gz)) * M.m_root + v_old * (M.m_root * (M.lin_damp + M.lin_damp * nv0)));
        if (lane == 0) L.fz_park = fz1;
    }
    // (3) v += a dt (clamped)
    if (lane < 6) {
        float x = L.base()[7 + lane] + L.acc0[lane] * dt;
        L.base()[7 + lane] = fminf(fmaxf(x, -M.max_vel), M.max_vel);
    } else if (lane < ND) {
        float x = L.qd()[lane - 6] + L.qdd[lane - 6] * dt;
        L.qd()[lane - 6] = fminf(fmaxf(x, -M.max_vel), M.max_vel);
    }
    lds_sync();

    // (4) rows -> registers.  The lane=row builder needs ~60 registers of its own, so it only
    // runs while at most the 128 friction-row registers are live: motors go to LDS first,
    // then friction A, friction B, normals through the 64-row staging, motors are loaded last.
    float RJ[kSlots], RM[kSlots];
    {
        LaneK K;
        K.h = lane >> 5;
        K.d = lane & 31;
        K.isdof = K.d < ND;
        K.m22 = K.d == 22 ? -1.0f : 0.0f;
        K.m24 = K.d == 24 ? 1.0f : 0.0f;
        K.spoff = K.d == 24 ? 1 : 0;
        K.bL = (K.d >= 6 && K.isdof) ? K.d - 5 : 0;
        K.oL = ld3(L.o[K.bL]);
        K.aL = (K.d >= 6 && K.isdof) ? ld3(L.ax[K.bL])
                                     : mk3(K.d == 0 ? 1.f : 0.f, K.d == 1 ? 1.f : 0.f, K.d == 2 ? 1.f : 0.f);
        build_batch_v2<LT, 0>(L, M, lane, nc);
        build_batch_v2<LT, 2>(L, M, lane, nc);
#pragma unroll
        for (int g = 0; g < 8; g++) {
            if (nc > 8 * g) {
#pragma unroll
                for (int s = 8 * g; s < 8 * g + 8; s++) {
                    RJ[kSlotFric + s] = 0.f; RM[kSlotFric + s] = 0.f;
                    load_slot<LT, 2>(L, K, s, nc, RJ[kSlotFric + s], RM[kSlotFric + s]);
                }
            } else {
#pragma unroll
                for (int s = 8 * g; s < 8 * g + 8; s++) { RJ[kSlotFric + s] = 0.f; RM[kSlotFric + s] = 0.f; }
            }
        }
        lds_sync();
        build_batch_v2<LT, 3>(L, M, lane, nc);
#pragma unroll
        for (int g = 0; g < 8; g++) {
            if (nc > 8 * g) {
#pragma unroll
                for (int s = 8 * g; s < 8 * g + 8; s++) load_slot<LT, 3>(L, K, s, nc, RJ[kSlotFric + s], RM[kSlotFric + s]);
            }
        }
        lds_sync();
        build_batch_v2<LT, 1>(L, M, lane, nc);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (nc > 16 * g) {
#pragma unroll
                for (int s = 8 * g; s < 8 * g + 8; s++) load_slot<LT, 1>(L, K, s, nc, RJ[kSlotNormal + s], RM[kSlotNormal + s]);
            } else {
#pragma unroll
                for (int s = 8 * g; s < 8 * g + 8; s++) { RJ[kSlotNormal + s] = 0.f; RM[kSlotNormal + s] = 0.f; }
            }
        }
#pragma unroll
        for (int s = 0; s < 8; s++) load_slot<LT, 0>(L, K, s, N, RJ[kSlotMotor + s], RM[kSlotMotor + s]);
        lds_sync();
    }

    // coupling scalars of the duos (motors and normals), once per substep:
    //   lower lane 25 <- RJ_lower . RM_upper  (used when the upper row goes first),
    //   upper lane 57 <- RJ_upper . RM_lower  (used when the lower row goes first)
    {
        const bool lo_ = lane < 32;
#pragma unroll
        for (int s = 0; s < kSlotFric; s++) {
            swap2 sw = half_swap(RM[s], RM[s]);             // a = [RM_lower, RM_lower], b = [RM_upper, RM_upper]
            float t = half_reduce(RJ[s] * (lo_ ? sw.b : sw.a));
            float c_lo = rdlane(t, 31), c_up = rdlane(t, 63);
            RJ[s] = wrlane(wrlane(RJ[s], c_lo, 25), c_up, 57);
        }
        if (M.cone == 0) {   // pyramid friction resolves the two directions one after the other
#pragma unroll
            for (int s = kSlotFric; s < kSlots; s++) {
                swap2 sw = half_swap(RM[s], RM[s]);
                float t = half_reduce(RJ[s] * (lo_ ? sw.b : sw.a));
                float c_lo = rdlane(t, 31), c_up = rdlane(t, 63);
                RJ[s] = wrlane(wrlane(RJ[s], c_lo, 25), c_up, 57);
            }
        }
    }

    // violated joint limits (rare): kept as LDS rows, processed by a generic path
    int nlim = 0;
    {
        bool viol = false;
        float sgn = 0.f, pen = 0.f;
        if (lane < N) {
            float qj = L.q()[lane];
            float plo = qj - M.jlo, phi = M.jhi - qj;
            if (plo <= 0.f) { viol = true; sgn = 1.f; pen = plo; }
            else if (phi <= 0.f) { viol = true; sgn = -1.f; pen = phi; }
        }
        unsigned long long bal = __ballot(viol);
        nlim = __builtin_amdgcn_readfirstlane(__popcll(bal));
        if (viol) {
            int idx = __popcll(bal & ((1ull << lane) - 1ull));
            float den = L.Mm[lane][6 + lane];
            float dinv = den > 1.1920929e-7f ? 1.0f / den : 0.f;
            float rel = sgn * L.qd()[lane];
            L.nc_joint[idx] = lane; L.nc_sign[idx] = sgn;
            L.nc_rhs[idx] = (-rel + (-pen) * M.limit_erp * M.inv_dt) * dinv;
            L.nc_dinv[idx] = dinv; L.nc_den[idx] = den;
            L.nc_lo[idx] = 0.f; L.nc_hi[idx] = M.limit_max; L.nc_app[idx] = 0.f;
        }
        if (nlim) lds_sync();
    }

    // (5) projected Gauss-Seidel on the register-resident rows
    float dv;
    {
        const int d = lane & 31;
        dv = d == 22 ? 1.0f : (d == 23 ? -1.0f : 0.0f);
        const float E2355 = d == 23 ? 1.0f : 0.0f;
        const float MU = mu;
        const float ENORM = d == 23 ? 1.0f : (d == 27 ? mu : 0.0f);   // normals also keep mu*lambda_n (lane 27/59)
        const float EPS = 1e-30f;
        const float mi = M.max_motor_imp;
        const float NMI = -mi, PMI = mi, ZERO = 0.f, BIG = 1e10f;
        const unsigned long long LOWMASK = 0x00000000FFFFFFFFull;
        const float thr = sqrtf(M.resid_thr);
        const int n_iter = M.n_iter;
        const bool cone = M.cone != 0;
        int it = 0;
        for (; it < n_iter; it++) {
            // Bullet leaves the sweep when max_rows |dI * den| <= threshold.  `exceeded` becomes 1
            // as soon as a group of rows shows a larger residual; from then on the residual-free
            // variants (RES = false) run.
            int exceeded = 0;
            int ncl = nc;
            asm volatile("" : "+s"(ncl));   // keeps the group-active compares from being hoisted and spilled
            auto check = [&](float lsq) {
                float m = fmaxf(rdlane(lsq, 24), rdlane(lsq, 56));
                exceeded = __builtin_amdgcn_readfirstlane(m > thr ? 1 : 0);
            };
            auto limit_rows = [&](bool fwd) {
                for (int jj = 0; jj < nlim; jj++) {
                    const int idx = fwd ? jj : nlim - 1 - jj;
                    const int j = __builtin_amdgcn_readfirstlane(L.nc_joint[idx]);
                    const float sg = L.nc_sign[idx];
                    float un = sg * rdlane(dv, 6 + j);
                    float a0 = L.nc_app[idx];
                    float dI = L.nc_rhs[idx] - un * L.nc_dinv[idx];
                    float sum = fminf(fmaxf(a0 + dI, L.nc_lo[idx]), L.nc_hi[idx]);
                    dI = sum - a0;
                    L.nc_app[idx] = sum;
                    float mv = (d < ND) ? L.Mm[j][d] : 0.f;
                    dv += sg * mv * dI;
                    if (__builtin_amdgcn_readfirstlane(fabsf(dI * L.nc_den[idx]) > thr ? 1 : 0)) exceeded = 1;
                }
            };
#define SNK_DUOS4(ORDER, BASE, LO, HI)                                                              \
    if (exceeded) { float l_ = 0.f; duos4<ORDER, false, BASE>(RJ, RM, dv, LO, HI, E2355, LOWMASK, l_); } \
    else { float l_ = 0.f; duos4<ORDER, true, BASE>(RJ, RM, dv, LO, HI, E2355, LOWMASK, l_); check(l_); }
#define SNK_DUOS4N(BASE)                                                                            \
    if (exceeded) { float l_ = 0.f; duos4<0, false, BASE>(RJ, RM, dv, ZERO, BIG, ENORM, LOWMASK, l_); }  \
    else { float l_ = 0.f; duos4<0, true, BASE>(RJ, RM, dv, ZERO, BIG, ENORM, LOWMASK, l_); check(l_); }
#define SNK_CONES8(G)                                                                              \
    if (exceeded) { float l_ = 0.f; cones8<false, G>(RJ, RM, dv, EPS, E2355, LOWMASK, l_); }    \
    else { float l_ = 0.f; cones8<true, G>(RJ, RM, dv, EPS, E2355, LOWMASK, l_); check(l_); }
            // non-contact rows: list = [limits..., motors 0..15], walked forwards on odd
            // iterations and backwards on even ones
            if (it & 1) {
                limit_rows(true);
                SNK_DUOS4(0, kSlotMotor, NMI, PMI)
                SNK_DUOS4(0, kSlotMotor + 4, NMI, PMI)
            } else {
                SNK_DUOS4(1, kSlotMotor + 4, NMI, PMI)
                SNK_DUOS4(1, kSlotMotor, NMI, PMI)
                limit_rows(false);
            }
            // contact normals in contact order, one scalar branch per 8 contacts; rows past the
            // active count are inert zeros
            if (ncl > 0) { SNK_DUOS4N(kSlotNormal + 0) }
            if (ncl > 8) { SNK_DUOS4N(kSlotNormal + 4) }
            if (ncl > 16) { SNK_DUOS4N(kSlotNormal + 8) }
            if (ncl > 24) { SNK_DUOS4N(kSlotNormal + 12) }
            if (ncl > 32) { SNK_DUOS4N(kSlotNormal + 16) }
            if (ncl > 40) { SNK_DUOS4N(kSlotNormal + 20) }
            if (ncl > 48) { SNK_DUOS4N(kSlotNormal + 24) }
            if (ncl > 56) { SNK_DUOS4N(kSlotNormal + 28) }
            // friction pairs in contact order
            if (cone) {
                if (ncl > 0) { SNK_CONES8(0) }
                if (ncl > 8) { SNK_CONES8(1) }
                if (ncl > 16) { SNK_CONES8(2) }
                if (ncl > 24) { SNK_CONES8(3) }
                if (ncl > 32) { SNK_CONES8(4) }
                if (ncl > 40) { SNK_CONES8(5) }
                if (ncl > 48) { SNK_CONES8(6) }
                if (ncl > 56) { SNK_CONES8(7) }
            } else {
                // pyramid friction (not Bullet's default here): the two directions are resolved one
                // after the other (a duo with its coupling scalar), bounds +-mu*lambda_n, and -- as
                // in Bullet -- skipped altogether while the normal impulse is zero
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    if (ncl > 8 * g) {
#pragma unroll
                        for (int ci = 8 * g; ci < 8 * g + 8; ci++) {
                            float lim = rdlane(RJ[kSlotNormal + (ci >> 1)], (ci & 1) ? 55 : 23) * MU;
                            if (__builtin_amdgcn_readfirstlane(lim > 0.f ? 1 : 0)) {
                                float nlim_ = -lim, lsq = 0.f;
                                duo_step<0, true>(RJ[kSlotFric + ci], RM[kSlotFric + ci], dv, nlim_, lim, E2355, LOWMASK, lsq);
                                if (!exceeded) check(lsq);
                            }
                        }
                    }
                }
            }
#undef SNK_DUOS4
#undef SNK_DUOS4N
#undef SNK_CONES8
            if (!exceeded || it >= n_iter - 1) { it++; break; }
        }
        iters = it;
        // accumulated impulses -> LDS, by (slot, half)
        if (d == 23) {
#pragma unroll
            for (int s = 0; s < kSlots; s++) L.app[2 * s + (lane >> 5)] = RJ[s];
        }
    }
    lds_sync();

    // (6) constraint pass for the joint-0 sensor [U]
    if (lane <= N) {
        const int b = lane;
        f3 eN = mk3(0, 0, 0), eF = mk3(0, 0, 0);
        for (int ci = 0; ci < nc; ci++) {
            if (L.ccbody[ci] == b) {
                f3 F = (mk3(0.f, 0.f, 1.f) * L.app[2 * kSlotNormal + ci] +
                        ld3(L.ccdir[ci][0]) * L.app[2 * (kSlotFric + ci)] +
                        ld3(L.ccdir[ci][1]) * L.app[2 * (kSlotFric + ci) + 1]) * M.inv_dt;
                eF = eF + F;
                eN = eN + cross(ld3(L.ccP[ci]) - ld3(L.o[b]), F);
            }
        }
        st3(&L.ext[b][0], eN);
        st3(&L.ext[b][3], eF);
    }
    if (lane < N) L.tauj[lane] = -M.joint_damp * L.qd_old[lane] + L.app[lane] * M.inv_dt;
    lds_sync();
    if (lane == 0) {
        for (int i = 0; i < nlim; i++) L.tauj[L.nc_joint[i]] += L.nc_sign[i] * L.nc_app[i] * M.inv_dt;
    }
    lds_sync();
    {
        f3 wp = ld3(L.base() + 7), vp = ld3(L.base() + 10);
        if (lane == 0) { st3(L.w[0], wp); st3(L.v[0], vp); }
        for (int b = 1; b <= N; b++) {
            f3 ax = ld3(L.ax[b]), rb = ld3(L.r[b]);
            float qdb = L.qd()[b - 1];
            f3 w = wp + ax * qdb, v = vp + cross(wp, rb);
            f3 za = cross(wp, ax) * qdb, zl = cross(wp, cross(wp, rb));
            if (lane == 0) { st3(L.w[b], w); st3(L.v[b], v); st3(&L.zeta[b][0], za); st3(&L.zeta[b][3], zl); }
            wp = w; vp = v;
        }
    }
    lds_sync();
    body_bias<LT, false>(L, M, lane);
    lds_sync();
    aba_main<LT, false>(L, M, lane);
    float fz;
    {
        f3 zb = mulRv(L.R[0], ld3(M.zbase));
        f3 v1 = ld3(L.base() + 10);
        float nv1 = sqrtf(dot(v1, v1));
        f3 a2 = ld3(&L.acc0[3]);
        fz = L.fz_park - dot(zb, a2 * M.m_root + v1 * (M.m_root * (M.lin_damp + M.lin_damp * nv1)));
    }
    // (7) apply the solver's delta-v (lower half's copy), motor torques, integrate positions
    if (lane < 6) {
        float x = L.base()[7 + lane] + dv;
        L.base()[7 + lane] = fminf(fmaxf(x, -M.max_vel), M.max_vel);
    } else if (lane < ND) {
        float x = L.qd()[lane - 6] + dv;
        x = fminf(fmaxf(x, -M.max_vel), M.max_vel);
        L.qd()[lane - 6] = x;
        L.q()[lane - 6] += dt * x;
    }
    if (lane < N) L.taum()[lane] = L.app[lane] * M.inv_dt;
    lds_sync();
    {
        float* bs = L.base();
        f3 om = ld3(bs + 7), vl = ld3(bs + 10);
        float fA = sqrtf(dot(om, om));
        const float kThr = 0.78539816339744831f;   // [U] ANGULAR_MOTION_THRESHOLD
        if (fA * dt > kThr) fA = kThr / dt;
        float sc;
        if (fA < 0.001f) sc = 0.5f * dt - (dt * dt * dt) * 0.020833333333f * fA * fA;
        else sc = sinf(0.5f * fA * dt) / fA;
        float dx = om.x * sc, dy = om.y * sc, dz = om.z * sc, dw = cosf(fA * dt * 0.5f);
        float qx = bs[3], qy = bs[4], qz = bs[5], qw = bs[6];
        float nw = dw * qw - dx * qx - dy * qy - dz * qz;
        float nx = dw * qx + dx * qw + dy * qz - dz * qy;
        float ny = dw * qy - dx * qz + dy * qw + dz * qx;
        float nz = dw * qz + dx * qy - dy * qx + dz * qw;
        float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
        lds_sync();
        if (lane == 0) {
            bs[0] += dt * vl.x; bs[1] += dt * vl.y; bs[2] += dt * vl.z;
            bs[3] = nx * inv; bs[4] = ny * inv; bs[5] = nz * inv; bs[6] = nw * inv;
            L.fz() = fz;
        }
    }
    lds_sync();
    fk_vel(L, M, lane);
}

}  // namespace snk
