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
//   * slots: 0..7 motors (motor s in the lower half, motor 8+s in the upper half),
//     8..39 contact normals (contact s lower, contact 32+s upper), 40..103 friction pairs
//     (direction A in the lower half, B in the upper half, so the two dots of Bullet's
//     cone-friction pair come out of one DPP reduction)
//   * a half's dot: 4 DPP row steps + row_bcast:15 -> lane 31 / 63 -> v_readlane
//   * single rows are processed in PHASES of one half with EXEC masked to that half (the
//     other half's copy of delta-v is refreshed once per phase by v_permlane32_swap);
//     friction pairs use both halves and exchange their contributions with the same swap
//   * the early-exit residual (max over rows of |dI * den|) is only evaluated until the
//     first row exceeds the threshold in an iteration: afterwards a scalar branch skips it
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
// KIND 0 motor (row = s + 8h, from L.Mm), 1 normal (contact s + 32h; both halves),
// 2 friction A (contact s -> lower half only), 3 friction B (contact s -> upper half only).
template <class LT, int KIND>
__device__ __forceinline__ void load_slot(LT& L, const LaneK& K, int s, int count, float& RJ, float& RM) {
    const int row = KIND == 0 ? s + 8 * K.h : (KIND == 1 ? s + 32 * K.h : s);
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

// Single-row update of the row living in half H of its slot.  HM is 1 in that half's lanes
// and 0 in the other's, so only that half's copy of delta-v changes (the other copy is
// refreshed by a half swap before rows of the other half run); E is 1 at the half's lane 23.
// lsq accumulates max |dI * den| in lane 24 (H=0) / 56 (H=1) for Bullet's early-exit test.
template <int H>
__device__ __forceinline__ void row_step(float& RJ, const float RM, float& dv, float lo, float hi, float E, float HM,
                                         float& lsq, float INF) {
    float s_dot = half_dot1<H>(RJ * dv);
    float s_a = rdlane(RJ, H ? 55 : 23);
    float nw = __builtin_amdgcn_fmed3f(-s_dot, lo, hi);
    float dI = nw - s_a;
    dv = fmaf(RM, dI * HM, dv);
    RJ = fmaf(E, dI, RJ);
    lsq = __builtin_amdgcn_fmed3f(lsq, fabsf(RJ * dI), INF);
    __builtin_amdgcn_sched_barrier(0);   // one row at a time: overlapping rows only adds register pressure
}

// Bullet's cone-friction pair: both dots from one reduction, radial projection onto the disc
// of radius mu * lambda_n, contributions exchanged between the halves.
__device__ __forceinline__ void cone_step(float& RJ, const float RM, float& dv, float s_an, float MU, float EPS,
                                          float E2355, bool lower, float& lsq, float INF) {
    float s_dA, s_dB;
    half_dot2(RJ * dv, s_dA, s_dB);
    float r2 = fmaf(s_dA, s_dA, EPS);
    r2 = fmaf(s_dB, s_dB, r2);
    float lim = s_an * MU;
    float sc = __builtin_amdgcn_fmed3f(lim * __builtin_amdgcn_rsqf(r2), 0.0f, 1.0f);
    float nA = -s_dA * sc, nB = -s_dB * sc;
    float s_aA = rdlane(RJ, 23), s_aB = rdlane(RJ, 55);
    float dIA = nA - s_aA, dIB = nB - s_aB;
    float dIh = lower ? dIA : dIB;
    float c = RM * dIh;
    swap2 sw = half_swap(c, c);
    dv += sw.a;
    dv += sw.b;
    RJ = fmaf(E2355, dIh, RJ);
    lsq = __builtin_amdgcn_fmed3f(lsq, fabsf(RJ * dIh), INF);
    __builtin_amdgcn_sched_barrier(0);
}

// slot map
constexpr int kSlotMotor = 0;     // 8 slots
constexpr int kSlotNormal = 8;    // 32 slots
constexpr int kSlotFric = 40;     // 64 slots
constexpr int kSlots = 104;

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
            if (nc > 8 * g) {
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
        const float E23 = lane == 23 ? 1.0f : 0.0f;
        const float E55 = lane == 55 ? 1.0f : 0.0f;
        const bool lower = lane < 32;
        const float HM0 = lower ? 1.0f : 0.0f;
        const float HM1 = lower ? 0.0f : 1.0f;
        const float MU = mu;
        const float EPS = 1e-30f;
        const float mi = M.max_motor_imp;
        const float thr = sqrtf(M.resid_thr);
        const int n_iter = M.n_iter;
        const bool cone = M.cone != 0;
        int it = 0;
        const float INF = __builtin_inff();
        for (; it < n_iter; it++) {
            float lsq0 = 0.f, lsq1 = 0.f, lsqP = 0.f;
            int exceeded = 0;
            int ncl = nc;
            asm volatile("" : "+s"(ncl));   // keeps the group-active compares from being hoisted and spilled
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
            // give the other half the fresh copy of delta-v
            auto sync_from_lower = [&]() { swap2 sw = half_swap(dv, dv); dv = sw.a; };
            auto sync_from_upper = [&]() { swap2 sw = half_swap(dv, dv); dv = sw.b; };
            // non-contact rows: list = [limits..., motors 0..15], walked forwards on odd
            // iterations and backwards on even ones; motors 0..7 live in the lower halves
            if (it & 1) {
                limit_rows(true);
#pragma unroll
                for (int j = 0; j < 8; j++) row_step<0>(RJ[kSlotMotor + j], RM[kSlotMotor + j], dv, -mi, mi, E23, HM0, lsq0, INF);
                sync_from_lower();
#pragma unroll
                for (int j = 0; j < 8; j++) row_step<1>(RJ[kSlotMotor + j], RM[kSlotMotor + j], dv, -mi, mi, E55, HM1, lsq1, INF);
                sync_from_upper();
            } else {
#pragma unroll
                for (int j = 7; j >= 0; j--) row_step<1>(RJ[kSlotMotor + j], RM[kSlotMotor + j], dv, -mi, mi, E55, HM1, lsq1, INF);
                sync_from_upper();
#pragma unroll
                for (int j = 7; j >= 0; j--) row_step<0>(RJ[kSlotMotor + j], RM[kSlotMotor + j], dv, -mi, mi, E23, HM0, lsq0, INF);
                sync_from_lower();
                limit_rows(false);
            }
            // contact normals in contact order: 0..31 live in the lower halves, 32..63 in the
            // upper; one scalar branch per group of 8, rows past the active count are inert
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (ncl > 8 * g) {
#pragma unroll
                    for (int ci = 8 * g; ci < 8 * g + 8; ci++)
                        row_step<0>(RJ[kSlotNormal + ci], RM[kSlotNormal + ci], dv, 0.f, 1e10f, E23, HM0, lsq0, INF);
                }
            }
            sync_from_lower();
            if (ncl > 32) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    if (ncl > 32 + 8 * g) {
#pragma unroll
                        for (int ci = 8 * g; ci < 8 * g + 8; ci++)
                            row_step<1>(RJ[kSlotNormal + ci], RM[kSlotNormal + ci], dv, 0.f, 1e10f, E55, HM1, lsq1, INF);
                    }
                }
                sync_from_upper();
            }
            // friction pairs in contact order
            if (cone) {
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    if (ncl > 8 * g) {
#pragma unroll
                        for (int ci = 8 * g; ci < 8 * g + 8; ci++) {
                            float s_an = rdlane(RJ[kSlotNormal + (ci & 31)], (ci & 32) ? 55 : 23);
                            cone_step(RJ[kSlotFric + ci], RM[kSlotFric + ci], dv, s_an, MU, EPS, E2355, lower, lsqP, INF);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    if (ncl > 8 * g) {
#pragma unroll
                        for (int ci = 8 * g; ci < 8 * g + 8; ci++) {
                            float lim = rdlane(RJ[kSlotNormal + (ci & 31)], (ci & 32) ? 55 : 23) * MU;
                            // lim == 0 clamps both impulses to 0 (Bullet skips the rows: same result)
                            row_step<0>(RJ[kSlotFric + ci], RM[kSlotFric + ci], dv, -lim, lim, E23, HM0, lsq0, INF);
                            sync_from_lower();
                            row_step<1>(RJ[kSlotFric + ci], RM[kSlotFric + ci], dv, -lim, lim, E55, HM1, lsq1, INF);
                            sync_from_upper();
                        }
                    }
                }
            }
            float lsq = fmaxf(fmaxf(rdlane(lsq0, 24), rdlane(lsq1, 56)), fmaxf(rdlane(lsqP, 24), rdlane(lsqP, 56)));
            if (__builtin_amdgcn_readfirstlane(lsq > thr ? 1 : 0)) exceeded = 1;
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
                f3 F = (mk3(0.f, 0.f, 1.f) * L.app[2 * (kSlotNormal + (ci & 31)) + (ci >> 5)] +
                        ld3(L.ccdir[ci][0]) * L.app[2 * (kSlotFric + ci)] +
                        ld3(L.ccdir[ci][1]) * L.app[2 * (kSlotFric + ci) + 1]) * M.inv_dt;
                eF = eF + F;
                eN = eN + cross(ld3(L.ccP[ci]) - ld3(L.o[b]), F);
            }
        }
        st3(&L.ext[b][0], eN);
        st3(&L.ext[b][3], eF);
    }
    if (lane < N) L.tauj[lane] = -M.joint_damp * L.qd_old[lane] + L.app[2 * (lane & 7) + (lane >> 3)] * M.inv_dt;
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
    if (lane < N) L.taum()[lane] = L.app[2 * (lane & 7) + (lane >> 3)] * M.inv_dt;
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
