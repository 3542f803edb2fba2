// snk_device.hpp -- gfx950 device code of the batched snake stepper.
//
// ONE WAVEFRONT (64 lanes) PER ENVIRONMENT, one wave per workgroup.  The env's state
// record is loaded coalesced from HBM into LDS once per env-step and written back once;
// every physics substep (0..41 per env-step, snake.py:284-304) runs on chip.
//
// Formulation (differs on purpose from oracle/): the chain of n+1 composite bodies is
// described in WORLD-ALIGNED axes with each body's quantities referenced to its own joint
// origin o_b, and classical accelerations [alpha_b ; a(o_b)].  Transfers between
// neighbouring bodies are then pure translations by r_b = o_b - o_{b-1}; the joint
// subspace is S_b = [axis_b ; 0].  Articulated-body algorithm = Featherstone's three
// sweeps in those coordinates.
//
// What replaces what (reference call sites, /root/reference):
//   env_step_kernel      SnakeGymEnv.step (SnakeGymEnv.py:33-50) + Snake.step servo loop
//                        (snake.py:274-306) + worker auto-reset (multiprocessing_env.py:13-15)
//   substep()            pybullet.stepSimulation (snake.py:286) after
//                        setJointMotorControlArray(POSITION_CONTROL) (snake.py:221)
//   write_obs()          Snake.getObservation (snake.py:209-217)
//   mean_height()        Snake.checkSnakeHeight (snake.py:237-245)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "snk_model.hpp"

// -DSNK_PROFILE (bullet-envs_amd/build.py --profile): s_memtime stamps between the phases of a substep; the phase
// durations (ticks) replace the motor torques of the record (tools/profile_phases.py)
#ifdef SNK_PROFILE
#define SNK_STAMP(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); prof_t[i] = t_; }
#else
#define SNK_STAMP(i)
#endif

namespace snk {

// ----------------------------------------------------------------------------------
// small vector helpers
// ----------------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, f3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// y = R v (R row-major 3x3)
__device__ __forceinline__ f3 mulRv(const float* R, f3 v) {
    return mk3(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z,
               R[6] * v.x + R[7] * v.y + R[8] * v.z);
}
__device__ __forceinline__ f3 mulRtv(const float* R, f3 v) {
    return mk3(R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z,
               R[2] * v.x + R[5] * v.y + R[8] * v.z);
}
// symmetric 3x3 stored xx xy xz yy yz zz
__device__ __forceinline__ f3 mulSv(const float* S, f3 v) {
    return mk3(S[0] * v.x + S[1] * v.y + S[2] * v.z, S[1] * v.x + S[3] * v.y + S[4] * v.z,
               S[2] * v.x + S[4] * v.y + S[5] * v.z);
}
// W = R S R^T for symmetric S (body -> world), result symmetric
__device__ __forceinline__ void rotSym(const float* R, const float* S, float* W) {
    float T[9];   // T = R S
#pragma unroll
    for (int i = 0; i < 3; i++) {
        T[3 * i + 0] = R[3 * i] * S[0] + R[3 * i + 1] * S[1] + R[3 * i + 2] * S[2];
        T[3 * i + 1] = R[3 * i] * S[1] + R[3 * i + 1] * S[3] + R[3 * i + 2] * S[4];
        T[3 * i + 2] = R[3 * i] * S[2] + R[3 * i + 1] * S[4] + R[3 * i + 2] * S[5];
    }
    W[0] = T[0] * R[0] + T[1] * R[1] + T[2] * R[2];
    W[1] = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];
    W[2] = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
    W[3] = T[3] * R[3] + T[4] * R[4] + T[5] * R[5];
    W[4] = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
    W[5] = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
}

// ----------------------------------------------------------------------------------
// wave primitives (wave64, DPP; gfx9 row_shr / row_bcast forms)
// ----------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {
    // x + (x moved by the DPP pattern; lanes with no source or masked rows add 0)
    int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, true);
    return x + __int_as_float(y);
}
// Sum over lanes 0..W-1 (W = 32 or 64), returned wave-uniform.
template <int W>
__device__ __forceinline__ float wave_sum(float x) {
    x = dpp_add<0xB1, 0xf>(x);    // quad_perm [1,0,3,2]
    x = dpp_add<0x4E, 0xf>(x);    // quad_perm [2,3,0,1]
    x = dpp_add<0x114, 0xf>(x);   // row_shr:4
    x = dpp_add<0x118, 0xf>(x);   // row_shr:8   -> lane 15 of each row = row total
    x = dpp_add<0x142, 0xa>(x);   // row_bcast:15 into rows 1,3 -> lane 31 = sum 0..31
    if (W == 64) {
        x = dpp_add<0x143, 0xc>(x);   // row_bcast:31 into rows 2,3 -> lane 63 = sum 0..63
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    }
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31));
}
// Sum over the active lanes 0 .. LAST (32 <= LAST < 48) of a wave running with exactly those lanes enabled, returned
// wave-uniform: the same DPP steps; lane LAST collects its own row's partial sum and the total of lanes 0 .. 31.
template <int LAST>
__device__ __forceinline__ float cols_sum(float x) {
    static_assert(LAST >= 32 && LAST < 48, "lane LAST must sit in row 2");
    x = dpp_add<0xB1, 0xf>(x);
    x = dpp_add<0x4E, 0xf>(x);
    x = dpp_add<0x114, 0xf>(x);
    x = dpp_add<0x118, 0xf>(x);
    x = dpp_add<0x142, 0xa>(x);
    x = dpp_add<0x143, 0xc>(x);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), LAST));
}
// ... and the maximum of non-negative values, same lanes
template <int LAST>
__device__ __forceinline__ float cols_max(float x) {
    static_assert(LAST >= 32 && LAST < 48, "lane LAST must sit in row 2");
    auto step = [](float v, auto ctrl, auto rows) {
        return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, decltype(rows)::value, 0xf, true)));
    };
    x = step(x, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
    x = step(x, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), LAST));
}
// the lane's index within its wave (= threadIdx.x of the one-wave workgroups here), recomputed instead of kept
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// The lane index as a value the compiler cannot see through: per-lane addresses built from it are computed where they
// are used.  (Built from the kernel's own threadIdx.x they are loop-invariant, get hoisted in front of the servo loop and
// stay live across it -- twenty VGPRs in round 3's first build, which the solve's row registers then paid for with
// reloads from scratch memory inside the Gauss-Seidel loop.)
__device__ __forceinline__ int launder_lane(int lane) {
    asm volatile("" : "+v"(lane));
    return lane;
}
__device__ __forceinline__ float lane_bcast(float x, int src_lane_uniform) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src_lane_uniform));
}

// one DPP step of six independent reductions (operands a .. f)
#define SNK_RED64x6_STEP(MODE)                              \
    "v_add_f32_dpp %[a], %[a], %[a] " MODE "\n\t"            \
    "v_add_f32_dpp %[b], %[b], %[b] " MODE "\n\t"            \
    "v_add_f32_dpp %[c], %[c], %[c] " MODE "\n\t"            \
    "v_add_f32_dpp %[d], %[d], %[d] " MODE "\n\t"            \
    "v_add_f32_dpp %[e], %[e], %[e] " MODE "\n\t"            \
    "v_add_f32_dpp %[f], %[f], %[f] " MODE "\n\t"
// (seven: operand g as well)
#define SNK_RED64x7_STEP(MODE) SNK_RED64x6_STEP(MODE) "v_add_f32_dpp %[g], %[g], %[g] " MODE "\n\t"

// ----------------------------------------------------------------------------------
// LDS image of one environment
// ----------------------------------------------------------------------------------
template <int N, int NL>
struct LdsCommon {
    static constexpr int kN = N;
    static constexpr int NB = N + 1;     // composite bodies
    static constexpr int ND = N + 6;     // generalized velocity [omega_w, v_w, qd]
    static constexpr int NC = 4 * N;     // contact slots: 2n cylinders x 2 end caps
    static constexpr int NR = 3 * NC;    // contact rows: normal + 2 friction
    static constexpr int REC = (N <= 16) ? 64 : 128;
    // HBM record, same order: base[13] q[N] qd[N] taum[N] fz prev_x
    float rec[REC];
    // per body, world axes
    float R[NB][9], o[NB][3], r[NB][3], ax[NB][3], cw[NB][3];
    float w[NB][3], v[NB][3], zeta[NB][6], p[NB][6];
    float IA[NB][21];            // articulated inertia: A(6 sym) B(9) C(6 sym)
    float Ua[NB][3], Ub[NB][3], Dinv[NB], u[NB];
    float Inv0[36];
    float qd_old[N], tauj[N], qdd[N], targets[N];
    float acc0[6];
    // non-contact rows kept in LDS (NL of them): in v1 limits + motors (2 N), in v2 only the (rare) limit rows (N)
    int nc_joint[NL];
    float nc_sign[NL], nc_rhs[NL], nc_dinv[NL], nc_den[NL], nc_lo[NL], nc_hi[NL], nc_app[NL];

    // SNK_POISON=1 (snk_create; tests): every float of the image becomes a NaN before an environment is loaded, so that a
    // read of something this substep did not write shows up in the outputs instead of depending on what the previous
    // environment -- or another kernel -- left behind.  (The integer tables are left alone: a NaN's bits as an index
    // would turn such a bug into a wild address.)
    __device__ __forceinline__ void poison_common(int lane) {
        float* a = rec;
        const int nf = (int)(reinterpret_cast<float*>(nc_joint) - a);
        for (int i = lane; i < nf; i += 64) a[i] = __int_as_float(0x7fc00000);
        float* b = nc_sign;
        for (int i = lane; i < 7 * NL; i += 64) b[i] = __int_as_float(0x7fc00000);
    }
    __device__ __forceinline__ float* base() { return rec; }
    __device__ __forceinline__ float* q() { return rec + 13; }
    __device__ __forceinline__ float* qd() { return rec + 13 + N; }
    __device__ __forceinline__ float* taum() { return rec + 13 + 2 * N; }
    __device__ __forceinline__ float& fz() { return rec[13 + 3 * N]; }
    __device__ __forceinline__ float& prev_x() { return rec[14 + 3 * N]; }
    __device__ __forceinline__ float& fz3() { return rec[15 + 3 * N]; }   // reaction Fz of the first motor joint (streamed-row solve)
};

// Register budget of the streamed-row solve (experiments: -DSNK_V1_RESN=.. etc. through build.py's `defines`; the defaults
// are what ships).  Round 4's sweep on configs[3] (profiles/r04_c32_ring_sweep.txt): look-ahead beyond 16 normals / 16
// friction pairs buys nothing, every resident normal saves its 320 bytes per iteration -- 40 / 16 / 16 runs at 91.7 k
// env-steps/s against 89.9 k for round 2-3's 32 / 32 / 16; 48 resident or 24 pairs in flight spill into the loop.
#ifndef SNK_V1_RESN
#define SNK_V1_RESN 40
#endif
#ifndef SNK_V1_RINGN
#define SNK_V1_RINGN 16
#endif
#ifndef SNK_V1_RINGF
#define SNK_V1_RINGF 16
#endif
#ifndef SNK_LB
#define SNK_LB 2
#endif
// (the same three numbers for the copy that runs inside the register-resident kernels, pgs_v1<LT, INPLACE = true>)
#ifndef SNK_IP_RESN
#define SNK_IP_RESN 32
#endif
#ifndef SNK_IP_RINGN
#define SNK_IP_RINGN 32
#endif
#ifndef SNK_IP_RINGF
#define SNK_IP_RINGF 16
#endif
#ifndef SNK_V1_LDAUX
#define SNK_V1_LDAUX 0      // cache policy bits of the streamed rows' buffer loads (experiments: 1 sc0, 2 nt, 16 sc1)
#endif

template <int N, bool V2>
struct Lds;

// v1: every constraint row staged in LDS (any chain length; used for the 32-link config)
template <int N>
struct Lds<N, false> : LdsCommon<N, 2 * N> {
    static constexpr bool kV2 = false;
    // ground-contact slots: 128 for BOTH chain lengths.  The 32-link chain's two end-cap points per cylinder; for the
    // 16-link chain every point its 32 cylinders' manifolds can hold (4 each) -- this solve is where an environment
    // goes whose contacts do not fit the register-resident solve's 64 slots (snk_api.hip: overflow list), so that no
    // 16-link contact is ever left without rows
    // contact slots: every point the 2N cylinders' manifolds can hold (4 each) -- Bullet has no limit, and neither has
    // this solve for the ground contacts
    static constexpr int NC = 8 * N, NR = 3 * NC, ND = N + 6;
    float ext_[LdsCommon<N, 2 * N>::NB][6];      // link forces of the constraint pass
    __device__ __forceinline__ float* ext(int b) { return ext_[b]; }
    __device__ __forceinline__ void poison(int lane) {
        this->poison_common(lane);
        poison_own(lane);
    }
    // the members behind the common part only: what a register-resident kernel's rare streamed-row substep finds there is
    // that kernel's own leftovers (finite numbers: the poison of the environment's load is long overwritten)
    __device__ __forceinline__ void poison_own(int lane) {
        for (int i = lane; i < LdsCommon<N, 2 * N>::NB * 6; i += 64) (&ext_[0][0])[i] = __int_as_float(0x7fc00000);
        for (int i = lane; i < (NC + kRing + 3) * 4; i += 64) (&acc[0][0])[i] = __int_as_float(0x7fc00000);
    }
    int clist[NC];               // compact contact index -> slot
    int cidx[NC];                // slot -> compact contact index (-1: not in contact)
    // The contact rows themselves (J and M^-1 J^T, 2 x 384 x 38 floats = 117 KB) do not fit LDS next
    // to anything else; they live in a per-resident-wave block of global memory that the solve
    // streams once per iteration (see pgs_v1), one 320-byte record [J | M^-1 J^T] per row.
    static constexpr int kRing = 32;                       // padding entries behind the ground contacts' impulses (the link-link contacts' live there)
    static constexpr int kResN = SNK_V1_RESN;              // contacts whose normal rows stay in registers over the solve
    static constexpr int kRingN = SNK_V1_RINGN;            // normal rows in flight behind them
    static constexpr int kRingF = SNK_V1_RINGF;            // friction pairs in flight
    // link-link (self-collision) contacts follow the ground contacts in the compact list: at most kMaxSelf of them,
    // geometry slots NC .. NC + kMaxSelf - 1
    static constexpr int kMaxSelf = kRing;
    static constexpr int NCT = NC + kMaxSelf;              // contact slots in all
    // record order: the NCT normal rows, then the NCT friction pairs (A, B) -- each phase of the solve streams its own
    // rows back to back, every fetched cache line used whole (interleaved by contact, a phase used 320 of every 960
    // bytes and paid for the neighbours' half lines), then 3 rows that stay zero.  Inside a record the vectors are
    // interleaved by column -- a friction pair's 640 bytes are [JA0 JB0 MA0 MB0 JA1 ...], and the normals of contacts 2p,
    // 2p + 1, which the solve resolves in one step, share 640 bytes [J(2p)0 M(2p)0 J(2p+1)0 M(2p+1)0 J(2p)1 ...] (since the end of
    // round 4; 320 bytes per contact before: +1.5 %) -- so that lane d fetches everything it needs for a step with ONE
    // 16-byte load
    static constexpr int kRows = 3 * NCT + 3;
    static constexpr int kFric = NCT;                      // first friction record
    // a row of the block: [J (ND floats), pad, M^-1 J^T (ND floats), pad], 320 B = five aligned 64-B
    // sectors for 304 useful bytes (separate, unaligned 152-B rows fetched 1.4x their size)
    static constexpr int kRS = 80;                         // floats per row of the block
    static constexpr int kMO = 40;                         // float offset of the M^-1 J^T half
    // The two pad columns of each half carry the row's scalars, so that the solve needs nothing but the accumulated
    // impulses in LDS (round 2: 7.7 KB of per-contact scalars {rhs, den, 1/den, a} shrank to 2.5 KB of a's):
    //   J half:        columns < ND  J / den;   column kSpec  -rhs = -target / den;   column kSpec + 1  0
    //   M^-1 J^T half: columns < ND  M^-1 J^T;  column kSpec  0;                      column kSpec + 1  den
    // With delta-v's lane kSpec held at 1 the row's dot IS (J.dv)/den - rhs, and lane kSpec + 1 of the step's
    // M^-1 J^T dI is dI * den, the row's residual (the same layout trick as the register-resident solve's d = 22 / 24).
    static constexpr int kSpec = kMO - 2;
    static_assert(ND <= kSpec, "row layout");
    // behind the rows: the contact geometry of the NCT slots, 20 floats each: P[3] (point on body kA), distance,
    // friction direction A[3], B[3], normal[3], PB[3] (point on body kB), kA, kB (-1: the ground), friction scale,
    // pad (written lane = slot by find_contacts_v1 / find_self_contacts_v1, read by the row builder and the
    // sensor pass)
    static constexpr int kGeo = 20;
    static constexpr size_t kGeoOff = (size_t)kRows * kRS;
    // behind the geometry: M^-1 e_j of the n motor / limit rows, kMO floats each (columns >= ND zero).  The solve keeps
    // them in registers; they left LDS (4.9 KB for 32 links) so that eight waves fit a CU
    static constexpr size_t kMmOff = kGeoOff + (size_t)NCT * kGeo;     // M^-1: ND rows (6 base, then the joints) of kMO floats
    // (+ 6 rows and one more Y for the free box of obstacle 2, a second "tree" with six velocity components of its own
    //  behind the snake's: lanes ND .. ND + 5 of the solve, body index N + 1 in the contact records)
    static constexpr int kBoxBody = N + 1;
    static constexpr size_t kYOff = kMmOff + (size_t)(N + 6 + 6) * kMO;     // Y_k of every body (build_rows_v1), 6 x kMO floats each
    static constexpr size_t kRowFloats = kYOff + (size_t)(N + 2) * 6 * kMO;
    static_assert(N + 6 + 6 <= kSpec || N > 16, "the free box's lanes must fit in front of the scalar columns (16 links)");
    // accumulated impulses of contact ci: {normal, friction A, friction B, -}
    alignas(16) float acc[NC + kRing + 3][4];        // (+ the entries the solve reads ahead of the last pair)
    static_assert(NCT <= NC + kRing, "the impulses of the link-link contacts live in the ring's padding entries");
    int nplane;                  // ground contacts of this substep (the link-link contacts follow them)
    // obstacle 2: the free box while this wave holds the environment -- state [pos3, quat4, omega3, vel3], its world
    // rotation and world inverse inertia (sym6) for this substep, its manifold with the plane (4 x (a3, b.x, b.y,
    // lambda)) and the point count; travels with the state record (d_box)
    float box[13], bR[9], bIw[6], bman[24];
    int bmn;
};

// v2: rows live in VGPRs during the solve; LDS only stages one 64-row batch while they are built
template <int N>
struct Lds<N, true> : LdsCommon<N, N> {
    static constexpr bool kV2 = true;
    static constexpr int NC = 4 * N, ND = N + 6;
    static_assert(N + 6 + 3 <= 32, "v2 packs two rows per 64-lane register");
    float Mm[N][ND];             // M^-1 e_j for the motor / limit rows
    static constexpr int kObs = 8;                        // room for contacts with the obstacle box (behind the ground's)
    float ccP[NC][3], ccdist[NC];                         // indexed by COMPACT contact index
    unsigned char ccbody[NC], ccds[NC];                   // ... the contact's body; its entry of cdir
    // friction directions A, B: one entry per CYLINDER (all ground contacts of a cylinder share them), then one per
    // obstacle contact, whose normals are obn (a ground contact's is +z)
    float cdir[2 * N + kObs][2][3], obn[kObs][3];
    float stM[64][25];           // staging of one 64-row batch: M^-1 J^T [22], rhs, den, 1/den
    // link forces of the constraint pass: columns 8..13 of the staging rows, which that pass uses in columns 0..5 only
    __device__ __forceinline__ float* ext(int b) { return &stM[b][8]; }
    float MmS[N][4];             // the motors' rhs, den, 1/den, target velocity change (their M^-1 rows are Mm)
    float fz_park, fz3_park;     // first-pass parts of the joint-0 force and of the first motor joint's reaction, parked across the solve
    int nplane;                  // ground contacts of this substep (the obstacle's follow them in the compact list)
    // contacts of cylinder c: compact indices [cylbase[c], + cyln[c]); cylkeep[c]: which of its cached manifold points
    // they are (bit j = point j has rows; contact_model 1)
    unsigned char cylbase[2 * N], cyln[2 * N], cylkeep[2 * N];
    float app[2 * (N / 2 + NC / 2 + NC)];   // accumulated impulses by (register slot, half)
    // contact_model 1: the environment's persistent contact manifolds stay HERE while a wave holds the environment
    // (read and updated every substep, lane = cylinder); they travel to and from global memory with the state record
    // only -- at the start and the end of an env-step and at a hand-off between waves.  Component-major, so that lane
    // = cylinder strides by one word: per cached point j the floats [6 j .. 6 j + 5] = point on the link in link
    // coordinates (3), point on the ground x, y (its z is the plane's: 0), the normal impulse of the last substep
    float mfl[24][2 * N];
    unsigned char mfn[2 * N];    // cached points of cylinder c
    __device__ __forceinline__ void poison(int lane) {
        this->poison_common(lane);
        auto fill = [&](float* a, int n) { for (int i = lane; i < n; i += 64) a[i] = __int_as_float(0x7fc00000); };
        fill(&Mm[0][0], N * ND); fill(&ccP[0][0], NC * 3); fill(ccdist, NC); fill(&cdir[0][0][0], (2 * N + kObs) * 6);
        fill(&obn[0][0], kObs * 3); fill(&stM[0][0], 64 * 25); fill(&MmS[0][0], N * 4);
        fill(app, 2 * (N / 2 + NC / 2 + NC)); fill(&mfl[0][0], 24 * 2 * N);
    }
};

__device__ __forceinline__ void lds_sync() { __syncthreads(); }

// What the caller knows about a substep's place in the servo loop (snake.py:283-304).  The joint-0
// force sensor (obs[55]) is only observable after the LAST substep of an env-step, so the
// register-resident substep runs its second ABA pass only when this substep can be the last one.
struct SensorHint {
    bool always;        // single-substep API: every substep is observable
    int counter_next;   // value of `counter` after this substep
    float h_prev;       // checkSnakeHeight's mean height of the pose the substep starts from
};

// True when the substep whose solve just produced `dv` (this lane's component of the velocity
// change) can be the LAST of its env-step, i.e. when obs[55] (the joint-0 force sensor, the second
// ABA pass) can be observed: the servo error after it is within the tolerance, or the counter
// reaches its cap, or the mean height can cross its threshold.  The first two are evaluated
// exactly as the loop does (with a 1e-3 safety factor on the tolerance); for the third, no
// sampled point can move further in one substep than dt * (|v| + L_chain * (|omega| + sum |qd|)):
// rigid rotations about the base and the joints.
template <class LT>
__device__ __forceinline__ bool sensor_pass_needed(LT& L, const DevModel& M, int lane, float dv, const SensorHint& hint) {
    constexpr int N = LT::kN;
    constexpr int ND = N + 6;
    if (hint.always) return true;
    const float dt = M.dt;
    float e = 0.f, wgt = 0.f;
    if (lane < ND) {
        const float vold = lane < 6 ? L.base()[7 + lane] : L.qd()[lane - 6];
        const float x = fminf(fmaxf(vold + dv, -M.max_vel), M.max_vel);
        if (lane >= 6) e = L.targets[lane - 6] - (L.q()[lane - 6] + dt * x);
        wgt = fabsf(x) * ((lane >= 3 && lane < 6) ? 1.0f : 0.0639f * (N + 2));
    }
    const float se = wave_sum<64>(e * e);
    const float reach = dt * wave_sum<64>(wgt);
    const float tol = M.servo_tol * 1.001f;
    const bool sensor = !(se > tol * tol) || hint.counter_next > M.max_counter || !(hint.h_prev + reach < M.height_thr);
    return __builtin_amdgcn_readfirstlane(sensor ? 1 : 0) != 0;
}

// What this wave stored to its own block of global memory (constraint rows, contact geometry: written lane = row, read
// lane = column) becomes visible to its own later loads: the stores have left the wave (vmcnt) and this CU's vector L1
// holds no line from before them (buffer_inv sc1).  The XCD's L2 is the point of coherence for writer and reader alike --
// the same wave -- so nothing has to be written back: __threadfence() here (rounds 1-4) also ran buffer_wbl2, a
// write-back of every dirty line of the XCD's L2, two to three times per streamed-row substep.
//
// THE INVARIANT THIS RELIES ON (load-bearing since round 4; DESIGN.md 4 has the full producer -> consumer table): nothing
// a wave stores with PLAIN stores is ever read by ANOTHER wave inside the same launch.  Every byte that crosses waves
// in a launch -- the state record, the contact cache block, the free box's record, the substep counter at a hand-off;
// the queue entry, tickets, counters -- is stored write-through (sc1 / dwordx4 sc1) or by an agent-scope atomic, and
// ordered by s_waitcnt vmcnt(0) in front of the queue entry (sched_push).  The L2 write-back that __threadfence() did
// here as a side effect is therefore not needed by any reader; a new cross-wave datum must come with its own
// write-through stores, not lean on this function.
// gfx942 / gfx950 ISA only: `vmcnt` counts stores there, and `buffer_inv sc1` is this family's L1 invalidate.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "own_stores_visible(): written for gfx942 / gfx950 (vmcnt covers stores, buffer_inv sc1); other targets need __threadfence()"
#endif
__device__ __forceinline__ void own_stores_visible() {
    asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
}

// ----------------------------------------------------------------------------------
// S1: forward kinematics + link velocities of the chain (serial recurrence, evaluated
// uniformly by the wave; lane 0 stores)
// ----------------------------------------------------------------------------------
template <class LT>
__device__ void fk_vel(LT& L, const DevModel& M, int lane) {
    constexpr int N = LT::kN;
    const float* bs = L.base();
    float qx = bs[3], qy = bs[4], qz = bs[5], qw = bs[6];
    float dd = qx * qx + qy * qy + qz * qz + qw * qw;
    float s2 = 2.0f / dd;
    float xs = qx * s2, ys = qy * s2, zs = qz * s2;
    float wx = qw * xs, wy = qw * ys, wz = qw * zs;
    float xx = qx * xs, xy = qx * ys, xz = qx * zs, yy = qy * ys, yz = qy * zs, zz = qz * zs;
    float Rp[9] = {1 - (yy + zz), xy - wz, xz + wy, xy + wz, 1 - (xx + zz), yz - wx, xz - wy, yz + wx, 1 - (xx + yy)};
    f3 op = ld3(bs), wp = ld3(bs + 7), vp = ld3(bs + 10);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) L.R[0][i] = Rp[i];
        st3(L.o[0], op); st3(L.w[0], wp); st3(L.v[0], vp);
        st3(L.r[0], mk3(0, 0, 0)); st3(L.ax[0], mk3(0, 0, 0));
#pragma unroll
        for (int i = 0; i < 6; i++) L.zeta[0][i] = 0.f;
    }
    // sin/cos of all joint angles at once (lane = joint); the serial chain below picks them up
    // with v_readlane instead of evaluating sincosf sixteen times one after the other
    float snv = 0.f, csv = 1.f;
    if (lane < N) sincosf(L.q()[lane], &snv, &csv);
#pragma unroll 4
    for (int b = 1; b <= N; b++) {
        const float* Rf = M.Rfix[b];
        float T[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                T[3 * i + j] = Rp[3 * i] * Rf[j] + Rp[3 * i + 1] * Rf[3 + j] + Rp[3 * i + 2] * Rf[6 + j];
        float qdb = L.qd()[b - 1];
        const float sn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(snv), b - 1));
        const float cs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(csv), b - 1));
        float Rn[9];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            Rn[3 * i + 0] = cs * T[3 * i] - sn * T[3 * i + 2];
            Rn[3 * i + 1] = T[3 * i + 1];
            Rn[3 * i + 2] = sn * T[3 * i] + cs * T[3 * i + 2];
        }
        f3 rb = mulRv(Rp, ld3(M.pfix[b]));
        f3 o = op + rb;
        f3 ax = mk3(T[1], T[4], T[7]);
        f3 w = wp + ax * qdb;
        f3 v = vp + cross(wp, rb);
        f3 za = cross(wp, ax) * qdb;
        f3 zl = cross(wp, cross(wp, rb));
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; i++) L.R[b][i] = Rn[i];
            st3(L.o[b], o); st3(L.r[b], rb); st3(L.ax[b], ax); st3(L.w[b], w); st3(L.v[b], v);
            st3(&L.zeta[b][0], za); st3(&L.zeta[b][3], zl);
        }
#pragma unroll
        for (int i = 0; i < 9; i++) Rp[i] = Rn[i];
        op = o; wp = w; vp = v;
    }
    lds_sync();
}

// checkSnakeHeight's mean z over {`base` link COM, OUTPUT_BODY origins} (snake.py:237-245)
template <class LT>
__device__ float mean_height(LT& L, const DevModel& M, int lane) {
    constexpr int N = LT::kN;
    float z = 0.f;
    if (lane == 0) z = L.o[0][2] + L.R[0][6] * M.hbase[0] + L.R[0][7] * M.hbase[1] + L.R[0][8] * M.hbase[2];
    else if (lane <= N) z = L.o[lane][2];
    return wave_sum<64>(z) * (1.0f / (N + 1));
}

// ----------------------------------------------------------------------------------
// S2: per-body bias forces (lane = body): p_b = [w x I w ; m w x (w x c)] - external
// ----------------------------------------------------------------------------------
template <class LT, bool FIRST>
__device__ void body_bias(LT& L, const DevModel& M, int lane) {
    constexpr int N = LT::kN;
    if (lane <= N) {
        const int b = lane;
        const float* R = L.R[b];
        f3 w = ld3(L.w[b]), v = ld3(L.v[b]);
        float m = M.mass[b];
        f3 cw = mulRv(R, ld3(M.com[b]));
        float Ibar[6];
        rotSym(R, M.Ib[b], Ibar);
        f3 pN = cross(w, mulSv(Ibar, w));
        f3 pF = cross(w, cross(w, cw)) * m;
        // [U] btMultiBody link damping, per original URDF link of the composite
        float Irw[6];
        rotSym(R, M.Irot[b], Irw);
        float nw = sqrtf(dot(w, w));
        pN = pN + mulSv(Irw, w) * (M.ang_damp + M.ang_damp * nw);
        const int ns = M.nsub[b];
        for (int s = 0; s < ns; s++) {
            f3 cs = mulRv(R, ld3(M.sub_c[b][s]));
            f3 vs = v + cross(w, cs);
            float nv = sqrtf(dot(vs, vs));
            f3 F = vs * (M.sub_m[b][s] * (M.lin_damp + M.lin_damp * nv));   // opposes motion
            pF = pF + F;
            pN = pN + cross(cs, F);
        }
        if (FIRST) {
            f3 G = mk3(0.f, 0.f, m * M.gz);
            pF = pF - G;
            pN = pN - cross(cw, G);
            st3(L.cw[b], cw);
            // articulated inertia initial value  [[Ibar, m[c]x], [-m[c]x, m 1]]
            float* IA = L.IA[b];
#pragma unroll
            for (int i = 0; i < 6; i++) IA[i] = Ibar[i];
            float hx = m * cw.x, hy = m * cw.y, hz = m * cw.z;
            IA[6] = 0.f; IA[7] = -hz; IA[8] = hy;
            IA[9] = hz;  IA[10] = 0.f; IA[11] = -hx;
            IA[12] = -hy; IA[13] = hx; IA[14] = 0.f;
            IA[15] = m; IA[16] = 0.f; IA[17] = 0.f; IA[18] = m; IA[19] = 0.f; IA[20] = m;
        } else {
            pN = pN - ld3(L.ext(b));
            pF = pF - ld3(L.ext(b) + 3);
        }
        st3(&L.p[b][0], pN);
        st3(&L.p[b][3], pF);
    }
}

// ----------------------------------------------------------------------------------
// ground contacts (lane = slot): cylinder c = slot/2 on body (c+1)/2, end cap = slot&1.
// Implicit cylinder + margin against the plane z = 0; kept when closer than the
// breaking threshold [U].  Friction directions (0,-1,0),(1,0,0) scaled anisotropically in
// the cylinder link's axes: d' = Rc diag(aniso) Rc^T d  (snake.py:104-106).
// ----------------------------------------------------------------------------------
// defined in snk_pgs_v2.hpp (shared by both solves)
__device__ __forceinline__ void rim_point(const DevModel& M, f3 dl, float& lx, float& ly);
constexpr int kMfFloats = 28;      // per cylinder: [count, 3 pad, 4 x (a3, b.x, b.y, lambda)]
__device__ __forceinline__ int lane_prefix3(int cnt, int lane, int& total);
__device__ __forceinline__ void friction_dirs(const DevModel& M, const float* Rw, f3& dA, f3& dB);
__device__ __forceinline__ void cyl_world_rot(const float* Rb, const float* Rc, float* Rw);
template <class LT>
__device__ int find_contacts_manifold_v1(LT& L, const DevModel& M, int lane, float* __restrict__ rows, float* __restrict__ mf,
                                         unsigned long long* __restrict__ ovf);
// obstacle 2, the free box (snk_freebox.hpp)
template <class LT>
__device__ __forceinline__ void box_frame_v1(LT& L, const DevModel& M, int lane);
template <class LT>
__device__ __forceinline__ int find_box_ground_v1(LT& L, const DevModel& M, int lane, float mu_ground, int first,
                                                  float* __restrict__ rows, unsigned long long* __restrict__ ovf);

template <class LT>
__device__ int find_contacts_v1(LT& L, const DevModel& M, int lane, float* __restrict__ rows, float* __restrict__ mf,
                                unsigned long long* __restrict__ ovf) {
    constexpr int N = LT::kN;
    if (M.contact_model == 1) return find_contacts_manifold_v1(L, M, lane, rows, mf, ovf);
    int total = 0;
    for (int base = 0; base < 4 * N; base += 64) {
        const int slot = base + lane;
        bool active = false;
        if (slot < 4 * N) {
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
            float lx, ly;
            rim_point(M, dl, lx, ly);
            float lz = (slot & 1) ? M.cyl_hl : -M.cyl_hl;
            f3 loc = mk3(lx + M.margin * dl.x, ly + M.margin * dl.y, lz + M.margin * dl.z);
            f3 P = ld3(L.o[b]) + mulRv(Rb, ld3(M.cyl_c[c])) + mulRv(Rw, loc);
            float dist = P.z;
            active = dist < M.break_thr;
            float* geo = rows + LT::kGeoOff + (size_t)slot * LT::kGeo;
            st3(geo, P);
            geo[3] = dist;
            f3 a = mk3(M.aniso[0], M.aniso[1], M.aniso[2]);
            f3 l1 = mulRtv(Rw, mk3(0.f, -1.f, 0.f));
            f3 l2 = mulRtv(Rw, mk3(1.f, 0.f, 0.f));
            st3(geo + 4, mulRv(Rw, mk3(l1.x * a.x, l1.y * a.y, l1.z * a.z)));
            st3(geo + 7, mulRv(Rw, mk3(l2.x * a.x, l2.y * a.y, l2.z * a.z)) * M.fricB);
            st3(geo + 10, mk3(0.f, 0.f, 1.f));
            st3(geo + 13, mk3(0.f, 0.f, 0.f));
            geo[16] = (float)b; geo[17] = -1.0f; geo[18] = 1.0f; geo[19] = 0.f;
        }
        unsigned long long bal = __ballot(active);
        if (slot < 4 * N) L.cidx[slot] = -1;
        if (active) {
            int idx = total + __popcll(bal & ((1ull << lane) - 1ull));
            L.clist[idx] = slot;
            L.cidx[slot] = idx;
        }
        total += __popcll(bal);
    }
    return total;
}

}  // namespace snk
#include "snk_selfcol.hpp"
namespace snk {

// ----------------------------------------------------------------------------------
// S3: ABA sweeps, evaluated uniformly by the wave (serial recurrence over the chain).
// FACTOR: also builds the articulated inertias IA, U = IA S, D = S^T U and the base inverse.
// ----------------------------------------------------------------------------------
template <class LT, bool FACTOR>
__device__ void aba_main(LT& L, const DevModel& M, int lane) {
    constexpr int N = LT::kN;
    float cA[6], cB[9], cC[6];   // child contribution to the parent's articulated inertia
#pragma unroll
    for (int i = 0; i < 6; i++) { cA[i] = 0.f; cC[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 9; i++) cB[i] = 0.f;
    f3 cN = mk3(0, 0, 0), cF = mk3(0, 0, 0);
    for (int b = N; b >= 1; b--) {
        float A[6], B[9], C[6];
        float* IA = L.IA[b];
#pragma unroll
        for (int i = 0; i < 6; i++) { A[i] = IA[i]; C[i] = IA[15 + i]; }
#pragma unroll
        for (int i = 0; i < 9; i++) B[i] = IA[6 + i];
        if (FACTOR) {
#pragma unroll
            for (int i = 0; i < 6; i++) { A[i] += cA[i]; C[i] += cC[i]; }
#pragma unroll
            for (int i = 0; i < 9; i++) B[i] += cB[i];
        }
        f3 pN = ld3(&L.p[b][0]) + cN, pF = ld3(&L.p[b][3]) + cF;
        f3 ax = ld3(L.ax[b]);
        f3 Ua, Ub;
        float Dinv;
        if (FACTOR) {
            Ua = mulSv(A, ax);
            Ub = mk3(B[0] * ax.x + B[3] * ax.y + B[6] * ax.z, B[1] * ax.x + B[4] * ax.y + B[7] * ax.z,
                     B[2] * ax.x + B[5] * ax.y + B[8] * ax.z);
            Dinv = 1.0f / dot(ax, Ua);
        } else {
            Ua = ld3(L.Ua[b]); Ub = ld3(L.Ub[b]); Dinv = L.Dinv[b];
        }
        f3 za = ld3(&L.zeta[b][0]), zl = ld3(&L.zeta[b][3]);
        float u = L.tauj[b - 1] - dot(ax, pN);
        // IA zeta
        f3 tN = mulSv(A, za) + mk3(B[0] * zl.x + B[1] * zl.y + B[2] * zl.z, B[3] * zl.x + B[4] * zl.y + B[5] * zl.z,
                                   B[6] * zl.x + B[7] * zl.y + B[8] * zl.z);
        f3 tF = mk3(B[0] * za.x + B[3] * za.y + B[6] * za.z, B[1] * za.x + B[4] * za.y + B[7] * za.z,
                    B[2] * za.x + B[5] * za.y + B[8] * za.z) + mulSv(C, zl);
        float uz = dot(Ua, za) + dot(Ub, zl);
        float s = (u - uz) * Dinv;
        f3 paN = pN + tN + Ua * s, paF = pF + tF + Ub * s;
        f3 r = ld3(L.r[b]);
        cN = paN + cross(r, paF);
        cF = paF;
        if (lane == 0) {
            L.u[b] = u;
            if (FACTOR) {
#pragma unroll
                for (int i = 0; i < 6; i++) { IA[i] = A[i]; IA[15 + i] = C[i]; }
#pragma unroll
                for (int i = 0; i < 9; i++) IA[6 + i] = B[i];
                st3(L.Ua[b], Ua); st3(L.Ub[b], Ub); L.Dinv[b] = Dinv;
            }
        }
        if (FACTOR) {
            // Ia = IA - U U^T / D
            float ua[3] = {Ua.x, Ua.y, Ua.z}, ub[3] = {Ub.x, Ub.y, Ub.z};
            float Ap[6], Bp[9], Cp[6];
            Ap[0] = A[0] - ua[0] * ua[0] * Dinv; Ap[1] = A[1] - ua[0] * ua[1] * Dinv; Ap[2] = A[2] - ua[0] * ua[2] * Dinv;
            Ap[3] = A[3] - ua[1] * ua[1] * Dinv; Ap[4] = A[4] - ua[1] * ua[2] * Dinv; Ap[5] = A[5] - ua[2] * ua[2] * Dinv;
            Cp[0] = C[0] - ub[0] * ub[0] * Dinv; Cp[1] = C[1] - ub[0] * ub[1] * Dinv; Cp[2] = C[2] - ub[0] * ub[2] * Dinv;
            Cp[3] = C[3] - ub[1] * ub[1] * Dinv; Cp[4] = C[4] - ub[1] * ub[2] * Dinv; Cp[5] = C[5] - ub[2] * ub[2] * Dinv;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) Bp[3 * i + j] = B[3 * i + j] - ua[i] * ub[j] * Dinv;
            // shift to the parent's origin: X = [[1,0],[-rx,1]];  IA_parent += X^T Ia X
            //   Bn = Bp + rx Cp ;  An = Ap - Bp rx + rx Bn^T ;  Cn = Cp
            f3 c0 = cross(r, mk3(Cp[0], Cp[1], Cp[2]));   // rx * column j of Cp (symmetric)
            f3 c1 = cross(r, mk3(Cp[1], Cp[3], Cp[4]));
            f3 c2 = cross(r, mk3(Cp[2], Cp[4], Cp[5]));
            float Bn[9] = {Bp[0] + c0.x, Bp[1] + c1.x, Bp[2] + c2.x, Bp[3] + c0.y, Bp[4] + c1.y, Bp[5] + c2.y,
                           Bp[6] + c0.z, Bp[7] + c1.z, Bp[8] + c2.z};
            // (-Bp rx) row i = r x row_i(Bp);  (rx Bn^T) column j = r x row_j(Bn)
            f3 e0 = cross(r, mk3(Bp[0], Bp[1], Bp[2])), e1 = cross(r, mk3(Bp[3], Bp[4], Bp[5])),
               e2 = cross(r, mk3(Bp[6], Bp[7], Bp[8]));
            f3 g0 = cross(r, mk3(Bn[0], Bn[1], Bn[2])), g1 = cross(r, mk3(Bn[3], Bn[4], Bn[5])),
               g2 = cross(r, mk3(Bn[6], Bn[7], Bn[8]));
            cA[0] = Ap[0] + e0.x + g0.x;
            cA[1] = Ap[1] + e0.y + g1.x;
            cA[2] = Ap[2] + e0.z + g2.x;
            cA[3] = Ap[3] + e1.y + g1.y;
            cA[4] = Ap[4] + e1.z + g2.y;
            cA[5] = Ap[5] + e2.z + g2.z;
#pragma unroll
            for (int i = 0; i < 9; i++) cB[i] = Bn[i];
#pragma unroll
            for (int i = 0; i < 6; i++) cC[i] = Cp[i];
        }
    }
    // base: [alpha0; a0] = -IA0^-1 p0
    f3 pN = ld3(&L.p[0][0]) + cN, pF = ld3(&L.p[0][3]) + cF;
    float p0[6] = {pN.x, pN.y, pN.z, pF.x, pF.y, pF.z};
    if (FACTOR) {
        float* IA = L.IA[0];
        float A[6], B[9], C[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { A[i] = IA[i] + cA[i]; C[i] = IA[15 + i] + cC[i]; }
#pragma unroll
        for (int i = 0; i < 9; i++) B[i] = IA[6 + i] + cB[i];
        float G[6][6];
        G[0][0] = A[0]; G[0][1] = A[1]; G[0][2] = A[2]; G[1][1] = A[3]; G[1][2] = A[4]; G[2][2] = A[5];
        G[1][0] = A[1]; G[2][0] = A[2]; G[2][1] = A[4];
        G[3][3] = C[0]; G[3][4] = C[1]; G[3][5] = C[2]; G[4][4] = C[3]; G[4][5] = C[4]; G[5][5] = C[5];
        G[4][3] = C[1]; G[5][3] = C[2]; G[5][4] = C[4];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) { G[i][3 + j] = B[3 * i + j]; G[3 + j][i] = B[3 * i + j]; }
        // Gauss-Jordan inverse of the SPD 6x6 (no pivoting)
        float V[6][6];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++) V[i][j] = (i == j) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            float piv = 1.0f / G[k][k];
#pragma unroll
            for (int j = 0; j < 6; j++) { G[k][j] *= piv; V[k][j] *= piv; }
#pragma unroll
            for (int i = 0; i < 6; i++) {
                if (i != k) {
                    float f = G[i][k];
#pragma unroll
                    for (int j = 0; j < 6; j++) { G[i][j] -= f * G[k][j]; V[i][j] -= f * V[k][j]; }
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) L.Inv0[6 * i + j] = V[i][j];
        }
        float a0[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) s -= V[i][j] * p0[j];
            a0[i] = s;
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) L.acc0[i] = a0[i];
        }
    } else {
        float a0[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) s -= L.Inv0[6 * i + j] * p0[j];
            a0[i] = s;
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) L.acc0[i] = a0[i];
        }
    }
    lds_sync();
    if (!FACTOR) return;   // the sensor pass only needs the base acceleration (acc0)
    // forward sweep: joint accelerations
    f3 al = ld3(&L.acc0[0]), a = ld3(&L.acc0[3]);
    for (int b = 1; b <= N; b++) {
        f3 r = ld3(L.r[b]);
        f3 ap = a + cross(al, r) + ld3(&L.zeta[b][3]);
        f3 alp = al + ld3(&L.zeta[b][0]);
        float qdd = (L.u[b] - (dot(ld3(L.Ua[b]), alp) + dot(ld3(L.Ub[b]), ap))) * L.Dinv[b];
        al = alp + ld3(L.ax[b]) * qdd;
        a = ap;
        if (lane == 0) L.qdd[b - 1] = qdd;
    }
    lds_sync();
}

// ----------------------------------------------------------------------------------
// S5: constraint rows of the streamed-row solve, built lane = velocity component (round 1 built them lane = row:
// one pair of ABA delta sweeps per row, see the history of this file).
// A contact row's M^-1 J^T is linear in the 6-dimensional wrench its unit impulse puts on its body:
//     M^-1 J^T = Y_k (tau, f),   tau = (P - o_k) x dir,  f = dir,   Y_k = M^-1 Jbody_k^T   (38 x 6),
// and the Y_k follow from the columns of M^-1 by a recursion down the chain (a wrench on body k about o_k is the
// wrench (tau + r_k x f, f) on body k-1 about o_{k-1} plus the torque ax_k . tau on joint k):
//     Y_0 = M^-1[:, 0..5],   Y_k(tau, f) = Y_{k-1}(tau + r_k x f, f) + M^-1[:, 6+k-1] (ax_k . tau).
// So instead of one pair of ABA delta sweeps PER ROW (416 rows = 7 trips of 64 lanes through two serial 32-body
// recurrences, 19 % of a substep) there is ONE trip of 38 sweeps -- the columns of M^-1, which the motor rows need
// anyway -- nine FMAs per lane and body for the recursion, and per row: six FMAs for M^-1 J^T, J from the lane's own
// joint axis, two wave reductions (denominator, relative velocity), and ONE coalesced store of the finished record.
// (btMultiBodyConstraintSolver::setupMultiBodyContactConstraint / btMultiBody::calcAccelerationDeltasMultiDof [U]: the
//  rows are the same linear map of the same unit impulses; only the order of the floating-point sums differs.)
// ----------------------------------------------------------------------------------
template <class LT>
__device__ void build_rows_v1(LT& L, const DevModel& M, int lane, int nc, int& n_noncontact, float* __restrict__ rows) {
    constexpr int N = LT::kN;
    constexpr int ND = N + 6;
    constexpr int kMO = LT::kMO;
    float* const Mmx = rows + LT::kMmOff;           // M^-1, ND rows of kMO floats (columns >= ND stay zero)
    float mden = 0.f;                               // lane 6+j: M^-1[6+j][6+j], motor j's denominator
    // ---- (a) the columns of M^-1: lane = velocity component d, unit generalized force on it -- the ABA delta sweeps of
    // btMultiBody::calcAccelerationDeltasMultiDof [U].  Backward sweep here; the forward sweep runs body by body inside
    // the contact loop below (fwd), because the spatial acceleration (al, a) it gives body b IS what the contacts of
    // body b need:  Y_b[d] = response of component d to a unit wrench on body b = (M^-1 symmetric) response of body b's
    // twist to a unit force on component d.  Until the end of round 3 Y_b was rebuilt from the TRANSPOSED entries -- the
    // wrench moved rigidly to the base plus a torque on every joint up to b, the responses of all of those summed --
    // which is the same number with ~30 x the round-off (the terms are large and cancel; 32 links: one-substep velocity
    // errors p90 5.9e-2 against the float32 oracle's 2.0e-3, tools/acc_distribution.py 1024 32).
    const bool dofl = lane < ND;
    float* const Mrow = Mmx + (size_t)(dofl ? lane : 0) * kMO;
    const int kj = lane - 5;                        // the joint's body (lanes >= 6)
    f3 al = mk3(0, 0, 0), a = mk3(0, 0, 0);         // this lane's sweep: spatial acceleration of the body reached so far
    if (dofl) {
        const bool isbase = lane < 6;
        f3 pN = mk3(0, 0, 0), pF = mk3(0, 0, 0);
#pragma unroll 4
        for (int b = N; b >= 1; b--) {
            f3 ax = ld3(L.ax[b]);
            float u = -dot(ax, pN);
            if (!isbase && b == kj) u += 1.0f;
            Mrow[6 + b - 1] = u;                    // (parked in the row the forward sweep overwrites with qdd_b)
            float t = u * L.Dinv[b];
            f3 paN = pN + ld3(L.Ua[b]) * t, paF = pF + ld3(L.Ub[b]) * t;
            pN = paN + cross(ld3(L.r[b]), paF);
            pF = paF;
        }
        float p0[6] = {pN.x, pN.y, pN.z, pF.x, pF.y, pF.z}, a0[6];
#pragma unroll
        for (int i = 0; i < 6; i++) if (lane == i) p0[i] = -1.0f;      // unit force on the base: bias -e_i
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) sacc -= L.Inv0[6 * i + j] * p0[j];
            a0[i] = sacc;
            Mrow[i] = sacc;
        }
        al = mk3(a0[0], a0[1], a0[2]);
        a = mk3(a0[3], a0[4], a0[5]);
    }
    // the free box (obstacle 2): a second multibody, block-diagonal in M^-1 -- rows ND .. ND + 5: the world inverse
    // inertia for its angular components, 1 / m for its linear ones
    const bool fbox = N <= 16 && M.obstacle == 2;         // (built for the 16-link chain only: snk_create)
    const bool boxlane = fbox && lane >= ND && lane < ND + 6;
    if (boxlane) {
        const int i = lane - ND;
        float* Mrow = Mmx + (size_t)lane * kMO;
        const float* W = L.bIw;
        const f3 r0 = i == 0 ? mk3(W[0], W[1], W[2]) : (i == 1 ? mk3(W[1], W[3], W[4]) : mk3(W[2], W[4], W[5]));
        Mrow[ND + 0] = i < 3 ? r0.x : 0.f; Mrow[ND + 1] = i < 3 ? r0.y : 0.f; Mrow[ND + 2] = i < 3 ? r0.z : 0.f;
        Mrow[ND + 3] = i == 3 ? M.obs_minv : 0.f; Mrow[ND + 4] = i == 4 ? M.obs_minv : 0.f; Mrow[ND + 5] = i == 5 ? M.obs_minv : 0.f;
    }
    // ---- (b) lane = velocity component d: what J[d] is made of, and the current velocity
    const int d = lane;
    const int jb = d >= 6 ? d - 5 : 0;              // the body this component's joint belongs to (0: the base)
    f3 Aj = mk3(0, 0, 0), Oj = mk3(0, 0, 0), Bj = mk3(0, 0, 0);
    float vd = 0.f;
    if (d < 3) { Aj = mk3(d == 0 ? 1.f : 0.f, d == 1 ? 1.f : 0.f, d == 2 ? 1.f : 0.f); Oj = ld3(L.o[0]); vd = L.base()[7 + d]; }
    else if (d < 6) { Bj = mk3(d == 3 ? 1.f : 0.f, d == 4 ? 1.f : 0.f, d == 5 ? 1.f : 0.f); vd = L.base()[7 + d]; }
    else if (d < ND) { Aj = ld3(L.ax[jb]); Oj = ld3(L.o[jb]); vd = L.qd()[d - 6]; }
    else if (boxlane) {
        const int i = d - ND;
        if (i < 3) { Aj = mk3(i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f); Oj = ld3(L.box); }
        else Bj = mk3(i == 3 ? 1.f : 0.f, i == 4 ? 1.f : 0.f, i == 5 ? 1.f : 0.f);
        vd = L.box[7 + i];
    }
    // does this lane's component move a point of body k?  (the snake's: the joints up to k; the box's six: the box)
    auto moves = [&](int k) { return boxlane ? k == LT::kBoxBody : (k <= N && jb <= k); };
    const bool colv = d < kMO;                      // lanes that own a column of the records
    // Y of the body the sweep has reached: (Yt, Yf) = (al, a); zero on the lanes without a velocity component
    f3 Yt = al, Yf = a;                             // Y_0
    // link-link / obstacle contacts: any body, so every Y_k is kept.  And so it is when the ground contacts do not come in
    // the order of their bodies (snk_params::contact_order): the forward sweep below cannot follow them then, they take their
    // Y from the block like the two-body contacts (a switch of the error bar, not the default: the extra 32 KB of stores and
    // loads per substep are its price)
    const bool sorted = M.contact_order == 0;
    const bool two_body = nc > L.nplane || !sorted;
    float* const Yb = rows + LT::kYOff;
    auto storeY = [&](int k) {
        if (two_body && colv) {
            float* y = Yb + (size_t)k * 6 * kMO + d;
            y[0] = Yt.x; y[kMO] = Yt.y; y[2 * kMO] = Yt.z; y[3 * kMO] = Yf.x; y[4 * kMO] = Yf.y; y[5 * kMO] = Yf.z;
        }
    };
    storeY(0);
    int kcur = 0;
    auto advance = [&]() {                          // the forward sweep's step to body kcur + 1: Y_kcur -> Y_kcur+1
        kcur++;
        if (dofl) {
            const int b = kcur;
            Yf = Yf + cross(Yt, ld3(L.r[b]));
            const float u = Mrow[6 + b - 1];
            const float qdd = (u - (dot(ld3(L.Ua[b]), Yt) + dot(ld3(L.Ub[b]), Yf))) * L.Dinv[b];
            Yt = Yt + ld3(L.ax[b]) * qdd;
            Mrow[6 + b - 1] = qdd;
            if (b == kj) mden = qdd;
        }
        storeY(kcur);
    };
    // one contact: three rows from its geometry record and the Y of its body (and of the other body of a pair)
    struct Geo { float4 g[5]; };
    auto load_geo = [&](int ci) {
        const int slot = __builtin_amdgcn_readfirstlane(ci < L.nplane ? L.clist[ci] : LT::NC + (ci - L.nplane));
        const float4* g = reinterpret_cast<const float4*>(rows + LT::kGeoOff + (size_t)slot * LT::kGeo);
        Geo G;
#pragma unroll
        for (int i = 0; i < 5; i++) G.g[i] = g[i];
        return G;
    };
    const float spec0 = d == LT::kSpec ? 1.0f : 0.f, spec1 = d == LT::kSpec + 1 ? 1.0f : 0.f;    // the records' scalar columns
    // The solve resolves the normals two at a time (row_step_normal2): the second row's dot is taken from the same
    // delta-v as the first's and corrected by  c dI_first,  c = (J_2 / den_2) . (M^-1 J_1^T)  -- exactly the sequential
    // sweep.  c of an odd contact with its predecessor rides in the spare float of its impulse entry.
    float prevMn = 0.f;                             // M^-1 J^T of the previous contact's normal row (this lane's column)
    auto assemble = [&](int ci, const Geo& G, const f3 YtA, const f3 YfA, const f3 YtB, const f3 YfB, int kA, int kB,
                        auto two_c) {
        constexpr bool TWO = decltype(two_c)::value;       // a pair of bodies (link-link) or one body against the world
        const f3 P = mk3(G.g[0].x, G.g[0].y, G.g[0].z);
        const float dist = G.g[0].w;
        const f3 dA = mk3(G.g[1].x, G.g[1].y, G.g[1].z), dB = mk3(G.g[1].w, G.g[2].x, G.g[2].y);
        const f3 dn = mk3(G.g[2].z, G.g[2].w, G.g[3].x), PB = mk3(G.g[3].y, G.g[3].z, G.g[3].w);
        const float fsc = G.g[4].z, lam0 = G.g[4].w;     // lam0: where the normal row starts (warm starting; else 0)
        // J[d] = A_d . ((P - O_d) x dir) + B_d . dir = dir . (A_d x (P - O_d) + B_d), and likewise
        // M^-1 J^T [d] = Yt . ((P - o_k) x dir) + Yf . dir = dir . (Yt x (P - o_k) + Yf): one vector per contact and
        // lane for each, a dot product per row
        // (a body's reference point: its joint origin; the free box's: its centre)
        auto org = [&](int k) { return (TWO && k == LT::kBoxBody) ? ld3(L.box) : ld3(L.o[k <= N ? k : 0]); };
        f3 Cj = cross(Aj, P - Oj) + Bj;
        if (!moves(kA)) Cj = mk3(0, 0, 0);
        f3 Dj = cross(YtA, P - org(kA)) + YfA;
        if (TWO && kB >= 0) {                                           // wave-uniform: minus the same for the other body
            f3 Cb = cross(Aj, PB - Oj) + Bj;
            if (!moves(kB)) Cb = mk3(0, 0, 0);
            Cj = Cj - Cb;
            Dj = Dj - (cross(YtB, PB - org(kB)) + YfB);
        }
        float Jr[3], Mr[3], red[7];
#pragma unroll
        for (int kind = 0; kind < 3; kind++) {
            const f3 dir = kind == 0 ? dn : (kind == 1 ? dA : dB);
            const float j = dot(dir, Cj);
            float m = dot(dir, Dj);
            if (TWO && kind != 0) m *= fsc;                             // (ground contacts: scale 1)
            Jr[kind] = j; Mr[kind] = m;
            red[kind] = j * m; red[3 + kind] = j * vd;
        }
        red[6] = Jr[0] * prevMn;                                        // the normals' coupling with the previous contact
        // seven sums over the wave, their DPP steps interleaved (each instruction is the others' wait state)
        asm volatile(
            "s_nop 1\n\t"      // the operands may have been written by the instructions just before (VALU write -> DPP read)
            SNK_RED64x7_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
            SNK_RED64x7_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
            SNK_RED64x7_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
            SNK_RED64x7_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
            SNK_RED64x7_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
            SNK_RED64x7_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
            : [a] "+v"(red[0]), [b] "+v"(red[1]), [c] "+v"(red[2]), [d] "+v"(red[3]), [e] "+v"(red[4]), [f] "+v"(red[5]),
              [g] "+v"(red[6]));
        float cpl = 0.f;
        float Jo[3], Mo[3];
#pragma unroll
        for (int kind = 0; kind < 3; kind++) {
            const float dn_ = lane_bcast(red[kind], 63), rv_ = lane_bcast(red[3 + kind], 63);
            float rc = __builtin_amdgcn_rcpf(dn_);
            rc = rc * (2.0f - dn_ * rc);                                // one Newton step: within an ulp of 1 / den
            const float dinv = dn_ > 1.1920929e-7f ? rc : 0.f;
            float target = -rv_;
            if (kind == 0) {
                const float pen = dist + M.slop;
                target += pen > 0.f ? -pen * M.inv_dt : -pen * M.contact_erp * M.inv_dt;
            }
            // the record's columns (Lds<N, false>): J / den with -rhs and 0 in the pad columns; M^-1 J^T with 0 and den
            // (J and M^-1 J^T are zero in the pad columns as they come)
            Jo[kind] = dinv * (Jr[kind] - spec0 * target);
            Mo[kind] = Mr[kind] + spec1 * dn_;
            if (kind == 0) cpl = (ci & 1) ? dinv * lane_bcast(red[6], 63) : 0.f;
        }
        prevMn = Mo[0];
        if (colv) {
            // contacts 2p and 2p + 1 share a 640-byte record [J0 M0 J1 M1] per column: the solve resolves them in one
            // step (row_step_normal2) and fetches them with one 16-byte load per lane
            *reinterpret_cast<float2*>(rows + (size_t)(ci >> 1) * 2 * LT::kRS + 4 * d + 2 * (ci & 1)) = make_float2(Jo[0], Mo[0]);
            *reinterpret_cast<float4*>(rows + (size_t)(LT::kFric + 2 * ci) * LT::kRS + 4 * d) = make_float4(Jo[1], Jo[2], Mo[1], Mo[2]);
        }
        if (lane == 0) *reinterpret_cast<float4*>(L.acc[ci]) = make_float4(lam0, 0.f, 0.f, cpl);
    };
    // ---- (c) ground contacts, in the order of their bodies: Y stays in registers
    const int nplane = L.nplane;
    {
        const f3 z3 = mk3(0, 0, 0);
        auto one = [&](int ci, const Geo& G) {
            const int kA = __builtin_amdgcn_readfirstlane((int)G.g[4].x);
            while (kcur < kA) advance();
            assemble(ci, G, Yt, Yf, z3, z3, kA, -1, std::false_type{});
        };
        if (nplane > 0 && sorted) {
            Geo ga = load_geo(0);                                       // two records in flight, no copies between them
            for (int ci = 0; ci < nplane; ci += 2) {
                const Geo gb = load_geo(ci + 1 < nplane ? ci + 1 : ci);
                one(ci, ga);
                if (ci + 1 < nplane) {
                    ga = load_geo(ci + 2 < nplane ? ci + 2 : ci + 1);
                    one(ci + 1, gb);
                }
            }
        }
    }
    while (kcur < N) advance();                     // (the rest of M^-1's rows: the motors need every joint's)
    // ---- (d) link-link and obstacle contacts: any two bodies, their Y from the block the sweep left behind
    if (two_body) {
        if (fbox && colv) {
            // Y of the box: a unit wrench (tau, f) about its centre moves its own six components only --
            // angular component i: row i of the world inverse inertia . tau, linear component i: f_i / m
            float* y = Yb + (size_t)LT::kBoxBody * 6 * kMO + d;
            const int i = d - ND;
            const float* W = L.bIw;
            f3 yt = mk3(0, 0, 0), yf = mk3(0, 0, 0);
            if (boxlane && i < 3) yt = i == 0 ? mk3(W[0], W[1], W[2]) : (i == 1 ? mk3(W[1], W[3], W[4]) : mk3(W[2], W[4], W[5]));
            if (boxlane && i >= 3) yf = mk3(i == 3 ? M.obs_minv : 0.f, i == 4 ? M.obs_minv : 0.f, i == 5 ? M.obs_minv : 0.f);
            y[0] = yt.x; y[kMO] = yt.y; y[2 * kMO] = yt.z; y[3 * kMO] = yf.x; y[4 * kMO] = yf.y; y[5 * kMO] = yf.z;
        }
        own_stores_visible();
        lds_sync();
        for (int ci = sorted ? nplane : 0; ci < nc; ci++) {
            const Geo G = load_geo(ci);
            const int kA = __builtin_amdgcn_readfirstlane((int)G.g[4].x), kB = __builtin_amdgcn_readfirstlane((int)G.g[4].y);
            auto ldY = [&](int k, f3& yt, f3& yf) {
                const float* y = Yb + (size_t)k * 6 * kMO + (colv ? d : 0);
                yt = colv ? mk3(y[0], y[kMO], y[2 * kMO]) : mk3(0, 0, 0);
                yf = colv ? mk3(y[3 * kMO], y[4 * kMO], y[5 * kMO]) : mk3(0, 0, 0);
            };
            f3 ytA, yfA, ytB = mk3(0, 0, 0), yfB = mk3(0, 0, 0);
            ldY(kA, ytA, yfA);
            if (kB >= 0) ldY(kB, ytB, yfB);
            assemble(ci, G, ytA, yfA, ytB, yfB, kA, kB, std::true_type{});
        }
    }
    lds_sync();
    // ---- (e) non-contact rows: violated joint limits first, then the n motors
    // (btMultiBodyJointLimitConstraint, btMultiBodyJointMotor [U])
    const float mden_j = __shfl(mden, lane + 6);    // motor / joint `lane`: its denominator sits in lane 6 + lane
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
        nlim = __popcll(bal);
        if (viol) {
            int idx = __popcll(bal & ((1ull << lane) - 1ull));
            float den = mden_j;
            float dinv = den > 1.1920929e-7f ? 1.0f / den : 0.f;
            float rel = sgn * L.qd()[lane];
            L.nc_joint[idx] = lane; L.nc_sign[idx] = sgn;
            L.nc_rhs[idx] = (-rel + (-pen) * M.limit_erp * M.inv_dt) * dinv;
            L.nc_dinv[idx] = dinv; L.nc_den[idx] = den;
            L.nc_lo[idx] = 0.f; L.nc_hi[idx] = M.limit_max; L.nc_app[idx] = 0.f;
        }
        if (lane < N) {
            int idx = nlim + lane;
            float den = mden_j;
            float dinv = den > 1.1920929e-7f ? 1.0f / den : 0.f;
            float cur = L.qd()[lane];
            float want = M.kp * (L.targets[lane] - L.q()[lane]) * M.inv_dt + cur + M.kd * (0.f - cur);
            L.nc_joint[idx] = lane; L.nc_sign[idx] = 1.f;
            L.nc_rhs[idx] = (want - cur) * dinv;
            L.nc_dinv[idx] = dinv; L.nc_den[idx] = den;
            L.nc_lo[idx] = -M.max_motor_imp; L.nc_hi[idx] = M.max_motor_imp; L.nc_app[idx] = 0.f;
        }
    }
    n_noncontact = nlim + N;
    lds_sync();
}

// ----------------------------------------------------------------------------------
// S6: projected Gauss-Seidel for chains too long for the register-resident solve (the 32-link
// config: 38 velocity components, up to 384 contact rows kept in LDS), lane = velocity component
// (btMultiBodyConstraintSolver::solveSingleIteration / resolveSingleConstraintRowGeneric /
//  resolveConeFrictionConstraintRows [U]).  Returns delta-v of this lane.
//
// Same lessons as the 16-link solve (snk_pgs_v2.hpp): the row steps are hand-written with the
// fewest VALU instructions (one multiply + a 6-step DPP reduction per dot, the clamp on
// wave-uniform values, J rows pre-divided by their denominator), a row's operands are read
// from LDS one row ahead of their use, and the motor rows -- unit Jacobians -- need no
// reduction at all.
// ----------------------------------------------------------------------------------
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

// Row steps of the streamed-row solve.  jv: the row's J half (J / den, -rhs in lane kSpec, where delta-v holds 1), mv: its
// M^-1 J^T half (den in lane kSpec + 1), both one value per lane; the dot  s = (J.dv)/den - rhs  comes out of one
// multiply and a 6-step DPP reduction, the clamp runs on wave-uniform values, and |M^-1 J^T dI| carries the row's
// residual |dI den| in lane kSpec + 1 (collected per lane in lsq, read once per iteration).
// a contact-normal row: a' = max(a - s, 0); dv += M^-1 J^T (a' - a).  Returns a'.  14 VALU.
template <int SUM_LANE>
__device__ __forceinline__ float row_step_normal(float jv, float mv, float acc, float& dv, float& lsq) {
    float t, x, P, s;
    asm volatile(
        "v_mul_f32 %[t], %[jv], %[dv]\n\t"
        "s_nop 1\n\t"
        SNK_RED64("%[t]")
        "s_nop 0\n\t"
        "v_readlane_b32 %[s], %[t], %[SL]\n\t"
        "s_nop 1\n\t"
        "v_subrev_f32 %[x], %[s], %[acc]\n\t"
        "v_max_f32 %[x], 0, %[x]\n\t"
        "v_sub_f32 %[t], %[x], %[acc]\n\t"
        "v_mul_f32 %[P], %[t], %[mv]\n\t"
        "v_add_f32 %[dv], %[dv], %[P]\n\t"
        "v_max_f32_e64 %[lsq], %[lsq], |%[P]|\n\t"
        : [t] "=&v"(t), [x] "=&v"(x), [P] "=&v"(P), [s] "=&s"(s), [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [jv] "v"(jv), [mv] "v"(mv), [acc] "v"(acc), [SL] "n"(SUM_LANE));
    return x;
}

// Two consecutive contact normals: both dots from the same delta-v, their reductions interleaved (one s_nop per stage
// instead of two per stage and row), the second row's sum corrected by  c dI_first  (c: see build_rows_v1).  28 VALU,
// 36 issue slots for the two rows against 2 x 30.
template <int SUM_LANE>
__device__ __forceinline__ void row_step_normal2(float jA, float mA, float jB, float mB, float& accA, float& accB, float c,
                                                 float& dv, float& lsq) {
    float tA, tB, xA, xB, dA, sA, sB;
    asm volatile(
        "v_mul_f32 %[tA], %[jA], %[dv]\n\t"
        "v_mul_f32 %[tB], %[jB], %[dv]\n\t"
        "s_nop 0\n\t"
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        "v_add_f32_dpp %[tA], %[tA], %[tA] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %[tB], %[tB], %[tB] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[sA], %[tA], %[SL]\n\t"
        "v_readlane_b32 %[sB], %[tB], %[SL]\n\t"
        "s_nop 0\n\t"
        "v_subrev_f32 %[xA], %[sA], %[accA]\n\t"
        "v_subrev_f32 %[xB], %[sB], %[accB]\n\t"
        "v_max_f32 %[xA], 0, %[xA]\n\t"
        "v_sub_f32 %[dA], %[xA], %[accA]\n\t"
        "v_fma_f32 %[xB], -%[c], %[dA], %[xB]\n\t"
        "v_max_f32 %[xB], 0, %[xB]\n\t"
        "v_mul_f32 %[tA], %[dA], %[mA]\n\t"
        "v_sub_f32 %[dA], %[xB], %[accB]\n\t"
        "v_add_f32 %[dv], %[dv], %[tA]\n\t"
        "v_mul_f32 %[tB], %[dA], %[mB]\n\t"
        "v_add_f32 %[dv], %[dv], %[tB]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[tA]|, |%[tB]|\n\t"
        : [tA] "=&v"(tA), [tB] "=&v"(tB), [xA] "=&v"(xA), [xB] "=&v"(xB), [dA] "=&v"(dA), [sA] "=&s"(sA), [sB] "=&s"(sB),
          [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [jA] "v"(jA), [mA] "v"(mA), [jB] "v"(jB), [mB] "v"(mB), [accA] "v"(accA), [accB] "v"(accB), [c] "v"(c),
          [SL] "n"(SUM_LANE));
    accA = xA;
    accB = xB;
}

// Bullet's cone-friction pair of one contact: both dots from the same delta-v, the new pair (a - s) projected
// radially onto the disc of radius lim.  31 VALU.
template <int SUM_LANE>
__device__ __forceinline__ void row_step_cone(float jA, float mA, float jB, float mB, float& accA, float& accB, float lim,
                                              float EPS, float& dv, float& lsq) {
    float tA, tB, xA, xB, r2, P, sA, sB;
    asm volatile(
        "v_mul_f32 %[tA], %[jA], %[dv]\n\t"
        "v_mul_f32 %[tB], %[jB], %[dv]\n\t"
        "s_nop 0\n\t"
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        SNK_RED64x2_STEP("%[tA]", "%[tB]", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        "v_add_f32_dpp %[tA], %[tA], %[tA] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %[tB], %[tB], %[tB] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[sA], %[tA], %[SL]\n\t"
        "v_readlane_b32 %[sB], %[tB], %[SL]\n\t"
        "s_nop 0\n\t"
        "v_subrev_f32 %[xA], %[sA], %[accA]\n\t"
        "v_subrev_f32 %[xB], %[sB], %[accB]\n\t"
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
        "v_mul_f32 %[r2], %[tB], %[mB]\n\t"
        "v_add_f32 %[dv], %[dv], %[P]\n\t"
        "v_add_f32 %[dv], %[dv], %[r2]\n\t"
        "v_max3_f32 %[lsq], %[lsq], |%[P]|, |%[r2]|\n\t"        // the two rows' residuals separately (their sum could cancel)
        : [tA] "=&v"(tA), [tB] "=&v"(tB), [xA] "=&v"(xA), [xB] "=&v"(xB), [r2] "=&v"(r2), [P] "=&v"(P), [sA] "=&s"(sA),
          [sB] "=&s"(sB), [dv] "+v"(dv), [lsq] "+v"(lsq)
        : [jA] "v"(jA), [mA] "v"(mA), [jB] "v"(jB), [mB] "v"(mB), [accA] "v"(accA), [accB] "v"(accB), [lim] "v"(lim),
          [EPS] "v"(EPS), [SL] "n"(SUM_LANE));
    accA = xA;
    accB = xB;
}

// INPLACE: the copy that runs inside the register-resident kernels for their rare substeps (substep()): it keeps rounds
// 2-3's 32 / 32 / 16 registers' split.  With the standalone kernels' 40 / 16 / 16 (round 4: +2 % for 32 links, +7 % for the
// 16-link streamed-row kernels) the register-resident kernel around it came out 3.6 % slower -- 352 k against 365 k
// env-steps/s with 19 such substeps in 80 000 -- for the registers live across the call.
template <class LT, bool INPLACE = false>
__device__ float pgs_v1(LT& L, const DevModel& M, int lane, int nc, int nn, float mu, int& iters,
                        float* __restrict__ rows) {
    constexpr int N = LT::kN;
    constexpr int ND = N + 6;
    static_assert(ND <= 64, "this solve is laid out for one row per 64-lane register");
    constexpr int kSpec = LT::kSpec;
    const bool act = lane < ND + ((N <= 16 && M.obstacle == 2) ? 6 : 0);      // (the free box's six components sit behind the snake's)
    const int nlim = nn - N;                       // violated joint limits come first in the non-contact list
    // model fields used inside the loops, read once (the model lives in global memory)
    // (v_readfirstlane: the loads go through vector memory, the values must be scalar for the
    //  branches on them to be scalar branches instead of exec-mask regions)
    const int n_iter = __builtin_amdgcn_readfirstlane(M.n_iter);
    const bool cone = __builtin_amdgcn_readfirstlane(M.cone) != 0;
    const float mi = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(M.max_motor_imp)));
    const float thr2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(M.resid_thr)));
    // motor rows: lane 6+j holds motor j's target velocity change and 1/den
    const bool mot = lane >= 6 && act;
    const int jm = mot ? lane - 6 : 0;
    const float DINVV = mot ? L.nc_dinv[nlim + jm] : 0.f;
    const float DENV = mot ? L.nc_den[nlim + jm] : 0.f;
    const float TARGV = (mot && DINVV > 0.f) ? L.nc_rhs[nlim + jm] * DENV : 0.f;
    float ACCV = 0.f;                               // accumulated motor impulses, motor j in lane 6+j
    const float EPS = 1e-30f;
    // Row operands come from global memory through register rings (kRingN normals / kRingF friction pairs in flight), a
    // contact's record requested a ring's depth before it is used (the next trip's loads are
    // issued one by one as this trip's slots are consumed).  Lane d < ND reads column d of a row, lanes kSpec and
    // kSpec + 1 the row's scalars (160 contiguous bytes per half row: coalesced); the other lanes sit the solve out.
    // The rows and impulses of the contacts between nc and the end of the last group are zeroed: resolving them
    // changes nothing (dI = 0 exactly).
    constexpr int kRN = INPLACE ? SNK_IP_RINGN : LT::kRingN;      // normals in flight
    // contacts are resolved in groups of 8 behind one scalar branch: the rows between nc and the next multiple of 8
    // are zeroed (inert), a ring trip ends at that multiple instead of running its full depth (round 1 padded
    // to a whole trip: 144 contacts -- 128 on the ground + 16 link-link -- cost 160)
    const int nc_pad = __builtin_amdgcn_readfirstlane((nc + 7) / 8 * 8);
    {
        // (normals in pairs: an odd count leaves the last contact's partner inside its record -- its two floats per column --,
        //  then whole records)
        const int nce = (nc + 1) & ~1;
        if ((nc & 1) && lane < LT::kMO)
            *reinterpret_cast<float2*>(rows + (size_t)(nc >> 1) * 2 * LT::kRS + 4 * lane + 2) = make_float2(0.f, 0.f);
        float* z = rows + (size_t)nce * LT::kRS;
        float* zf = rows + (size_t)(LT::kFric + 2 * nc) * LT::kRS;
        const int nz = (nc_pad - nc) * LT::kRS;
        for (int i = lane; i < (nc_pad - nce) * LT::kRS; i += 64) z[i] = 0.f;
        for (int i = lane; i < 2 * nz; i += 64) zf[i] = 0.f;
        for (int i = lane; i < 4 * (nc_pad - nc); i += 64) L.acc[nc][i] = 0.f;
    }
    own_stores_visible();     // the rows were written lane = row, they are read lane = column
    lds_sync();
    constexpr int kRS = LT::kRS;
    // Addressing: a record's address is a wave-uniform base (scalar arithmetic) plus the lane's column, 4 * lane bytes,
    // so a load is  global_load_dword v, v_column, s[base:base+1] offset:imm  with no vector arithmetic at all
    // (round-2 measurement: per-lane 64-bit pointers cost 4 VALU per friction step, a tenth of its issue slots).
    // For that every lane that takes part must stride alike, so the solve runs with lanes 0 .. kMO - 1 only (the
    // 38 velocity components and the two scalar columns); the sum of a row's products lands in lane kMO - 1.
    constexpr unsigned kRecB = kRS * 4, kHalfB = LT::kMO * 4;        // bytes per record, offset of its M^-1 J^T half
    constexpr unsigned kFricB = (unsigned)LT::kFric * kRecB;          // the first friction pair
    // (buffer loads: resource descriptor and record offset in SGPRs, the column in a VGPR, the half in the immediate)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(rows, 0, (int)(LT::kRowFloats * sizeof(float)), 0x00020000);
    const int vcol = 4 * lane;
    auto ldJ = [&](unsigned rec_bytes) {             // a column of a plain (motor) row
        return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vcol, (int)rec_bytes, 0));
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto ldN = [&](unsigned rec_bytes, float& j, float& m) {       // {J, M^-1 J^T} of a normal's record
        // (rec_bytes = contact x kRecB as before; inside its pair's record the contact's two floats sit at 16 d + 8 (ci & 1))
        const unsigned odd = (rec_bytes / kRecB) & 1u;
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, 4 * vcol, (int)(rec_bytes - odd * kRecB + odd * 8u), SNK_V1_LDAUX);
        j = __uint_as_float(v.x); m = __uint_as_float(v.y);
    };
    auto ldN2 = [&](unsigned rec_bytes, float& j0, float& m0, float& j1, float& m1) {      // rec_bytes: of the EVEN contact
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 4 * vcol, (int)rec_bytes, SNK_V1_LDAUX);
        j0 = __uint_as_float(v.x); m0 = __uint_as_float(v.y); j1 = __uint_as_float(v.z); m1 = __uint_as_float(v.w);
    };
    auto ldF = [&](unsigned rec_bytes, float& ja, float& jb, float& ma, float& mb) {    // a friction pair's record
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 4 * vcol, (int)rec_bytes, SNK_V1_LDAUX);
        ja = __uint_as_float(v.x); jb = __uint_as_float(v.y); ma = __uint_as_float(v.z); mb = __uint_as_float(v.w);
    };
    // (Round 5 tried the solve with all 64 lanes enabled -- neutral -- and further resident rows in the upper 24 lanes of
    //  the resident registers, fetched with ds_bpermute: bit-identical and 2 % SLOWER, a stashed row costs more issue
    //  slots than the stream it saves; and one reduction for a step's two dots (v_permlane32_swap, 17 % fewer VALU
    //  instructions): no change -- the solve runs at the latency of each wave's chain of dependent row steps, not at a
    //  byte or an issue rate; profiles/r05_c32_stash_experiment.txt)
    float dv = lane == kSpec ? 1.0f : 0.f;        // lane kSpec: the constant that multiplies the rows' -rhs column
    int it = 0;
    if (lane < LT::kMO) {
    // the motors' M^-1 columns stay in registers for the whole solve (they are read 50 x n times)
    float RMm[N];
#pragma unroll
    for (int j = 0; j < N; j++) RMm[j] = ldJ((unsigned)(LT::kMmOff * 4) + (unsigned)(6 + j) * kHalfB);   // columns >= ND: zero
    // ... and so do the normal rows of the first kResN contacts: the registers the rings leave free hold an eighth of
    // the stream (the kernel is bound by that stream)
    constexpr int kResN = INPLACE ? SNK_IP_RESN : LT::kResN;
    // What the build-time knobs (SNK_V1_* / SNK_IP_*) must satisfy for the loops below to stay inside the records: rows
    // come in pairs (ldN2), the early exits and the ring's refill address are evaluated at every eighth row only, the
    // first ring trip starts behind the resident rows.  A sweep value such as 36 or 44 would compile and read beyond
    // nc_pad.
    static_assert(kResN % 8 == 0 && kResN > 0, "resident normal rows: a multiple of 8 (groups of 8 behind one scalar branch)");
    static_assert(kRN % 8 == 0 && kRN > 0, "normal rows in flight: a multiple of 8 (refill address taken at k % 8 == 0)");
    static_assert(kResN + kRN <= LT::NCT, "resident + in-flight normal rows exceed the contact slots of this layout");
    static_assert((INPLACE ? SNK_IP_RINGF : LT::kRingF) % 8 == 0, "friction pairs in flight: a multiple of 8");
    float RNJ[kResN], RNM[kResN];
#pragma unroll
    for (int k = 0; k < kResN; k += 2) ldN2((unsigned)k * kRecB, RNJ[k], RNM[k], RNJ[k + 1], RNM[k + 1]);
    if (__builtin_amdgcn_readfirstlane(M.warm_start)) {
        // warm starting: the normal rows start at the impulses build_rows_v1 took from the contact cache, delta-v at the
        // sum of M^-1 J^T of those (the scalar columns' part of that sum is dropped again: lane kSpec stays 1, and
        // lane kSpec + 1 is only ever read through lsq)
        for (int ci = 0; ci < nc; ci++) {
            const float a = L.acc[ci][0];
            if (__builtin_amdgcn_readfirstlane(a != 0.f ? 1 : 0)) {
                float jj, mm;
                ldN((unsigned)ci * kRecB, jj, mm);
                dv += act ? a * mm : 0.f;
            }
        }
    }
    for (; it < n_iter; it++) {
        float lsq = 0.f;       // per lane max |M^-1 J^T dI| of the contact rows: lane kSpec + 1 holds max |dI * den|
        float lsq_nc = 0.f;    // max |dI * den| of the limit and motor rows
        auto limit_rows = [&](bool fwd) {
            for (int jj = 0; jj < nlim; jj++) {
                const int idx = fwd ? jj : nlim - 1 - jj;
                const int j = __builtin_amdgcn_readfirstlane(L.nc_joint[idx]);
                const float sg = L.nc_sign[idx];
                float un = sg * lane_bcast(dv, 6 + j);
                float a0 = L.nc_app[idx];
                float dI = L.nc_rhs[idx] - un * L.nc_dinv[idx];
                float sum = fminf(fmaxf(a0 + dI, L.nc_lo[idx]), L.nc_hi[idx]);
                dI = sum - a0;
                L.nc_app[idx] = sum;   // uniform value, every lane stores it: no barrier needed
                const float mv = ldJ((unsigned)(LT::kMmOff * 4) + (unsigned)(6 + j) * kHalfB);      // a violated limit is rare
                dv += sg * mv * dI;
                lsq_nc = fmaxf(lsq_nc, fabsf(dI * L.nc_den[idx]));
            }
        };
        auto motor_rows = [&](auto fwd_c) {
            constexpr bool fwd = decltype(fwd_c)::value;
            float Uv = 0.f;                                              // lane 6+j: the dI motor j got in this sweep
#pragma unroll
            for (int jj = 0; jj < N; jj++) {
                const int j = fwd ? jj : N - 1 - jj;
                float u = (TARGV - dv) * DINVV;                          // every motor's candidate dI, lane-local (DINVV = 0 beyond the joints)
                if (mi < 1e30f) u = fminf(fmaxf(ACCV + u, -mi), mi) - ACCV;   // (a lane's ACCV only matters at its own step)
                const float sdI = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), 6 + j));
                Uv = (lane == 6 + j) ? u : Uv;
                dv += sdI * RMm[j];                                      // zero beyond the velocity components
            }
            ACCV += Uv;
            lsq_nc = fmaxf(lsq_nc, cols_max<LT::kMO - 1>(fabsf(Uv * DENV)));
        };
        if (it & 1) { limit_rows(true); motor_rows(std::true_type{}); }
        else { motor_rows(std::false_type{}); limit_rows(false); }
        if (nc > 0) {
            // normals: slot k of the ring holds contact (trip base + k).  A slot is refilled with
            // the contact kRingN further on as soon as it has been consumed -- if there is one: round 1 refilled
            // unconditionally, and with the usual 128 contacts the last trip of each phase fetched 32 (16) contacts'
            // worth of rows nobody used, 14 % of the stream this kernel is bound by; the contact's
            // accumulated impulse comes from LDS one step ahead.
            float jr[kRN], mr[kRN];
#pragma unroll
            for (int k = 0; k < kRN; k += 2) ldN2((unsigned)(kResN + k) * kRecB, jr[k], mr[k], jr[k + 1], mr[k + 1]);
            // two contacts per step (row_step_normal2): {impulse of the even one, impulse and coupling of the odd one} come
            // from LDS one step ahead
            float a0n = L.acc[0][0];
            float2 a1n = make_float2(L.acc[1][0], L.acc[1][3]);
#pragma unroll
            for (int k = 0; k < kResN; k += 2) {                                // the resident rows
                if ((k & 7) == 0 && k > 0 && k >= nc_pad) break;                // wave-uniform
                float a0 = a0n, a1 = a1n.x;
                const float c1 = a1n.y;
                a0n = L.acc[k + 2][0];
                a1n = make_float2(L.acc[k + 3][0], L.acc[k + 3][3]);
                row_step_normal2<LT::kMO - 1>(RNJ[k], RNM[k], RNJ[k + 1], RNM[k + 1], a0, a1, c1, dv, lsq);
                L.acc[k][0] = a0;
                L.acc[k + 1][0] = a1;
            }
            unsigned rb = 0;           // record the current group of eight refills counts from (wave-uniform)
            for (int base = kResN; base < nc_pad; base += kRN) {
#pragma unroll
                for (int k = 0; k < kRN; k += 2) {
                    if ((k & 7) == 0 && k > 0 && base + k >= nc_pad) break;      // wave-uniform
                    float a0 = a0n, a1 = a1n.x;
                    const float c1 = a1n.y;
                    a0n = L.acc[base + k + 2][0];
                    a1n = make_float2(L.acc[base + k + 3][0], L.acc[base + k + 3][3]);
                    row_step_normal2<LT::kMO - 1>(jr[k], mr[k], jr[k + 1], mr[k + 1], a0, a1, c1, dv, lsq);
                    L.acc[base + k][0] = a0;
                    L.acc[base + k + 1][0] = a1;
                    // the refill: the contacts kRN further on if there are any, else this trip's once more
                    // (a cache hit instead of a fetch of rows nobody uses; no branch, the load is issued either way).
                    // Issued after the step: the register pairs are free then and take the new records as they are
                    if ((k & 7) == 0) rb = (unsigned)((base + kRN + k < nc_pad) ? base + kRN : base) * kRecB;
                    ldN2(rb + (unsigned)k * kRecB, jr[k], mr[k], jr[k + 1], mr[k + 1]);
                }
            }
            if (cone) {
                // friction pairs: half a ring of contacts in flight (four vectors per contact).
                // A pair whose contact carries no normal impulse and no friction impulse yet resolves to exactly
                // nothing (the disc it is projected onto has radius 0: a' = a = 0, dI = 0) -- typically one contact in
                // six.  Its step is skipped and its record is not fetched: `live`, one bit per contact, is fixed for the
                // whole phase (the normal impulses are this iteration's final ones, a pair's own impulses only change
                // at its own step), built with four ballots and kept in SGPRs, so a step's test is scalar.
                constexpr int kC = INPLACE ? SNK_IP_RINGF : LT::kRingF;
                // (the solve runs on lanes 0 .. kMO - 1: one ballot covers kMO contacts; kP ballots, kW 64-bit words)
                constexpr int kP = (LT::NCT + LT::kMO - 1) / LT::kMO, kW = (kP * LT::kMO + 63) / 64;
                unsigned long long mw[kW + 1];
#pragma unroll
                for (int w = 0; w <= kW; w++) mw[w] = 0ull;
#pragma unroll
                for (int p = 0; p < kP; p++) {
                    const int c = LT::kMO * p + lane;
                    bool lv = false;
                    if (c < nc_pad) {
                        const float4 a = *reinterpret_cast<const float4*>(L.acc[c]);
                        lv = (mu * a.x > 0.f) || (a.y != 0.f) || (a.z != 0.f);     // (-0 counts as zero: a pair projected onto radius 0)
                    }
                    const unsigned long long b = __ballot(lv);
                    const int w = (LT::kMO * p) / 64, sh = (LT::kMO * p) % 64;       // compile-time after unrolling
                    mw[w] |= b << sh;
                    if (sh + LT::kMO > 64) mw[w + 1] |= b >> (64 - sh);
                }
                // (the register is shifted down by kC contacts per trip)
                constexpr unsigned kZeroB = (unsigned)(LT::kRows - 3) * kRecB;   // 960 bytes of zeros: the refill of a skipped pair
                float jA[kC], jB[kC], mA[kC], mB[kC];
                {
                    const unsigned long long l0 = mw[0];
#pragma unroll
                    for (int k = 0; k < kC; k++)
                        ldF((l0 >> k) & 1ull ? kFricB + (unsigned)k * 2u * kRecB : kZeroB, jA[k], jB[k], mA[k], mB[k]);
                }
                float4 fn = *reinterpret_cast<const float4*>(L.acc[0]);
                for (int base = 0; base < nc_pad; base += kC) {
                    unsigned long long lv = mw[0];                  // bits 0 .. kC - 1: this trip, kC .. 2 kC - 1: the next
                    static_assert(2 * kC <= 64, "two trips' bits in one word");
#pragma unroll
                    for (int w = 0; w < kW; w++) mw[w] = (mw[w] >> kC) | (mw[w + 1] << (64 - kC));      // (mw[kW] stays 0)
#pragma unroll
                    for (int k = 0; k < kC; k++) {
                        if ((k & 7) == 0 && k > 0 && base + k >= nc_pad) break;  // wave-uniform
                        const float4 c = fn;
                        fn = *reinterpret_cast<const float4*>(L.acc[base + k + 1]);
                        asm volatile("" : "+s"(lv));         // the test stays a scalar bit test here (hoisted, sixteen lane masks spill SGPRs)
                        if ((lv >> k) & 1ull) {
                            float aA = c.y, aB = c.z;
                            row_step_cone<LT::kMO - 1>(jA[k], mA[k], jB[k], mB[k], aA, aB, mu * c.x, EPS, dv, lsq);
                            *reinterpret_cast<float2*>(&L.acc[base + k][1]) = make_float2(aA, aB);
                        }
                        // the refill, issued after the step (the four registers are free then): the pair kC further on
                        ldF((lv >> (kC + k)) & 1ull ? kFricB + (unsigned)(base + kC + k) * 2u * kRecB : kZeroB, jA[k], jB[k], mA[k], mB[k]);
                    }
                }
            } else {
                // pyramid friction (not Bullet's default here): box-clamped rows, one after the other
                for (int ci = 0; ci < nc; ci++) {
                    const unsigned o = kFricB + (unsigned)ci * 2u * kRecB;
                    float* ac = L.acc[ci];
                    const float lim = mu * ac[0];
                    if (!(lim > 0.f)) continue;
                    const float accA = ac[1], accB = ac[2];
                    float fjA, fjB, fmA, fmB;
                    ldF(o, fjA, fjB, fmA, fmB);
                    const float sA = fminf(fmaxf(accA - cols_sum<LT::kMO - 1>(fjA * dv), -lim), lim);
                    const float PA = fmA * (sA - accA);
                    dv += PA;
                    const float sB = fminf(fmaxf(accB - cols_sum<LT::kMO - 1>(fjB * dv), -lim), lim);
                    const float PB = fmB * (sB - accB);
                    dv += PB;
                    ac[1] = sA; ac[2] = sB;
                    lsq = fmaxf(lsq, fmaxf(fabsf(PA), fabsf(PB)));
                }
            }
        }
        const float res = fmaxf(lane_bcast(lsq, kSpec + 1), lsq_nc);
        if (res * res <= thr2 || it >= n_iter - 1) { it++; break; }
    }
    }
    it = __builtin_amdgcn_readfirstlane(it);
    if (lane >= 6 && lane < 6 + N) L.nc_app[nlim + lane - 6] = ACCV;
    lds_sync();
    iters = it;
    return act ? dv : 0.f;
}

// ----------------------------------------------------------------------------------
// one physics substep
// ----------------------------------------------------------------------------------
template <class LT, bool INPLACE = false>
__device__ __forceinline__ void substep_v1(LT& L, const DevModel& M, int lane, float mu, int& iters, int& ncontacts, float* __restrict__ rows,
                           const SensorHint& hint, float* __restrict__ mf, unsigned long long* __restrict__ ovf) {
    constexpr int N = LT::kN;
    constexpr int ND = N + 6;
    const float dt = M.dt;
#ifdef SNK_PROFILE
    unsigned long long prof_t[10];
#endif
    SNK_STAMP(0)
    const bool fbox = N <= 16 && M.obstacle == 2;         // the obstacle as a free body (16-link chain only): six more velocity components (lanes ND ..)
    if (fbox) box_frame_v1(L, M, lane);
    // (1) contacts of the current pose, (2) bias forces with gravity, joint damping torque
    int nc = find_contacts_v1(L, M, lane, rows, mf, ovf);
    SNK_STAMP(1)
    if (lane == 0) L.nplane = nc;
    const int nplane = nc;
    if (M.self_collision || M.obstacle)
        nc += find_self_contacts_v1(L, M, lane, mu, rows, ovf);   // link-link and obstacle contacts follow the ground's
    if (fbox) nc += find_box_ground_v1(L, M, lane, mu, nc - nplane, rows, ovf);     // ... and the box's own with the ground
    ncontacts = nc;
    own_stores_visible(); // contact geometry: written lane = slot, read lane = row
    lds_sync();
    SNK_STAMP(2)
    if (lane < N) {
        float qd = L.qd()[lane];
        L.qd_old[lane] = qd;
        L.tauj[lane] = -M.joint_damp * qd;   // PyBullet adds URDF joint damping as a torque [U]
    }
    body_bias<LT, true>(L, M, lane);
    lds_sync();
    aba_main<LT, true>(L, M, lane);
    SNK_STAMP(3)
    // joint-0 force sensor, first pass [U]: -zb . [m_r (a - g) + m_r v (k + k|v|)]
    f3 zb = mulRv(L.R[0], ld3(M.zbase));
    f3 v_old = ld3(L.base() + 10);
    float nv0 = sqrtf(dot(v_old, v_old));
    f3 a1 = ld3(&L.acc0[3]);
    float fz = -dot(zb, (a1 - mk3(0.f, 0.f, M.gz)) * M.m_root + v_old * (M.m_root * (M.lin_damp + M.lin_damp * nv0)));
    // reaction through the first motor joint (Bullet joint 3, snake_gait_test.py:33-40): what body 0 does not use up
    // of the forces on it, Newton on body 0 alone:  F = -(m_0 (a_0 + alpha_0 x c_0) + own force bias), z of body 1
    auto joint1_fz = [&]() {
        const f3 al0 = ld3(&L.acc0[0]), a0l = ld3(&L.acc0[3]);
        const f3 F = -((a0l + cross(al0, ld3(L.cw[0]))) * M.mass[0] + ld3(&L.p[0][3]));
        return F.x * L.R[1][2] + F.y * L.R[1][5] + F.z * L.R[1][8];
    };
    float fz3 = joint1_fz();
    // (3) v += a dt (clamped)
    if (lane < 6) {
        float x = L.base()[7 + lane] + L.acc0[lane] * dt;
        L.base()[7 + lane] = fminf(fmaxf(x, -M.max_vel), M.max_vel);
    } else if (lane < ND) {
        float x = L.qd()[lane - 6] + L.qdd[lane - 6] * dt;
        L.qd()[lane - 6] = fminf(fmaxf(x, -M.max_vel), M.max_vel);
    } else if (fbox && lane < ND + 6) {
        // the free box, a btMultiBody without links [U]: gravity, the base's damping m v (k + k|v|), I w (k + k|w|),
        // the gyroscopic term; lanes ND .. ND + 2 its angular, ND + 3 .. ND + 5 its linear components
        const f3 om = ld3(L.box + 7), vl = ld3(L.box + 10);
        const float nw = sqrtf(dot(om, om)), nv = sqrtf(dot(vl, vl));
        // world inertia times omega: R diag(1 / iinv) R^T omega
        const f3 wl = mulRtv(L.bR, om);
        const f3 Iw = mulRv(L.bR, mk3(M.obs_iinv[0] > 0.f ? wl.x / M.obs_iinv[0] : 0.f, M.obs_iinv[1] > 0.f ? wl.y / M.obs_iinv[1] : 0.f,
                                      M.obs_iinv[2] > 0.f ? wl.z / M.obs_iinv[2] : 0.f));
        const f3 tq = -cross(om, Iw) - Iw * (M.ang_damp + M.ang_damp * nw);
        const f3 al = mulSv(L.bIw, tq);
        const f3 a = mk3(0.f, 0.f, M.gz) - vl * (M.lin_damp + M.lin_damp * nv);
        const int i = lane - ND;
        const float acc = i == 0 ? al.x : (i == 1 ? al.y : (i == 2 ? al.z : (i == 3 ? a.x : (i == 4 ? a.y : a.z))));
        const float x = L.box[7 + i] + acc * dt;
        L.box[7 + i] = fminf(fmaxf(x, -M.max_vel), M.max_vel);
    }
    lds_sync();
    // (4) rows, (5) PGS
    int nn = 0;
    SNK_STAMP(4)
    build_rows_v1(L, M, lane, nc, nn, rows);
    SNK_STAMP(5)
    // the solve runs at a higher wave priority than everything around it (snk_pgs_v2.hpp: substep_v2)
    __builtin_amdgcn_s_setprio(3);
    float dv = pgs_v1<LT, INPLACE>(L, M, lane, nc, nn, mu, iters, rows);
    __builtin_amdgcn_s_setprio(0);
    SNK_STAMP(6)
    if (M.contact_model == 1 && lane < 2 * N) {
        // the normal impulses go back into the contact cache (btManifoldPoint::m_appliedImpulse [U]), write-through
        // like the cache itself
        int idx = L.cidx[2 * lane];
        const int mask = L.cidx[2 * lane + 1] >> 8;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((mask >> j) & 1) {
                const float a = L.acc[idx++][0];
                asm volatile("global_store_dword %0, %1, off sc1" : : "v"(mf + (size_t)lane * kMfFloats + 9 + 6 * j), "v"(a) : "memory");
            }
    }
    // only when this substep can be the last of its env-step (sensor_pass_needed)
    const bool sensor = sensor_pass_needed(L, M, lane, dv, hint);
    // delta-v crosses the sensor pass in LDS, not in a register (the pass is full of lane-dependent regions,
    // and a copy or a reload the allocator places inside one moves the active lanes only -- snk_pgs_v2.hpp has the
    // story; the register-resident substep's pass is a function of its own, here the value is parked): the solve is
    // over, so the non-contact rows' rhs column is free (the pass reads their joint, sign and impulse only)
    constexpr int kPark = 2 * N;        // = NL of this image (LdsCommon<N, 2 N>): 64 lanes for 32 links, 32 for 16
    static_assert(kPark >= N + 6 + (N <= 16 ? 6 : 0), "every lane with a velocity component has a slot");
    if (lane < kPark) L.nc_rhs[lane] = dv;
    if (sensor) {
        // (6) constraint pass for the joint-0 sensor [U]: ABA at the velocities after (3) with the
        // constraint forces as the only link forces, joint torques still applied
        if (lane <= N) {
            const int b = lane;
            f3 eN = mk3(0, 0, 0), eF = mk3(0, 0, 0);
            // contact slots of body b: cylinders 2b-1, 2b (body 0: cylinder 0) = slots 4b-2 .. 4b+1, in
            // contact order
            if (M.contact_model == 1) {
                // Bullet's manifolds: up to four points per cylinder, geometry stored by compact index;
                // cidx[2c], cidx[2c + 1] = first index and count of cylinder c (find_contacts_manifold_v1)
    #pragma unroll
                for (int cc = 0; cc < 2; cc++) {
                    const int c = 2 * b - 1 + cc;
                    if (c < 0 || c >= 2 * N) continue;
                    const int first = L.cidx[2 * c], cnt = L.cidx[2 * c + 1] & 0xff;
                    for (int j = 0; j < cnt; j++) {
                        const int ci = first + j;
                        const float* geo = rows + LT::kGeoOff + (size_t)ci * LT::kGeo;
                        f3 F = (mk3(0.f, 0.f, 1.f) * L.acc[ci][0] + ld3(geo + 4) * L.acc[ci][1] +
                                ld3(geo + 7) * L.acc[ci][2]) * M.inv_dt;
                        eF = eF + F;
                        eN = eN + cross(ld3(geo) - ld3(L.o[b]), F);
                    }
                }
            } else {
    #pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int slot = 4 * b - 2 + j;
                    const int ci = (slot >= 0 && slot < 4 * N) ? L.cidx[slot] : -1;
                    if (ci >= 0) {
                        const float* geo = rows + LT::kGeoOff + (size_t)slot * LT::kGeo;
                        f3 F = (mk3(0.f, 0.f, 1.f) * L.acc[ci][0] + ld3(geo + 4) * L.acc[ci][1] +
                                ld3(geo + 7) * L.acc[ci][2]) * M.inv_dt;
                        eF = eF + F;
                        eN = eN + cross(ld3(geo) - ld3(L.o[b]), F);
                    }
                }
            }
            // link-link contacts: equal and opposite forces on the two bodies (friction back from the solve's units)
            for (int ci = nplane; ci < nc; ci++) {
                const float* geo = rows + LT::kGeoOff + (size_t)(LT::NC + ci - nplane) * LT::kGeo;
                const int kA = (int)geo[16], kB2 = (int)geo[17];
                if (kA != b && kB2 != b) continue;
                const float fs = geo[18];
                const f3 F = (ld3(geo + 10) * L.acc[ci][0] + ld3(geo + 4) * (L.acc[ci][1] * fs) +
                              ld3(geo + 7) * (L.acc[ci][2] * fs)) * M.inv_dt;
                if (kA == b) { eF = eF + F; eN = eN + cross(ld3(geo) - ld3(L.o[b]), F); }
                if (kB2 == b) { eF = eF - F; eN = eN - cross(ld3(geo + 13) - ld3(L.o[b]), F); }
            }
            st3(L.ext(b), eN);
            st3(L.ext(b) + 3, eF);
        }
        if (lane < N) L.tauj[lane] = -M.joint_damp * L.qd_old[lane];
        lds_sync();
        if (lane == 0) {
            for (int i = 0; i < nn; i++) L.tauj[L.nc_joint[i]] += L.nc_sign[i] * L.nc_app[i] * M.inv_dt;
        }
        lds_sync();
        // velocities changed in (3): refresh w, v, zeta of every body (pose unchanged)
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
        {
            f3 v1 = ld3(L.base() + 10);
            float nv1 = sqrtf(dot(v1, v1));
            f3 a2 = ld3(&L.acc0[3]);
            fz += -dot(zb, a2 * M.m_root + v1 * (M.m_root * (M.lin_damp + M.lin_damp * nv1)));
            fz3 += joint1_fz();
        }
    }
    SNK_STAMP(7)
    if (fbox && lane == 0) {
        // the box's contacts with the ground are the last L.bmn-or-fewer of the list: their normal impulses go back
        // into its manifold (btManifoldPoint::m_appliedImpulse [U])
        int nb = 0;
        for (int ci = nplane; ci < nc; ci++) {
            const float* geo = rows + LT::kGeoOff + (size_t)(LT::NC + ci - nplane) * LT::kGeo;
            if ((int)geo[16] == LT::kBoxBody) { if (nb < 4) L.bman[6 * nb + 5] = L.acc[ci][0]; nb++; }
        }
    }
    // (7) apply the solver's delta-v (clamped), motor torques, integrate positions
    dv = lane < kPark ? L.nc_rhs[lane] : 0.f;
    if (lane < 6) {
        float x = L.base()[7 + lane] + dv;
        L.base()[7 + lane] = fminf(fmaxf(x, -M.max_vel), M.max_vel);
    } else if (lane < ND) {
        float x = L.qd()[lane - 6] + dv;
        x = fminf(fmaxf(x, -M.max_vel), M.max_vel);
        L.qd()[lane - 6] = x;
        L.q()[lane - 6] += dt * x;
    } else if (fbox && lane < ND + 6) {
        const float x = L.box[7 + lane - ND] + dv;
        L.box[7 + lane - ND] = fminf(fmaxf(x, -M.max_vel), M.max_vel);
    }
    if (lane < N) {
        // motor rows sit after the limit rows in the non-contact list
        L.taum()[lane] = L.nc_app[nn - N + lane] * M.inv_dt;
    }
    lds_sync();
    {
        float* bs = L.base();
        f3 om = ld3(bs + 7), vl = ld3(bs + 10);
        float fA = sqrtf(dot(om, om));
        const float kThr = 0.78539816339744831f;   // 0.5 * pi/2  [U] ANGULAR_MOTION_THRESHOLD
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
            L.fz3() = fz3;
        }
    }
    lds_sync();
    if (fbox) {
        // the box's pose: btMultiBody::stepPositionsMultiDof, the same exponential-map update as the snake's base [U]
        float* bx = L.box;
        const f3 om = ld3(bx + 7), vl = ld3(bx + 10);
        float fA = sqrtf(dot(om, om));
        const float kThr = 0.78539816339744831f;
        if (fA * dt > kThr) fA = kThr / dt;
        const float sc = fA < 0.001f ? 0.5f * dt - (dt * dt * dt) * 0.020833333333f * fA * fA : sinf(0.5f * fA * dt) / fA;
        const float dx = om.x * sc, dy = om.y * sc, dz = om.z * sc, dw = cosf(fA * dt * 0.5f);
        const float qx = bx[3], qy = bx[4], qz = bx[5], qw = bx[6];
        const float nw = dw * qw - dx * qx - dy * qy - dz * qz;
        const float nx = dw * qx + dx * qw + dy * qz - dz * qy;
        const float ny = dw * qy - dx * qz + dy * qw + dz * qx;
        const float nz = dw * qz + dx * qy - dy * qx + dz * qw;
        const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
        lds_sync();
        if (lane == 0) {
            bx[0] += dt * vl.x; bx[1] += dt * vl.y; bx[2] += dt * vl.z;
            bx[3] = nx * inv; bx[4] = ny * inv; bx[5] = nz * inv; bx[6] = nw * inv;
        }
        lds_sync();
    }
    SNK_STAMP(8)
    // pose of the new state: feeds checkSnakeHeight and the next substep
    fk_vel(L, M, lane);
    SNK_STAMP(9)
#ifdef SNK_PROFILE
    lds_sync();
    if (lane < 9) L.taum()[lane] = (float)(prof_t[lane + 1] - prof_t[lane]);
    lds_sync();
#endif
}

}  // namespace snk
#include "snk_pgs_v2.hpp"
#include "snk_freebox.hpp"
namespace snk {

// snk_contact_overflow's three counters, then the histogram of contact points per physics substep (kHistBins values;
// a 32-link snake's manifolds hold 256 ground points at most, plus 32 link-link / obstacle contacts)
constexpr int kOvfCounters = 3;
constexpr int kHistBins = 320;

// (defined with the record <-> LDS movers below)
template <class LT>
__device__ __forceinline__ void load_mf(LT& L, const float* __restrict__ mf, int lane);
template <class LT, bool THROUGH>
__device__ __forceinline__ void store_mf(LT& L, float* __restrict__ mf, int lane);

// One out-of-line copy of the streamed-row substep for the register-resident kernels' rare substeps (below): inlined
// there it would double those kernels; the streamed-row kernels themselves inline it (as a called function its LDS
// accesses go through flat addresses: -10 % on those kernels when the compiler chose that by itself, round 3).
#ifdef SNK_V1_INLINE
#define SNK_V1_CALL_ATTR __forceinline__
#else
#define SNK_V1_CALL_ATTR __noinline__
#endif
template <class LT>
__device__ SNK_V1_CALL_ATTR void substep_v1_call(LT& L, const DevModel& M, int lane, float mu, int& iters, int& ncontacts,
                                             float* __restrict__ rows, const SensorHint& hint, float* __restrict__ mf,
                                             unsigned long long* __restrict__ ovf) {
    substep_v1<LT, true>(L, M, lane, mu, iters, ncontacts, rows, hint, mf, ovf);
}

template <class LT>
__device__ __forceinline__ void substep(LT& L, const DevModel& M0, int lane_in, float mu, int& iters, int& ncontacts,
                                        const SensorHint& hint, float* __restrict__ rows, float* __restrict__ mf,
                                        unsigned long long* __restrict__ ovf) {
    int lane = lane_in;
    // Launder the model pointer once per substep: otherwise ~100 per-lane model constants are
    // hoisted out of the substep loop and stay live (or spilled) across the whole solve.
    const DevModel* Mq = &M0;
    asm volatile("" : "+s"(Mq));
    const DevModel& M = *Mq;
    // ... and the lane id: hundreds of per-lane LDS addresses are loop-invariant and would
    // otherwise be computed in the kernel prologue and spilled.
    asm volatile("" : "+v"(lane));
    if constexpr (LT::kV2) {
        substep_v2(L, M, lane, mu, iters, ncontacts, hint, ovf);
        if (ncontacts < 0) {
            // The contacts of this pose do not fit the register-resident solve's 64 slots (a snake at rest gathers up to
            // four points per cylinder: 128; Bullet has no limit).  Nothing has been touched yet: THIS substep goes through
            // the streamed-row solve of the same chain instead (128 + 32 slots: every point a 16-link snake's manifolds
            // can hold), in place -- the two LDS images share their first members (record, poses, ABA workspace), the
            // contact cache travels through its block of global memory, where the streamed-row kernels keep it.  The
            // rule is per substep and a function of the state alone, so results do not depend on the schedule, and the
            // single-substep API takes the same path.  Counted (snk_contact_overflow[0]), never silent.
            using L1 = Lds<LT::kN, false>;
            static_assert(sizeof(L1) <= sizeof(LT), "the streamed-row image must fit the register-resident one's allocation");
            L1& Lx = *reinterpret_cast<L1*>(&L);
            // The cache goes out with plain stores and this CU's vector L1 is invalidated behind them (what an agent-scope
            // acquire does: s_waitcnt vmcnt(0), buffer_inv sc1) before the streamed-row substep loads it, and again before it
            // comes back.  Round 4: with write-through (sc1) stores and no invalidate, the loads that followed could hit
            // lines this CU had cached when the environment was loaded -- an sc1 store does not refresh the storing CU's
            // own L1 -- and a few cylinders' manifolds came back one substep old: 64 against 65 contacts among replicas of
            // one state, whenever the line had survived (tools/dbg/replica_sub.py; it took other ring sizes in the
            // streamed solve, i.e. other timing, to show in test_schedule_does_not_change_results).
            store_mf<LT, false>(L, mf, lane);
            asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            if (M.poison) { lds_sync(); Lx.poison_own(lane); lds_sync(); }
            substep_v1_call(Lx, M, launder_lane(lane), mu, iters, ncontacts, rows, hint, mf, ovf);
            asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            load_mf(L, mf, lane);
        }
    } else {
        substep_v1(L, M, lane, mu, iters, ncontacts, rows, hint, mf, ovf);
    }
    // contact points of this substep, counted per value (snk_contact_histogram): what decides how many row slots a
    // register-resident solve needs.  One fire-and-forget atomic per substep (every lane with its own operand, folded
    // into one memory operation: see sched_pop for why there is no `if (lane == 0)`).
    // Off unless snk_contact_histogram_enable asked for it: always on it cost 0.7 % of the headline (A/B on one box, three
    // runs each: 361.1-362.1 k against 363.8-364.0 k env-steps/s).
#ifndef SNK_HIST_MODE
#define SNK_HIST_MODE 2
#endif
#if SNK_HIST_MODE == 2
    if (M.hist)
#elif SNK_HIST_MODE == 0
    if (false)
#endif
    {
        int bin = __builtin_amdgcn_readfirstlane(ncontacts);
        bin = bin < 0 ? 0 : (bin > kHistBins - 1 ? kHistBins - 1 : bin);
        atomicAdd(ovf + kOvfCounters + bin, lane_id() == 0 ? 1ull : 0ull);
    }
}

// ----------------------------------------------------------------------------------
// record <-> LDS, observation packing (snake.py:209-217)
// ----------------------------------------------------------------------------------
template <class LT>
__device__ __forceinline__ void load_rec(LT& L, const float* __restrict__ rec, int lane) {
    constexpr int N = LT::kN;
    lane = launder_lane(lane);
    for (int i = lane; i < LT::REC; i += 64) L.rec[i] = rec[i];
    lds_sync();
}
template <class LT>
__device__ __forceinline__ void store_rec(LT& L, float* __restrict__ rec, int lane) {
    constexpr int N = LT::kN;
    lane = launder_lane(lane);
    lds_sync();
    for (int i = lane; i < LT::REC; i += 64) rec[i] = L.rec[i];
}
// The same store, write-through (sc1): the record leaves this XCD's L2 for memory at once, so a wave on another
// XCD can take the env-step over after an agent-scope acquire without this wave writing its whole L2 back
// (MI355X_MICROARCH.md, inter-workgroup visibility: every handed-off byte stored sc1 and drained with
// s_waitcnt vmcnt(0) before the flag needs no agent release).  One 16-byte store per lane.
template <class LT>
__device__ __forceinline__ void store_rec_through(LT& L, float* __restrict__ rec, int lane) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    lane = launder_lane(lane);
    lds_sync();
    if (lane < LT::REC / 4) {
        const v4f v = reinterpret_cast<const v4f*>(L.rec)[lane];
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(rec + 4 * lane), "v"(v) : "memory");
    }
}
// The register-resident kernels keep an environment's contact manifolds (contact_model 1) in LDS while a wave holds
// it (Lds<N, true>::mfl); these move them from / to the environment's block of global memory
// ([2n][kMfFloats] = per cylinder [count, 3 pad, 4 x (a3, b.x, b.y, lambda)]) together with the state record.
// An environment's cache is one contiguous block: 2n cylinders x kMfFloats = 28 floats = 224 quads of 16 bytes (3.5 KB for 16
// links).  It travels in four wave-wide dwordx4 instructions, instruction k moving quads 64 k .. 64 k + 63: 1 KB of
// consecutive bytes, so every 128-byte line is written WHOLE by one store instruction -- the form MI355X_MICROARCH.md's
// hand-off table lists for write-through stores another XCD's wave then loads (a first version gave every lane the quads
// of "its" cylinder: lines assembled from four instructions' pieces, and results began to depend on where a launch
// handed env-steps over).  Quad Q holds floats 4 (Q % 7) .. + 3 of cylinder Q / 7's block [count, 3 pad, 4 x 6].
// (Round 3 moved only the LIVE floats, one write-through dword per lane and instruction: less payload, but every such store
// is a memory request of its own -- ~240 per hand-off, counted at 64 bytes each: 53 of the 82 MB that WRITE_SIZE showed per
// launch, and 42 of the 66 MB of FETCH_SIZE, were this (profiles/r04_write_size_suspects.txt).)  Slots beyond a
// cylinder's count are written as zeros, by every store alike, so what the block holds does not depend on the schedule.
template <class LT>
__device__ __forceinline__ void load_mf(LT& L, const float* __restrict__ mf, int lane) {
    if constexpr (LT::kV2) {
        if (mf) {
            lane = launder_lane(lane);
            typedef float v4f __attribute__((ext_vector_type(4)));
            constexpr int kQuads = 2 * LT::kN * kMfFloats / 4;
#pragma unroll
            for (int k = 0; k < (kQuads + 63) / 64; k++) {
                const int Q = 64 * k + lane;
                if (Q < kQuads) {
                    const v4f q = reinterpret_cast<const v4f*>(mf)[Q];
                    const int c = Q / 7, part = Q - 7 * c;
                    const float f[4] = {q.x, q.y, q.z, q.w};
                    if (part == 0) L.mfn[c] = (unsigned char)(f[0] < 0.f ? 0.f : (f[0] > 4.f ? 4.f : f[0]));
                    else {
#pragma unroll
                        for (int e = 0; e < 4; e++) L.mfl[4 * part - 4 + e][c] = f[e];
                    }
                }
            }
            lds_sync();
        }
    }
}
template <class LT, bool THROUGH>
__device__ __forceinline__ void store_mf(LT& L, float* __restrict__ mf, int lane) {
    if constexpr (LT::kV2) {
        if (mf) {
            lane = launder_lane(lane);
            lds_sync();
            typedef float v4f __attribute__((ext_vector_type(4)));
            constexpr int kQuads = 2 * LT::kN * kMfFloats / 4;
#pragma unroll
            for (int k = 0; k < (kQuads + 63) / 64; k++) {
                const int Q = 64 * k + lane;
                if (Q < kQuads) {
                    const int c = Q / 7, part = Q - 7 * c;
                    const int cnt = (int)L.mfn[c];
                    v4f v;
                    if (part == 0) {
                        v.x = (float)cnt; v.y = 0.f; v.z = 0.f; v.w = 0.f;
                    } else {
                        const int f0 = 4 * part - 4;          // first of the four floats of mfl this quad holds
                        v.x = f0 < 6 * cnt ? L.mfl[f0][c] : 0.f;
                        v.y = f0 + 1 < 6 * cnt ? L.mfl[f0 + 1][c] : 0.f;
                        v.z = f0 + 2 < 6 * cnt ? L.mfl[f0 + 2][c] : 0.f;
                        v.w = f0 + 3 < 6 * cnt ? L.mfl[f0 + 3][c] : 0.f;
                    }
                    float* dst = mf + 4 * Q;
                    if (THROUGH) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
                    else *reinterpret_cast<v4f*>(dst) = v;
                }
            }
        }
    }
}
// obstacle 2: the free box of the environment (state 13, point count, manifold 24: kBoxFloats per env in d_box) travels
// with the state record, like the contact cache
// (64 floats = 256 bytes per environment: two 128-byte lines of its own, written whole by the one store instruction of a
//  hand-off -- MI355X_MICROARCH.md's form for write-through stores another XCD's wave loads; at 40 floats an env's tail shared
//  a line with its neighbour's head)
constexpr int kBoxFloats = 64;
template <class LT>
__device__ __forceinline__ void load_box(LT& L, const float* __restrict__ bx, int lane) {
    if constexpr (!LT::kV2) {
        if (bx) {
            lane = launder_lane(lane);
            if (lane < 13) L.box[lane] = bx[lane];
            else if (lane == 13) L.bmn = (int)bx[13];
            else if (lane < 38) L.bman[lane - 14] = bx[lane];
            lds_sync();
        }
    }
}
template <class LT, bool THROUGH>
__device__ __forceinline__ void store_box(LT& L, float* __restrict__ bx, int lane) {
    if constexpr (!LT::kV2) {
        if (bx) {
            lane = launder_lane(lane);
            lds_sync();
            {
                const float v = lane < 13 ? L.box[lane] : (lane == 13 ? (float)L.bmn : (lane < 38 ? L.bman[lane - 14] : 0.f));
                if (THROUGH) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(bx + lane), "v"(v) : "memory");
                else bx[lane] = v;
            }
        }
    }
}
template <class LT>
__device__ __forceinline__ void write_obs(LT& L, float* __restrict__ obs, int lane) {
    constexpr int N = LT::kN;
    lane = launder_lane(lane);
    // obs = [q, qd, tau_motor | pos3 quat4 | fz]; rec = [pos3 quat4 w3 v3 | q qd taum | fz px]
    for (int i = lane; i < 3 * N + 8; i += 64) {
        float x;
        if (i < 3 * N) x = L.rec[13 + i];
        else if (i < 3 * N + 7) x = L.rec[i - 3 * N];
        else x = L.rec[13 + 3 * N];
        obs[i] = x;
    }
}
template <class LT>
__device__ __forceinline__ void soft_reset(LT& L, int lane) {
    constexpr int N = LT::kN;
    lane = launder_lane(lane);
    // snake.py:96-99,119-127: base pose/twist and joint q, qd; motor-torque and sensor caches persist [U]
    for (int i = lane; i < 13 + 2 * N; i += 64) L.rec[i] = (i == 6) ? 1.0f : 0.0f;
}

// ----------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------
template <int N, bool V2>
__global__ __launch_bounds__(64, 2) void env_step_kernel(const DevModel* __restrict__ Mp, float* __restrict__ recs,
                                                      const float* __restrict__ mu_plane,
                                                      float* __restrict__ actions, float* __restrict__ obs,
                                                      float* __restrict__ rew, uint8_t* __restrict__ done,
                                                      int32_t* __restrict__ substeps, int vec_mode, int n_envs,
                                                      const int32_t* __restrict__ order, float* __restrict__ rows_all,
                                                      float* __restrict__ mf_all, unsigned long long* __restrict__ ovf,
                                                      float* __restrict__ box_all, int obs_stride, int packed) {
    extern __shared__ float4 smem_raw[];
    using LT = Lds<N, V2>;
    LT& L = *reinterpret_cast<LT*>(smem_raw);
    const DevModel& M = *Mp;
    // longest-first schedule: workgroup b takes the envs with the b-th, (b + G)-th, ... largest predicted work (G
    // workgroups: as many as the chip holds at once; the block of streamed constraint rows belongs to the WORKGROUP)
    for (int slot_ = blockIdx.x; slot_ < n_envs; slot_ += gridDim.x) {
    const int env = order ? __builtin_amdgcn_readfirstlane(order[slot_]) : slot_;
    const int lane = threadIdx.x;
    if (M.poison) { L.poison(lane); lds_sync(); }
    load_rec(L, recs + (size_t)env * LT::REC, lane);
    const int A = M.act_dim;
    // checkBound (SnakeGymEnv.py:82-88) clips the caller's array in place
    float act = 0.f;
    if (lane < A) {
        act = actions[(size_t)env * A + lane];
        float cl = fminf(fmaxf(act, -1.0f), 1.0f);
        if (cl != act) actions[(size_t)env * A + lane] = cl;
        act = cl;
    }
    // createAction (snake.py:247-269) + convertActionToJointCommand (snake.py:223-225)
    if (lane < N) L.targets[lane] = 0.f;
    lds_sync();
    if (lane < A) {
        int slot = (M.gait == 0) ? 2 * lane : ((M.gait == 1) ? 2 * lane + 1 : lane);
        L.targets[slot] = act * M.scaling;
    }
    lds_sync();
    float mu = fminf(M.mu_link * mu_plane[env], 10.0f);
    // constraint rows of the streamed-row solve (also behind the register-resident one, for the substeps whose contacts
    // outgrow its slots: substep())
    float* env_rows = rows_all + (size_t)blockIdx.x * Lds<N, false>::kRowFloats;
    float* env_mf = mf_all ? mf_all + (size_t)env * (2 * N * kMfFloats) : nullptr;   // contact cache (contact_model 1)
    load_mf(L, env_mf, lane);
    float* env_box = box_all ? box_all + (size_t)env * kBoxFloats : nullptr;
    load_box(L, env_box, lane);
    fk_vel(L, M, lane);
    // Snake.step servo loop (snake.py:283-304)
    int counter = 0;
    bool end_height = false;
    int it_dummy = 0, nc_dummy = 0;
    SensorHint hint;
    hint.always = false;
    hint.h_prev = mean_height(L, M, lane);
    while (true) {
        float e = (lane < N) ? (L.targets[lane] - L.q()[lane]) : 0.f;
        float nrm = sqrtf(wave_sum<64>(e * e));
        if (!(nrm > M.servo_tol)) break;
        hint.counter_next = counter + 1;
        substep(L, M, lane, mu, it_dummy, nc_dummy, hint, env_rows, env_mf, ovf);
        counter++;
        hint.h_prev = mean_height(L, M, lane);
        if (hint.h_prev > M.height_thr) { end_height = true; break; }
        if (counter > M.max_counter) break;
    }
    // SnakeGymEnv.step (SnakeGymEnv.py:36-42)
    float en = (lane < N) ? L.qd()[lane] * L.taum()[lane] * M.energy_dt : 0.f;   // snake.py:336-341
    float energy = wave_sum<64>(en);
    float x = L.rec[0], y = L.rec[1], fzv = L.fz();
    float r = M.alpha * (x - L.prev_x()) + (fabsf(fzv) > M.coll_force ? M.coll_pen : 0.f) - M.beta * fabsf(y) -
              M.gamma * energy;
    bool dn = fabsf(L.rec[13 + M.term_index]) > M.term_angle;
    if (!dn) dn = mean_height(L, M, lane) > M.height_thr;
    if (!dn) dn = end_height;
    if (dn) r += M.done_pen;
    float* ob = obs + (size_t)env * obs_stride;
    if (!(dn && vec_mode)) write_obs(L, ob, lane);
    lds_sync();
    if (dn) {
        soft_reset(L, lane);
        lds_sync();
        if (vec_mode) write_obs(L, ob, lane);   // worker returns env.reset()'s obs
    }
    lds_sync();
    if (lane == 0) {
        // _observation = terminal obs (SnakeGymEnv.py:42); the worker's reset() refreshes it
        L.prev_x() = (dn && vec_mode) ? 0.0f : x;
        if (packed) {       // snk_step_packed: [obs | reward | done] rows (StepArgs::packed)
            ob[3 * N + 8] = r;
            reinterpret_cast<uint32_t*>(ob)[3 * N + 9] = dn ? 1u : 0u;
        } else {
            rew[env] = r;
            done[env] = dn ? 1 : 0;
        }
        if (substeps) substeps[env] = counter;
    }
    store_rec(L, recs + (size_t)env * LT::REC, lane);
    store_mf<LT, false>(L, env_mf, lane);
    store_box<LT, false>(L, env_box, lane);
    lds_sync();
    }
}

template <int N, bool V2>
__global__ __launch_bounds__(64, 2) void substep_kernel(const DevModel* __restrict__ Mp, float* __restrict__ recs,
                                                     const float* __restrict__ mu_plane,
                                                     const float* __restrict__ targets, int k,
                                                     int32_t* __restrict__ info, int n_envs, float* __restrict__ rows_all,
                                                     float* __restrict__ mf_all, unsigned long long* __restrict__ ovf,
                                                     float* __restrict__ box_all) {
    extern __shared__ float4 smem_raw[];
    using LT = Lds<N, V2>;
    LT& L = *reinterpret_cast<LT*>(smem_raw);
    const DevModel& M = *Mp;
    const int lane = threadIdx.x;
    for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {      // (the block of streamed rows belongs to the workgroup)
    if (M.poison) { L.poison(lane); lds_sync(); }
    load_rec(L, recs + (size_t)env * LT::REC, lane);
    if (lane < N) L.targets[lane] = targets[(size_t)env * N + lane];
    lds_sync();
    float mu = fminf(M.mu_link * mu_plane[env], 10.0f);
    fk_vel(L, M, lane);
    int iters = 0, nc = 0;
    SensorHint hint;
    hint.always = true; hint.counter_next = 0; hint.h_prev = 0.f;
    float* env_rows = rows_all + (size_t)blockIdx.x * Lds<N, false>::kRowFloats;
    float* env_mf = mf_all ? mf_all + (size_t)env * (2 * N * kMfFloats) : nullptr;
    load_mf(L, env_mf, lane);
    float* env_box = box_all ? box_all + (size_t)env * kBoxFloats : nullptr;
    load_box(L, env_box, lane);
    for (int s = 0; s < k; s++) substep(L, M, lane, mu, iters, nc, hint, env_rows, env_mf, ovf);
    if (info && lane == 0) { info[2 * env] = iters; info[2 * env + 1] = nc; }
    store_rec(L, recs + (size_t)env * LT::REC, lane);
    store_mf<LT, false>(L, env_mf, lane);
    store_box<LT, false>(L, env_box, lane);
    lds_sync();
    }
}

template <int N, bool V2>
__global__ __launch_bounds__(64) void reset_kernel(float* __restrict__ recs, const uint8_t* __restrict__ mask,
                                                   float* __restrict__ obs, int hard, int n_envs) {
    extern __shared__ float4 smem_raw[];
    using LT = Lds<N, V2>;
    LT& L = *reinterpret_cast<LT*>(smem_raw);
    const int env = blockIdx.x;
    const int lane = threadIdx.x;
    if (env >= n_envs) return;
    if (mask && !mask[env]) return;
    load_rec(L, recs + (size_t)env * LT::REC, lane);
    soft_reset(L, lane);
    lds_sync();
    if (hard) {
        for (int i = 13 + 2 * N + lane; i < LT::REC; i += 64) L.rec[i] = 0.f;
        lds_sync();
    }
    if (lane == 0) L.prev_x() = 0.0f;   // _observation = reset obs (SnakeGymEnv.py:30), x = 0
    lds_sync();
    if (obs) write_obs(L, obs + (size_t)env * (3 * N + 8), lane);
    store_rec(L, recs + (size_t)env * LT::REC, lane);
}

template <int N, bool V2>
__global__ __launch_bounds__(64) void obs_kernel(const DevModel* __restrict__ Mp, const float* __restrict__ recs,
                                                 float* __restrict__ obs, float* __restrict__ height,
                                                 float* __restrict__ linkpos, int n_envs) {
    extern __shared__ float4 smem_raw[];
    using LT = Lds<N, V2>;
    LT& L = *reinterpret_cast<LT*>(smem_raw);
    const DevModel& M = *Mp;
    const int env = blockIdx.x;
    const int lane = threadIdx.x;
    if (env >= n_envs) return;
    load_rec(L, recs + (size_t)env * LT::REC, lane);
    if (obs) write_obs(L, obs + (size_t)env * (3 * N + 8), lane);
    if (height || linkpos) {
        fk_vel(L, M, lane);
        if (height) {
            float h = mean_height(L, M, lane);
            if (lane == 0) height[env] = h;
        }
        if (linkpos && lane <= N) {
            // getLinkPositions (snake.py:138-146): COM of Bullet links 0,3,...,3n -- the `base` link and
            // the OUTPUT_BODY links (COM at their joint origin) -- as [x.., y.., z..]
            f3 c = ld3(L.o[lane]);
            if (lane == 0) c = c + mulRv(L.R[0], ld3(M.hbase));
            float* out = linkpos + (size_t)env * 3 * (N + 1);
            out[lane] = c.x; out[(N + 1) + lane] = c.y; out[2 * (N + 1) + lane] = c.z;
        }
    }
}

// ----------------------------------------------------------------------------------
// Launch planning.  An env-step costs 0..41 substeps depending on how far the joints are
// from their targets (snake.py:228-235), and a substep is a latency-bound ~0.5 ms chain, so
// the launch time is set by envs with many substeps that start late.  This one-block kernel
// sorts the envs by their initial servo error, largest first (counting sort on a 256-bin
// key); env_step_kernel's workgroup b then runs order[b].  Pure scheduling: results do not
// depend on the order.
// ----------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(1024) void plan_kernel(const DevModel* __restrict__ Mp, const float* __restrict__ recs,
                                                    const float* __restrict__ actions, int32_t* __restrict__ order,
                                                    int n_envs) {
    constexpr int REC = (N <= 16) ? 64 : 128;
    constexpr int NBIN = 256;
    __shared__ int hist[NBIN];
    __shared__ int base[NBIN];
    const DevModel& M = *Mp;
    const int tid = threadIdx.x;
    for (int i = tid; i < NBIN; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int A = M.act_dim;
    auto key_of = [&](int e) {
        const float* q = recs + (size_t)e * REC + 13;
        float err2 = 0.f;
        for (int j = 0; j < N; j++) {
            int k = (M.gait == 0) ? ((j & 1) ? -1 : j / 2) : ((M.gait == 1) ? ((j & 1) ? j / 2 : -1) : j);
            float t = 0.f;
            if (k >= 0 && k < A) t = fminf(fmaxf(actions[(size_t)e * A + k], -1.f), 1.f) * M.scaling;
            float d = t - q[j];
            err2 += d * d;
        }
        // larger error -> smaller bin index -> earlier workgroup.  log scale, 256 bins.
        float l = __log2f(fmaxf(err2, 1e-12f));          // about [-40, 8]
        int b = (int)((8.0f - l) * 5.0f);
        return b < 0 ? 0 : (b > NBIN - 1 ? NBIN - 1 : b);
    };
    for (int e = tid; e < n_envs; e += blockDim.x) atomicAdd(&hist[key_of(e)], 1);
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < NBIN; i++) { base[i] = run; run += hist[i]; }
    }
    __syncthreads();
    for (int e = tid; e < n_envs; e += blockDim.x) {
        int pos = atomicAdd(&base[key_of(e)], 1);
        order[pos] = e;
    }
}

// ----------------------------------------------------------------------------------
// In-launch scheduling of env-steps (env_step_sched_kernel).
//
// An env-step is 0..41 sequential substeps (snake.py:283-304); a launch of E env-steps on G
// resident waves ends when the slowest wave does.  With whole env-steps as the unit, that is
// (longest + shortest) substeps when E = 2 G -- 45 for the bench workload whose mean load is 35.5
// per wave (tools/balance_dump.py) -- because the jobs are too coarse to level.  An env-step can be
// cut at any substep boundary, though: its whole state is the env's record (plus the substep
// counter).  So G persistent waves pull env-steps from a queue, run them for `quantum` substeps
// and put them back at the end of the queue -- unless no waiting env has more work left than
// this one, in which case the wave just carries on ("longest remaining time first", which
// levels the finish times to within about a quantum).  The remaining work is known almost
// exactly: the position motors shrink the servo error by (1 - kp) per substep [U], so
// remaining = log(err / tol) / -log(1 - kp), capped by the substep counter's limit.
//
// Queue: one ring of {ticket, remaining, env} entries.  A pop is ONE returning atomic add on
// `head` (a ticket), then a wait for that ticket's entry; a push is one atomic add on `tail` and
// one 8-byte agent-scope store (the slot is a tagged granule, read back with a returning atomic: sched_pop).  (A compare-and-swap pop costs O(G^2) attempts when G waves reach
// a slice boundary together: 14 ms per launch, measured.)  Tickets are never reset; unsigned
// wrap-around is harmless because the ring size is a power of two (tests preset head/tail just below 2^32).  Pops in excess of pushes wait for an entry that may never come; they
// leave when `finished` says every env-step is complete, and the next launch starts its tickets at
// `head`.  waiting[r] counts queued env-steps with r substeps left (the carry-on test).
// The record hand-off between waves follows MI355X_MICROARCH.md "inter-workgroup visibility":
// every handed-off byte stored write-through (sc1), s_waitcnt vmcnt(0), then the queue entry;
// consumer: entry seen, agent acquire, wait, plain loads.  Results do not depend on the schedule: a slice boundary
// stores and reloads exactly the floats a continuing wave keeps (the property test-mode telemetry
// relies on, tests/test_gpu_env.py).
//
// Every wait is bounded in wall-clock time (kWaitTicks): a wave that gives up raises the
// host-visible word `alarm` and leaves; the others follow, so the grid always drains.
// ----------------------------------------------------------------------------------
constexpr int kBuckets = 64;
constexpr long long kWaitTicks = 200000000;    // wall_clock64() runs at 100 MHz: 2 s (an env-step is < 50 ms)

// The model of the scheduled kernel lives in constant memory: its queue atomics and fences make the
// compiler treat every load through a global pointer as clobbered (vector loads where the plain
// kernel has scalar ones); loads from __constant__ stay scalar.  One slot per live handle
// (snk_api.hip hands them out).
constexpr int kModelSlots = 32;
__constant__ DevModel g_models[kModelSlots];

struct Sched {
    uint32_t* head;             // tickets claimed
    uint32_t* tail;             // tickets issued
    unsigned long long* ent;    // [cap]: (ticket << 32) | (remaining << 24) | env; all-ones when never written
    int32_t* waiting;           // [kBuckets] queued env-steps by substeps left
    int32_t* counter;           // [n_envs] substeps done so far in this env-step
    int32_t* finished;          // env-steps completed in this launch
    int32_t* alarm;             // host-mapped: set when a bounded wait ran out
    uint32_t cap;               // ring size: a power of two >= 2 n_envs (an env is queued at most once), so that the
                                // slot of a ticket, tk & (cap - 1), stays consistent when the 32-bit tickets wrap
    int32_t quantum;            // substeps per slice
    int32_t hyst;               // a slice's env-step is handed off when a waiting one has at least this many more substeps
                                // left.  1 = strict longest-remaining-first: two env-steps of equal length then swap places
                                // after every substep (each hand-off moves the record and the contact cache through
                                // memory); 3 levels the finish times as well and hands off a third as often: measured
                                // 336.6 k -> 342.7 k env-steps/s (1, 3, 4, 6, 8: 336.6 / 342.7 / 341.6 / 341.4 / 322.9)
    long long* wstat;           // SNK_SCHED_DEBUG: [grid][4] ticks waiting, ticks alive, slices, substeps
};

__device__ __forceinline__ int predict_remaining(const DevModel& M, float err, int counter) {
    if (!(err > M.servo_tol)) return 0;
    const float decay = fmaxf(-__log2f(fminf(fmaxf(1.0f - M.kp, 1e-6f), 0.999f)), 1e-3f);
    const float r = ceilf(__log2f(err / M.servo_tol) / decay);
    const int cap = M.max_counter + 1 - counter;
    int R = (int)fminf(r, (float)cap);
    R = R < 1 ? 1 : R;
    return R > kBuckets - 1 ? kBuckets - 1 : R;
}

// a wave-uniform condition as a scalar the compiler knows to be uniform (keeps the scheduler's loops out of
// exec-mask control flow)
__device__ __forceinline__ bool uni(bool c) { return __builtin_amdgcn_readfirstlane(c ? 1 : 0) != 0; }

// most substeps left among the queued env-steps (-1: queue empty)
__device__ __forceinline__ int sched_top(const Sched& sc, int lane) {
    const int w = __hip_atomic_load(&sc.waiting[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long m = __ballot(w > 0);
    return m ? 63 - __clzll(m) : -1;
}

__device__ __forceinline__ void sched_alarm(const Sched& sc, int lane) {
    (void)lane;
    __hip_atomic_store(sc.alarm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // every lane, same word
}

// next env for this wave, or -1 when every env-step of the launch is complete (or on alarm).
// Two waits.  (1) `tail` -- a word only agent-scope atomic adds touch -- is polled with a 4-byte sc1 load until
// ticket tk has been issued (MI355X_MICROARCH.md, hand-off table, row 3: "agent-scope atomic adds ... a
// global_load_dword sc1 poll of that counter").  (2) A producer takes its ticket BEFORE it stores the entry
// (sched_push), so the slot may still hold the entry of ticket tk - cap: the slot is a tagged 8-byte granule
// {ticket, remaining|env} and is re-read until the tag matches.  That re-read is a RETURNING ATOMIC (an add of a
// zero the compiler cannot see through): it executes where agent-scope atomics execute, beyond the per-XCD L2s,
// so no cached copy of the slot -- in this CU's L1 or in this XCD's L2 -- can answer it.  (Round 1 read the slot
// with `__hip_atomic_load`, i.e. `global_load_dwordx2 sc1`, which is served by the XCD's own L2; the guide lists
// that as observed-fresh for granules, not as guaranteed, and its row 3 excludes dwordx2 loads outright.  The one
// hang on record, gpurun_out/d3000.log, predates the first committed scheduler and had the lane-threaded back edge
// described below as its cause; the atomic read removes the remaining reliance on an observed behaviour.)
__device__ __forceinline__ int sched_pop(const Sched& sc, int lane, int n_envs) {
    // NO `if (lane == 0)` around the queue operations of this file: with a lane-dependent branch at the top of the
    // scheduling loop the compiler threads the loop's back edge per lane, lane 0 and lanes 1..63 then run the loop
    // body in separate passes, and every cross-lane operation of the solver breaks (observed: a wave that
    // re-processes one env for ever).  Every lane issues the atomic with its own operand instead (the atomic
    // optimizer folds the 64 into one memory operation).
    uint32_t tk = atomicAdd(sc.head, lane == 0 ? 1u : 0u);
    tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    const long long t_start = wall_clock64();
    int nap = 1;
    for (;;) {
        const uint32_t t = __hip_atomic_load(sc.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)t) - tk) > 0) break;    // ticket tk has been issued
        const int fin = __hip_atomic_load(sc.finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(fin) >= n_envs) return -1;
        if (uni(wall_clock64() - t_start > kWaitTicks)) {
            // give up once nobody can still be working, or when somebody else already has
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(sc.alarm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) ||
                uni(wall_clock64() - t_start > 4 * kWaitTicks)) {
                sched_alarm(sc, lane);
                return -1;
            }
        }
        for (int i = 0; i < nap; i++) __builtin_amdgcn_s_sleep(16);      // ~0.5 us, backing off to ~7 us
        if (nap < 16) nap++;
    }
    unsigned long long* e = sc.ent + (tk & (sc.cap - 1u));
    unsigned long long zero = 0ull;
    asm volatile("" : "+v"(zero));      // opaque: an add of a literal 0 would be folded into a plain atomic load
    for (;;) {
        // every lane adds 0 to the same slot (the atomic optimizer folds the 64 into one memory operation)
        const unsigned long long v = __hip_atomic_fetch_add(e, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        if (hi == tk) {
            atomicAdd(&sc.waiting[lo >> 24], lane == 0 ? -1 : 0);
            return (int)(lo & 0xFFFFFFu);
        }
        if (uni(wall_clock64() - t_start > 4 * kWaitTicks)) break;
        __builtin_amdgcn_s_sleep(4);
    }
    sched_alarm(sc, lane);
    return -1;
}

// hand an unfinished env-step (record and counter already stored write-through by this wave) to whoever pops it
__device__ __forceinline__ void sched_push(const Sched& sc, int lane, int env, int remaining) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the write-through stores have left before the entry does
    atomicAdd(&sc.waiting[remaining], lane == 0 ? 1 : 0);
    const uint32_t tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)atomicAdd(sc.tail, lane == 0 ? 1u : 0u));
    __hip_atomic_store(sc.ent + (tk & (sc.cap - 1u)),        // every lane stores the same 8 bytes
                       ((unsigned long long)tk << 32) | ((unsigned long long)remaining << 24) |
                           (unsigned long long)(uint32_t)env,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One block: the queue of a launch, env-steps with the most predicted substeps first (counting sort).
template <int N>
__global__ __launch_bounds__(1024) void plan_sched_kernel(const DevModel* __restrict__ Mp, const float* __restrict__ recs,
                                                          const float* __restrict__ actions, Sched sc, int n_envs) {
    constexpr int REC = (N <= 16) ? 64 : 128;
    __shared__ uint32_t hist[kBuckets], base[kBuckets];
    const DevModel& M = *Mp;
    const int tid = threadIdx.x;
    if (tid < kBuckets) hist[tid] = 0;
    __syncthreads();
    const int A = M.act_dim;
    auto key_of = [&](int e) {
        const float* q = recs + (size_t)e * REC + 13;
        float err2 = 0.f;
        for (int j = 0; j < N; j++) {
            int k = (M.gait == 0) ? ((j & 1) ? -1 : j / 2) : ((M.gait == 1) ? ((j & 1) ? j / 2 : -1) : j);
            float t = 0.f;
            if (k >= 0 && k < A) t = fminf(fmaxf(actions[(size_t)e * A + k], -1.f), 1.f) * M.scaling;
            float d = t - q[j];
            err2 += d * d;
        }
        return predict_remaining(M, sqrtf(err2), 0);
    };
    constexpr int kKeep = 8;                 // keys of the first 8 envs of a thread stay in registers for the second pass
    int keys[kKeep];
#pragma unroll
    for (int i = 0; i < kKeep; i++) {
        const int e = tid + i * 1024;
        keys[i] = e < n_envs ? key_of(e) : 0;
        if (e < n_envs) atomicAdd(&hist[keys[i]], 1u);
    }
    for (int e = tid + kKeep * 1024; e < n_envs; e += 1024) atomicAdd(&hist[key_of(e)], 1u);
    __syncthreads();
    const uint32_t t0 = *sc.head;       // tickets the previous launch's leaving waves took are skipped
    if (tid == 0) {
        uint32_t run = t0;
        for (int b = kBuckets - 1; b >= 0; b--) { base[b] = run; run += hist[b]; }
        *sc.tail = run;
        *sc.finished = 0;
    }
    if (tid < kBuckets) sc.waiting[tid] = (int32_t)hist[tid];
    __syncthreads();
    auto enqueue = [&](int e, int b) {
        const uint32_t tk = atomicAdd(&base[b], 1u);
        sc.ent[tk & (sc.cap - 1u)] = ((unsigned long long)tk << 32) | ((unsigned long long)b << 24) | (unsigned long long)(uint32_t)e;
        sc.counter[e] = 0;
    };
#pragma unroll
    for (int i = 0; i < kKeep; i++) {
        const int e = tid + i * 1024;
        if (e < n_envs) enqueue(e, keys[i]);
    }
    for (int e = tid + kKeep * 1024; e < n_envs; e += 1024) enqueue(e, key_of(e));
}

// Arguments of the scheduled step kernel: ONE struct, passed by value -- i.e. it IS the kernel-argument segment -- and
// read through step_args(), a pointer to that segment the compiler cannot see through, at the place of use.  Passed
// as ordinary kernel arguments these twenty pointers are loaded once at kernel entry and then sit in scalar registers
// for the whole launch; the register-resident solve has none to spare, so ~190 of them were spilled into lanes of FOUR
// vector registers, which the solve's row registers then paid for with reloads from scratch memory inside the
// Gauss-Seidel loop (round-3 ISA).  Read where needed, a pointer lives for a few instructions.
struct StepArgs {
    float* recs;
    const float* mu_plane;
    float* actions;
    float* obs;
    float* rew;
    uint8_t* done;
    int32_t* substeps;
    float* rows_all;
    float* mf_all;
    unsigned long long* ovf;
    float* box_all;
    Sched sc;
    int32_t model_slot, vec_mode, n_envs;
    // obs row stride in floats (3n + 8 for the dense [n_envs x obs_dim] output).  packed != 0 (snk_step_packed): reward
    // and done flag of env e go into its obs row, at float index obs_dim (f32) and obs_dim + 1 (u32 0 / 1), instead of
    // rew[] / done[]: one [n_envs x stride] buffer that a sharded vector env gathers as it is (device_env.py)
    int32_t obs_stride, packed, pad_;
};
typedef const StepArgs __attribute__((address_space(4))) * StepArgPtr;
__device__ __forceinline__ StepArgPtr step_args() {
    StepArgPtr p = (StepArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ Sched load_sched(StepArgPtr p) {
    Sched sc;
    sc.head = p->sc.head; sc.tail = p->sc.tail; sc.ent = p->sc.ent; sc.waiting = p->sc.waiting;
    sc.counter = p->sc.counter; sc.finished = p->sc.finished; sc.alarm = p->sc.alarm;
    sc.cap = p->sc.cap; sc.quantum = p->sc.quantum; sc.hyst = p->sc.hyst; sc.wstat = p->sc.wstat;
    return sc;
}

template <int N, bool V2>
__global__ __launch_bounds__(64, SNK_LB) void env_step_sched_kernel(StepArgs args_by_value) {
    (void)args_by_value;            // read through step_args() only
    extern __shared__ float4 smem_raw[];
    using LT = Lds<N, V2>;
    LT& L = *reinterpret_cast<LT*>(smem_raw);
    int lane = threadIdx.x;
#ifdef SNK_SCHED_DEBUG
    long long t_wait = 0, n_slices = 0, n_sub = 0, n_chk = 0, n_req = 0, s_top = 0, s_rem = 0;
    const long long t_birth = wall_clock64();
#endif
    for (;;) {
#ifdef SNK_SCHED_DEBUG
        const long long t_p0 = wall_clock64();
#endif
        const StepArgPtr ap = step_args();
        const DevModel& M = g_models[ap->model_slot];
        const int A = M.act_dim;
        const int n_envs = ap->n_envs;
        int env;
        {
            const Sched sc = load_sched(ap);
            if (__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(1))) != 64) {   // the 64 lanes stay together (see sched_pop)
                sched_alarm(sc, lane);
                break;
            }
            env = __builtin_amdgcn_readfirstlane(sched_pop(sc, lane, n_envs));
#ifdef SNK_SCHED_DEBUG
            if (env >= 0) { t_wait += wall_clock64() - t_p0; n_slices++; }
            else if (lane == 0) {
                long long* w = sc.wstat + 8 * (size_t)blockIdx.x;
                w[0] = t_wait; w[1] = t_p0 - t_birth; w[2] = n_slices; w[3] = n_sub;
                w[4] = n_chk; w[5] = n_req; w[6] = s_top; w[7] = s_rem;
            }
#endif
        }
        if (env < 0) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (M.poison) { L.poison(lane); lds_sync(); }
        load_rec(L, ap->recs + (size_t)env * LT::REC, lane);
        int counter = __builtin_amdgcn_readfirstlane(ap->sc.counter[env]);
        if (counter < 0 || counter > M.max_counter + 1 || env >= n_envs) {     // never, unless a hand-off went wrong
            sched_alarm(load_sched(ap), lane);
            break;
        }
        // checkBound (SnakeGymEnv.py:82-88) clips the caller's array in place
        float act = 0.f;
        if (lane < A) {
            float* actions = ap->actions;
            act = actions[(size_t)env * A + lane];
            float cl = fminf(fmaxf(act, -1.0f), 1.0f);
            if (cl != act) actions[(size_t)env * A + lane] = cl;
            act = cl;
        }
        // createAction (snake.py:247-269) + convertActionToJointCommand (snake.py:223-225)
        if (lane < N) L.targets[lane] = 0.f;
        lds_sync();
        if (lane < A) {
            int slot = (M.gait == 0) ? 2 * lane : ((M.gait == 1) ? 2 * lane + 1 : lane);
            L.targets[slot] = act * M.scaling;
        }
        lds_sync();
        const float mu = fminf(M.mu_link * ap->mu_plane[env], 10.0f);
        // contact cache of the environment (contact_model 1): recomputed from the argument segment wherever it is used
        auto env_mf = [&](StepArgPtr q) -> float* {
            float* mf_all = q->mf_all;
            return mf_all ? mf_all + (size_t)env * (2 * N * kMfFloats) : nullptr;
        };
        load_mf(L, env_mf(ap), lane);
        auto env_box = [&](StepArgPtr q) -> float* {           // the free box of obstacle 2 (null otherwise)
            float* box_all = q->box_all;
            return box_all ? box_all + (size_t)env * kBoxFloats : nullptr;
        };
        load_box(L, env_box(ap), lane);
        fk_vel(L, M, lane);
        // Snake.step servo loop (snake.py:283-304), `quantum` substeps at a time
        bool end_height = false, complete = false;
        int it_dummy = 0, nc_dummy = 0, in_slice = 0;
        SensorHint hint;
        hint.always = false;
        hint.h_prev = mean_height(L, M, lane);
        while (true) {
            float e = (lane < N) ? (L.targets[lane] - L.q()[lane]) : 0.f;
            float nrm = sqrtf(wave_sum<64>(e * e));
            if (uni(!(nrm > M.servo_tol))) { complete = true; break; }
            const StepArgPtr aq = step_args();
            if (in_slice >= aq->sc.quantum) {
                // slice boundary: carry on unless an env with more work left is waiting
                const Sched sc = load_sched(aq);
                const int remaining = predict_remaining(M, __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nrm))), counter);
                const int top = sched_top(sc, lane);
#ifdef SNK_SCHED_DEBUG
                n_chk++; s_top += top; s_rem += remaining; if (top > remaining) n_req++;
#endif
                if (top >= remaining + sc.hyst) {
                    store_rec_through(L, aq->recs + (size_t)env * LT::REC, lane);
                    store_mf<LT, true>(L, env_mf(aq), lane);        // the contact cache travels with the record
                    store_box<LT, true>(L, env_box(aq), lane);
                    __hip_atomic_store(&sc.counter[env], counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sched_push(sc, lane, env, remaining);
                    break;
                }
                in_slice = 0;
            }
            hint.counter_next = counter + 1;
            {
                // constraint rows of the streamed-row solve, one block per resident wave (also behind the register-resident
                // solve, for the substeps whose contacts outgrow its slots: substep())
                float* env_rows = aq->rows_all + (size_t)blockIdx.x * Lds<N, false>::kRowFloats;
                substep(L, M, lane, mu, it_dummy, nc_dummy, hint, env_rows, env_mf(aq), aq->ovf);
                lane = lane_id();       // (not kept in a register across the solve)
            }
#ifdef SNK_SCHED_DEBUG
            n_sub++;
#endif
            counter++;
            in_slice++;
            hint.h_prev = mean_height(L, M, lane);
            if (uni(hint.h_prev > M.height_thr)) { end_height = true; complete = true; break; }
            if (counter > M.max_counter) { complete = true; break; }
        }
        if (!complete) continue;
        const StepArgPtr af = step_args();
        // SnakeGymEnv.step (SnakeGymEnv.py:36-42)
        float en = (lane < N) ? L.qd()[lane] * L.taum()[lane] * M.energy_dt : 0.f;   // snake.py:336-341
        float energy = wave_sum<64>(en);
        float x = L.rec[0], y = L.rec[1], fzv = L.fz();
        float r = M.alpha * (x - L.prev_x()) + (fabsf(fzv) > M.coll_force ? M.coll_pen : 0.f) - M.beta * fabsf(y) -
                  M.gamma * energy;
        bool dn = uni(fabsf(L.rec[13 + M.term_index]) > M.term_angle);
        if (!dn) dn = uni(mean_height(L, M, lane) > M.height_thr);
        if (!dn) dn = end_height;
        if (dn) r += M.done_pen;
        const int vec_mode = af->vec_mode;
        float* ob = af->obs + (size_t)env * af->obs_stride;
        if (!(dn && vec_mode)) write_obs(L, ob, lane);
        lds_sync();
        if (dn) {
            soft_reset(L, lane);
            lds_sync();
            if (vec_mode) write_obs(L, ob, lane);   // worker returns env.reset()'s obs
        }
        lds_sync();
        if (lane == 0) {
            // _observation = terminal obs (SnakeGymEnv.py:42); the worker's reset() refreshes it
            L.prev_x() = (dn && vec_mode) ? 0.0f : x;
            if (af->packed) {
                ob[3 * N + 8] = r;
                reinterpret_cast<uint32_t*>(ob)[3 * N + 9] = dn ? 1u : 0u;
            } else {
                af->rew[env] = r;
                af->done[env] = dn ? 1 : 0;
            }
            int32_t* substeps = af->substeps;
            if (substeps) substeps[env] = counter;
        }
        store_rec(L, af->recs + (size_t)env * LT::REC, lane);
        store_mf<LT, false>(L, env_mf(af), lane);
        store_box<LT, false>(L, env_box(af), lane);
        atomicAdd(af->sc.finished, lane == 0 ? 1 : 0);
    }
}

// self test of wave_sum / lane_bcast: out[0] = sum32, out[1] = sum64, out[2] = bcast
__global__ __launch_bounds__(64) void selftest_kernel(float* out) {
    const int lane = threadIdx.x;
    float x = (float)(lane + 1);
    float s32 = wave_sum<32>(lane < 22 ? x : 0.f);
    float s64 = wave_sum<64>(x);
    float s64b = wave_sum<64>(lane < 38 ? x * 0.5f : 0.f);
    float b = lane_bcast(x, 17);
    if (lane == 5) { out[0] = s32; out[1] = s64; out[2] = b; out[3] = s64b; }
    // v2 primitives: per-half sums at lanes 31 / 63, and the half swap
    float hr = half_reduce(x);
    if (lane == 31) out[4] = hr;
    if (lane == 63) out[5] = hr;
    swap2 sw = half_swap(x, x * 100.f);
    out[8 + lane] = sw.a;        // expect [x.lo, (100x).lo]
    out[8 + 64 + lane] = sw.b;   // expect [x.hi, (100x).hi]
}

}  // namespace snk
