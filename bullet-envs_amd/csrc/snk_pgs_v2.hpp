// snk_pgs_v2.hpp -- register-resident constraint solve for the 16-link chain (ND = 22).
//
// The projected Gauss-Seidel sweep is a strictly sequential chain over ~208 rows x 50
// iterations per physics substep; in v1 every row step paid several LDS round trips
// (~500 cycles).  Here ALL rows stay in VGPRs for the whole solve:
//
//   * one 64-lane register holds TWO rows, one per 32-lane half; lane = (half, d), d = lane&31
//       d <  22 : component d of the row vector
//       d == 22 : -rhs impulse               (RJ)        delta-v register holds +1 there
//       d == 31 : accumulated impulse a      (RJ)        delta-v register holds -1 there
//       d == 24 : row denominator J M^-1 J^T (RM; delta-v collects dI*den there: the residual)
//       lane 62 : coupling scalar of the slot's two rows (RJ of a duo, d = 30 of the upper half; delta-v is 0 there)
//       d = 30, 29 of a friction slot: the 2 x 2 coupling block with the previous contact's pair (cone2_step)
//     RJ = J / den (pre-scaled), RM = M^-1 J^T, both halves of `dv` carry the same delta-v.
//     Then   sum_d RJ[d] dv[d] = (J.dv)/den - rhs - a   and the new accumulated impulse of
//     the row is simply clamp(-sum): no per-row scalar is fetched from memory.
//   * 96 slots: 0..31 contact normals (contacts 2s, 2s+1 in the lower / upper half),
//     32..95 friction pairs (contact s: direction A lower, B upper), built in units of mu
//   * a half's dot: 4 DPP row steps + row_bcast:15 -> lane 31 / 63 -> v_readlane; one
//     reduction therefore yields the dots of BOTH rows of a slot
//   * friction pair (cone_step): Bullet resolves both directions from the same velocity, so
//     the two dots are used as they are.  Two consecutive normals (duo_step): the second row
//     must see the first row's update; its dot is corrected exactly by  c * dI_first  with the
//     coupling scalar c = RJ_second . RM_first, precomputed once per substep.  This halves the
//     number of sequential steps of the Gauss-Seidel chain at equal work.
//   * motor rows (motor_step) need no reduction at all: their Jacobian is a unit vector
//   * contributions to delta-v cross halves with v_permlane32_swap, so both halves always
//     hold the complete delta-v
//   * what the hardware charges for (tools/ubench_lat.hip, ubench_step.hip): a lone wave issues
//     one VALU per ~5 clocks whatever the dependencies, a dependent DPP add costs 12.6, a
//     v_readlane -> use round trip ~20, every s_nop is an issue slot.  The steps are therefore
//     hand-written with the fewest instructions on the shortest chain: 26 VALU per duo, 28
//     per friction pair, 4 per motor row, and only the wait states gfx950 requires.
//     (Measured and rejected: software-pipelined look-ahead reductions, a lane-local variant
//     with exec-masked scatter, skipping inert friction pairs -- DESIGN.md section 4.)
//   * Bullet's early-exit residual (max over rows of |dI * den|) is folded into every step
//     with one v_max3 and tested once per iteration
//
// Row construction (M^-1 J^T by ABA delta sweeps, one row per lane) goes through a 64-row
// LDS staging buffer: motors, friction (two batches of 32 contacts x {A, B}), normals; J
// itself is evaluated directly in the (half, d) layout from the contact point and the
// per-lane joint axis/origin.
//
// Restates the same Bullet steps as v1's build_rows_v1/pgs_v1 (snk_device.hpp); the order
// of row updates is identical, so both versions track the oracle.
#pragma once

namespace snk {

// -DSNK_PROFILE: s_memtime stamps between the phases of substep_v2; the phase durations (ticks)
// overwrite the motor-torque outputs of the substep (tools/profile_phases.py reads them).
// (SNK_STAMP: snk_device.hpp)

// slot map (two rows per slot) and the layout of the accumulated impulses handed back in L.app
constexpr int kSlotNormal = 0;    // 32 slots: contacts 2s (lower half), 2s+1 (upper half)
constexpr int kSlotFric = 32;     // 64 slots: contact s, direction A lower / B upper
constexpr int kSlots = 96;
constexpr int kAppMotor = 0;      // app[j]                motor j
constexpr int kAppNormal = 16;    // app[16 + ci]          normal of contact ci
constexpr int kAppFric = 80;      // app[80 + 2 ci + {0,1}] friction A / B of contact ci
constexpr unsigned long long kLowMask = 0x00000000FFFFFFFFull;

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
// ------------------------------------------------------------------------------------
// contacts of the current pose, written at their COMPACT index.
//
// Two contact models, selected per handle (DevModel::contact_model):
//   0  stateless: lane = slot (cylinder slot/2, end cap slot&1); each end cap contributes the lowest point of its
//      rim -- of the implicit cylinder, or of the hull_sides-gon PyBullet imports a URDF <cylinder> as [U] --
//      when closer than the breaking threshold;
//   1  Bullet's persistent manifold [U] (btConvexPlaneCollisionAlgorithm + btPersistentManifold), lane = cylinder:
//      one new support point per step merged into a cache of <= 4 points that lives in global memory
//      (manifold_update below); every cached point gets rows.
// Both leave the same description behind: contact ci = 0..nc-1 in (cylinder, point) order with L.ccP / ccdist /
// ccbody / ccds -> cdir, and per cylinder the range [L.cylbase[c], + L.cyln[c]) of its contacts (the sensor pass sums a
// body's contact forces over the ranges of its cylinders).  At most 4 N contacts get rows (the register-resident
// solve has that many slots): further manifold points are left out, in manifold order.
// ------------------------------------------------------------------------------------
// lowest rim point (x, y in the cylinder's frame) towards dl = world "down" in that frame
__device__ __forceinline__ void rim_point(const DevModel& M, f3 dl, float& lx, float& ly) {
    lx = 0.f; ly = 0.f;
    if (M.hull_sides > 0) {
        float best = -3.0e38f;
        for (int s = 0; s < M.hull_sides; s++) {          // first maximum, the importer's vertex order
            const float vx = M.hull_xy[s][0], vy = M.hull_xy[s][1];
            const float val = dl.x * vx + dl.y * vy;
            if (val > best) { best = val; lx = vx; ly = vy; }
        }
    } else {
        const float rr = sqrtf(dl.x * dl.x + dl.y * dl.y);
        if (rr > 1e-12f) { lx = M.cyl_r * dl.x / rr; ly = M.cyl_r * dl.y / rr; }
    }
}
// world rotation of cylinder c's frame
__device__ __forceinline__ void cyl_world_rot(const float* Rb, const float* Rc, float* Rw) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            Rw[3 * i + j] = Rb[3 * i] * Rc[j] + Rb[3 * i + 1] * Rc[3 + j] + Rb[3 * i + 2] * Rc[6 + j];
}
// friction directions (0,-1,0), (1,0,0) scaled anisotropically in the link's axes: d' = Rw diag(aniso) Rw^T d
__device__ __forceinline__ void friction_dirs(const DevModel& M, const float* Rw, f3& dA, f3& dB) {
    const f3 a = mk3(M.aniso[0], M.aniso[1], M.aniso[2]);
    const f3 l1 = mulRtv(Rw, mk3(0.f, -1.f, 0.f));
    const f3 l2 = mulRtv(Rw, mk3(1.f, 0.f, 0.f));
    dA = mulRv(Rw, mk3(l1.x * a.x, l1.y * a.y, l1.z * a.z));
    // friction_directions 1: the second tangent is a zero vector -- its row comes out J = 0, M^-1 J^T = 0, den = 0 and
    // resolves to nothing in either friction form (the cone over (x, 0) is the box bound of the one row)
    dB = mulRv(Rw, mk3(l2.x * a.x, l2.y * a.y, l2.z * a.z)) * M.fricB;
}

// One cached manifold point: the point on the link in the LINK's coordinates (cylinder frame + cyl_zoff along z),
// the point on the ground in world coordinates (z = 0: the plane's), the refreshed distance, and the normal impulse
// the point carried in the last substep (btManifoldPoint::m_appliedImpulse [U]; warm starting reads it).
struct MPt {
    f3 a, b;
    float d, lam;
};
typedef float mf_v4 __attribute__((ext_vector_type(4)));
// dst = c ? src : dst, field by field.  (A conditional struct assignment inside an unrolled `if (j == where)` chain gets
// its stores merged into ONE store at a computed address, which puts the whole point cache into scratch memory: ~100
// scratch round trips per substep, measured in the ISA of round 3's first build.)
__device__ __forceinline__ void mpt_sel(MPt& dst, const MPt& src, bool c) {
    dst.a.x = c ? src.a.x : dst.a.x; dst.a.y = c ? src.a.y : dst.a.y; dst.a.z = c ? src.a.z : dst.a.z;
    dst.b.x = c ? src.b.x : dst.b.x; dst.b.y = c ? src.b.y : dst.b.y; dst.b.z = c ? src.b.z : dst.b.z;
    dst.d = c ? src.d : dst.d; dst.lam = c ? src.lam : dst.lam;
}
__device__ __forceinline__ void f3_sel(f3& dst, f3 src, bool c) {
    dst.x = c ? src.x : dst.x; dst.y = c ? src.y : dst.y; dst.z = c ? src.z : dst.z;
}

// btPersistentManifold::sortCachedPoints with gContactCalcArea3Points [U]: which cached point the new one replaces
__device__ __forceinline__ int manifold_sort_cached(const MPt (&p)[4], const MPt& np) {
    int mpi = -1;
    float mp = np.d;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (p[i].d < mp) { mpi = i; mp = p[i].d; }
    auto area = [](f3 a1, f3 a0, f3 b1, f3 b0) { const f3 c = cross(a1 - a0, b1 - b0); return dot(c, c); };
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    if (mpi != 0) res[0] = area(np.a, p[1].a, p[3].a, p[2].a);
    if (mpi != 1) res[1] = area(np.a, p[0].a, p[3].a, p[2].a);
    if (mpi != 2) res[2] = area(np.a, p[0].a, p[3].a, p[1].a);
    if (mpi != 3) res[3] = area(np.a, p[0].a, p[2].a, p[1].a);
    int best = 0;
    float bv = fabsf(res[0]);
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (fabsf(res[i]) > bv) { bv = fabsf(res[i]); best = i; }
    return best;
}

// The manifold of this lane's cylinder (n cached points p[0 .. n-1], registers), updated for the current pose -- see
// the oracle's find_contacts_manifold for the Bullet calls restated: a link's collider is a btCompoundShape, so
// btCompoundCollisionAlgorithm first REFRESHES the child's manifold from the new pose (positions, distances, and the
// removal of points that lifted off or drifted: refreshContactPoints), then the child's convex-plane algorithm adds
// this step's support point (getCacheEntry / replaceContactPoint / addManifoldPoint with sortCachedPoints, which sees
// the refreshed distances) [U].  Returns the new number of cached points; their world positions on the link and
// distances in wa / p[].d.  Everything is a select (mpt_sel): the cache stays in registers.
__device__ __forceinline__ int manifold_core(const DevModel& M, int n, MPt (&p)[4], const float* Rw, f3 centre, f3 dl,
                                             f3 (&wa)[4]) {
    n = n < 0 ? 0 : (n > 4 ? 4 : n);
    const float thr = M.break_thr;
    const f3 zoff = mk3(0.f, 0.f, M.cyl_zoff);
    // refresh from the current pose, then drop what lifted off or drifted (last to first, the last one moves in)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        wa[j] = centre + mulRv(Rw, p[j].a - zoff);
        p[j].d = wa[j].z - p[j].b.z;
    }
#pragma unroll
    for (int j = 3; j >= 0; j--) {
        bool drop = !(p[j].d <= thr);
        {
            const float dx = p[j].b.x - wa[j].x, dy = p[j].b.y - wa[j].y, dz = p[j].b.z - (wa[j].z - p[j].d);
            drop = drop || (dx * dx + dy * dy + dz * dz > thr * thr);
        }
        drop = drop && j < n;
        const int last = n - 1;
        MPt pl = p[0];
        f3 wl = wa[0];
#pragma unroll
        for (int k = 1; k < 4; k++) { mpt_sel(pl, p[k], k == last); f3_sel(wl, wa[k], k == last); }
        mpt_sel(p[j], pl, drop && j != last);
        f3_sel(wa[j], wl, drop && j != last);
        n = drop ? n - 1 : n;
    }
    // the new point: support vertex towards the plane (+ margin along that direction)
    f3 sv;
    if (M.hull_sides > 0) {
        float best = -3.0e38f;
        sv = mk3(0.f, 0.f, 0.f);
        for (int k = 0; k < 2 * M.hull_sides; k++) {      // the importer's order: (+z, -z) of vertex 0, 1, ...
            const f3 c = mk3(M.hull_xy[k >> 1][0], M.hull_xy[k >> 1][1], (k & 1) ? -M.cyl_hl : M.cyl_hl);
            const float val = dot(dl, c);
            if (val > best) { best = val; sv = c; }
        }
    } else {                                              // btCylinderShapeZ's support function [U]
        const float rr = sqrtf(dl.x * dl.x + dl.y * dl.y);
        sv = rr != 0.f ? mk3(M.cyl_r * dl.x / rr, M.cyl_r * dl.y / rr, 0.f) : mk3(M.cyl_r, 0.f, 0.f);
        sv.z = dl.z < 0.f ? -M.cyl_hl : M.cyl_hl;
    }
    MPt np;
    const f3 loc = sv + dl * M.margin;                    // in the cylinder's own (centred) frame
    np.a = loc + zoff;
    const f3 wnew = centre + mulRv(Rw, loc);
    np.d = wnew.z;
    np.b = mk3(wnew.x, wnew.y, 0.f);
    np.lam = 0.f;
    {
        const bool add = np.d < thr;
        int nearest = -1;
        float shortest = thr * thr;
        float lam_near = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f3 d = p[j].a - np.a;
            const float dd = dot(d, d);
            const bool nr = j < n && dd < shortest;
            shortest = nr ? dd : shortest; nearest = nr ? j : nearest; lam_near = nr ? p[j].lam : lam_near;
        }
        const int evict = manifold_sort_cached(p, np);
        // replaceContactPoint keeps the cached point's applied impulse; a point that is added, or that evicts another
        // one (addManifoldPoint -> sortCachedPoints), starts at zero [U]
        np.lam = nearest >= 0 ? lam_near : 0.f;
        const int where = nearest >= 0 ? nearest : (n < 4 ? n : evict);
        n = (add && nearest < 0 && n < 4) ? n + 1 : n;
#pragma unroll
        for (int j = 0; j < 4; j++) { mpt_sel(p[j], np, add && j == where); f3_sel(wa[j], wnew, add && j == where); }
    }
    return n;
}

// Which of a cylinder's n cached points get rows when the environment holds more points than the solve has slots for
// (`room` of them; Bullet has no such limit, snk_contact_overflow counts how often this build's is hit).  Every cylinder
// ranks its points the way Bullet's own manifold reduction values them (sortCachedPoints: the deepest point, then
// spread): first the deepest, second the one farthest from it (the other end cap), third the one that spans the
// larger triangle with those two, then the last; ties go to the lower manifold index.  Slots are handed out in
// passes: every cylinder's first point, in cylinder order, then every cylinder's second, ... until they are used up,
// so a resting snake keeps one point per end cap of every cylinder before any cylinder keeps a third.  Lane =
// cylinder; returns the lane's bit mask of kept points (the oracle's max_contacts mirrors the rule for the tests).
__device__ __forceinline__ int manifold_keep_mask(int n, const MPt (&p)[4], int lane, int total, int room) {
    if (total <= room) return (1 << n) - 1;
    int rank[4] = {4, 4, 4, 4};
    int p0 = 0, p1 = -1, p2 = -1;
    {
        float best = 3.0e38f;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < n && p[i].d < best) { best = p[i].d; p0 = i; }
    }
    f3 a0 = p[0].a;
#pragma unroll
    for (int i = 1; i < 4; i++) f3_sel(a0, p[i].a, i == p0);
    {
        float best = -1.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const f3 d = p[i].a - a0;
            const float dd = dot(d, d);
            if (i < n && i != p0 && dd > best) { best = dd; p1 = i; }
        }
    }
    f3 a1 = p[0].a;
#pragma unroll
    for (int i = 1; i < 4; i++) f3_sel(a1, p[i].a, i == p1);
    {
        float best = -1.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const f3 c = cross(a1 - a0, p[i].a - a0);
            const float cc = dot(c, c);
            if (i < n && i != p0 && i != p1 && cc > best) { best = cc; p2 = i; }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) rank[i] = i == p0 ? 0 : (i == p1 ? 1 : (i == p2 ? 2 : 3));
    int granted = 0;                       // passes in which this cylinder got a slot (monotone: once refused, refused)
    int left = room;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const unsigned long long el = __ballot(n > pass);
        const int before = __popcll(el & ((1ull << lane) - 1ull));
        if (n > pass && before < left) granted = pass + 1;
        left -= __popcll(el);
        left = left < 0 ? 0 : left;
    }
    int mask = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < n && rank[i] < granted) mask |= 1 << i;
    return mask;
}

// The streamed-row kernels keep the manifolds in global memory: mfc -> the cylinder's kMfFloats floats
// [count, 3 pad, 4 x (a3, b.x, b.y, lambda)], read and written write-through (sc1): an env-step moves between waves at
// substep boundaries, and the bytes must be where the next wave's loads look (same rule as the state record,
// store_rec_through).
__device__ __forceinline__ int manifold_load_global(const float* __restrict__ mfc, MPt (&p)[4]) {
    mf_v4 v[7];
    asm volatile(
        "global_load_dwordx4 %0, %7, off sc1\n\t"
        "global_load_dwordx4 %1, %7, off offset:16 sc1\n\t"
        "global_load_dwordx4 %2, %7, off offset:32 sc1\n\t"
        "global_load_dwordx4 %3, %7, off offset:48 sc1\n\t"
        "global_load_dwordx4 %4, %7, off offset:64 sc1\n\t"
        "global_load_dwordx4 %5, %7, off offset:80 sc1\n\t"
        "global_load_dwordx4 %6, %7, off offset:96 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6])
        : "v"(mfc)
        : "memory");
    float f[kMfFloats];
#pragma unroll
    for (int i = 0; i < 7; i++) { f[4 * i] = v[i].x; f[4 * i + 1] = v[i].y; f[4 * i + 2] = v[i].z; f[4 * i + 3] = v[i].w; }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        p[j].a = mk3(f[4 + 6 * j], f[5 + 6 * j], f[6 + 6 * j]);
        p[j].b = mk3(f[7 + 6 * j], f[8 + 6 * j], 0.f);
        p[j].lam = f[9 + 6 * j];
        p[j].d = 0.f;
    }
    const int n = (int)f[0];
    return n < 0 ? 0 : (n > 4 ? 4 : n);      // (manifold_core clamps as well; the count indexes registers here)
}
__device__ __forceinline__ void manifold_store_global(float* __restrict__ mfc, int n, const MPt (&p)[4]) {
    mf_v4 v[7];
    float f[kMfFloats];
    f[0] = (float)n; f[1] = 0.f; f[2] = 0.f; f[3] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f[4 + 6 * j] = p[j].a.x; f[5 + 6 * j] = p[j].a.y; f[6 + 6 * j] = p[j].a.z;
        f[7 + 6 * j] = p[j].b.x; f[8 + 6 * j] = p[j].b.y; f[9 + 6 * j] = p[j].lam;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) { v[i].x = f[4 * i]; v[i].y = f[4 * i + 1]; v[i].z = f[4 * i + 2]; v[i].w = f[4 * i + 3]; }
    asm volatile(
        "global_store_dwordx4 %7, %0, off sc1\n\t"
        "global_store_dwordx4 %7, %1, off offset:16 sc1\n\t"
        "global_store_dwordx4 %7, %2, off offset:32 sc1\n\t"
        "global_store_dwordx4 %7, %3, off offset:48 sc1\n\t"
        "global_store_dwordx4 %7, %4, off offset:64 sc1\n\t"
        "global_store_dwordx4 %7, %5, off offset:80 sc1\n\t"
        "global_store_dwordx4 %7, %6, off offset:96 sc1"
        :
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(mfc)
        : "memory");
}

// exclusive prefix sum over lanes of a small count (0..7), wave-uniform total in `total`
__device__ __forceinline__ int lane_prefix3(int cnt, int lane, int& total) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long b0 = __ballot(cnt & 1), b1 = __ballot(cnt & 2), b2 = __ballot(cnt & 4);
    total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
    return __popcll(b0 & below) + 2 * __popcll(b1 & below) + 4 * __popcll(b2 & below);
}

// Where cylinder `lane`'s points start in the compact contact list: the exclusive prefix of the kept counts in the ORDER THE
// SOLVER SWEEPS THE MANIFOLDS (snk_params::contact_order; 0 = cylinder = link order, the prefix over the lanes as they are).
// Otherwise position r fetches the count of the cylinder that sits there (DevModel::cyl_at), the prefix runs over
// positions, and every cylinder fetches its own from its position (cyl_rank): two ds_bpermute per substep.
// (wave-uniform branch: the model sits in constant memory)
__device__ __forceinline__ int manifold_base(const DevModel& M, int ncyl, int kept, int lane, int& total) {
    if (M.contact_order == 0) return lane_prefix3(kept, lane, total);
    const int at = lane < ncyl ? (int)M.cyl_at[lane] : lane;
    const int rk = lane < ncyl ? (int)M.cyl_rank[lane] : lane;
    const int kept_r = __shfl(kept, at);
    const int base_r = lane_prefix3(kept_r, lane, total);
    return __shfl(base_r, rk);
}

// Contacts with the obstacle box for the register-resident solve (lane = cylinder, the two-tier GJK of
// snk_selfcol.hpp): at most LT::kObs of them on this solve (a substep with more goes through the streamed-row solve,
// like one with more ground points than slots: find_contacts_v2 returns -1), appended behind the ground contacts.  The narrow phase runs BEFORE the ground contacts are compacted, because the slots the
// box's contacts take come out of the ground's share (64 in all).
struct ObsHit {
    bool hit;
    f3 P, n, dA, dB;
    float dist;
};
template <class LT>
__device__ __forceinline__ int find_obstacle_v2(LT& L, const DevModel& M, int lane, ObsHit& h,
                                                unsigned long long* __restrict__ ovf) {
    constexpr int N = LT::kN;
    h.hit = false;
    h.P = mk3(0, 0, 0); h.n = mk3(0, 0, 1); h.dA = mk3(0, 0, 0); h.dB = mk3(0, 0, 0);
    h.dist = 0.f;
    if (!M.obstacle) return 0;
    const float rb = sqrtf(M.cyl_r * M.cyl_r + M.cyl_hl * M.cyl_hl) + M.margin;
    Cvx Bx;
    Bx.c = mk3(M.obs_c[0], M.obs_c[1], M.obs_c[2]);
#pragma unroll
    for (int i = 0; i < 9; i++) Bx.R[i] = (i % 4 == 0) ? 1.f : 0.f;
    Bx.box = 1;
    Bx.half = mk3(M.obs_h[0], M.obs_h[1], M.obs_h[2]);
    const float rbox = sqrtf(dot(Bx.half, Bx.half));
    const int a = lane < 2 * N ? lane : 2 * N - 1;
    const int ba = (a + 1) >> 1;
    const f3 ca = ld3(L.o[ba]) + mulRv(L.R[ba], ld3(M.cyl_c[a]));
    const f3 d = ca - Bx.c;
    const float reach_ob = rb + rbox + M.break_thr;
    const f3 axa = mulRv(L.R[ba], mk3(M.cyl_R[a][2], M.cyl_R[a][5], M.cyl_R[a][8]));
    // (the box's bounding sphere is 0.42 m wide: every cylinder of a snake in front of it passes that test; the face
    //  normals of the box decide -- snk_selfcol.hpp)
    const bool cand = lane < 2 * N && dot(d, d) <= reach_ob * reach_ob && cyl_box_may_touch(M, ca, axa, Bx.c, Bx.R, Bx.half);
    if (!__any(cand)) return 0;
    if (cand) {
        Cvx A;
        cyl_world_rot(L.R[ba], M.cyl_R[a], A.R);
        A.c = ca;
        A.box = 0;
        A.half = mk3(0.f, 0.f, 0.f);
        f3 pa, pb;
        float mg = M.margin;
        float dd = gjk_distance(M, A, Bx, 0.f, pa, pb);
        if (dd < 0.f) {
            dd = gjk_distance(M, A, Bx, kShrink, pa, pb);
            mg = M.margin + kShrink;
        }
        if (dd < 0.f) {
            const float nn = sqrtf(dot(d, d));
            h.n = nn > 0.f ? d * (1.0f / nn) : mk3(-1.f, 0.f, 0.f);
            h.P = A.c;
            h.dist = -2.0f * mg;
        } else {
            h.n = (pa - pb) * (1.0f / dd);
            h.dist = dd - 2.0f * mg;
            h.P = pa - h.n * mg;
        }
        h.hit = h.dist < M.break_thr;
        plane_space(h.n, h.dA, h.dB);
        h.dA = aniso_scale(M, A.R, h.dA);
        h.dB = aniso_scale(M, A.R, h.dB) * M.fricB;
    }
    // (more than kObs of them: the caller sends the substep to the streamed-row solve, which has room for every cylinder)
    return __popcll(__ballot(h.hit));
}
// ... and their records, behind the nplane ground contacts
template <class LT>
__device__ __forceinline__ void append_obstacle_v2(LT& L, int lane, const ObsHit& h, int nplane) {
    constexpr int N = LT::kN;
    const unsigned long long bal = __ballot(h.hit);
    if (h.hit) {
        const int k = __popcll(bal & ((1ull << lane) - 1ull));
        if (k < LT::kObs) {
            const int idx = nplane + k;
            st3(L.ccP[idx], h.P);
            L.ccdist[idx] = h.dist;
            L.ccbody[idx] = (unsigned char)((lane + 1) >> 1);
            L.ccds[idx] = (unsigned char)(2 * N + k);
            st3(L.cdir[2 * N + k][0], h.dA);
            st3(L.cdir[2 * N + k][1], h.dB);
            st3(L.obn[k], h.n);
            L.app[kAppNormal + idx] = 0.f;       // (no contact cache for the box's contacts: nothing to warm-start from)
        }
    }
}

template <class LT>
__device__ int find_contacts_v2(LT& L, const DevModel& M, int lane, unsigned long long* __restrict__ ovf) {
    constexpr int N = LT::kN;
    static_assert(4 * N == 64, "one contact slot per lane");
    // link-link contacts have two-body rows, which only the streamed-row solve builds: a substep in which a pair of links
    // may be within the breaking threshold goes there (never under the reference's command range: snk_selfcol.hpp)
    if (M.self_collision && any_self_pair_v2(L, M, lane)) {
        if (lane == 0) atomicAdd(ovf, 1ull);
        return -1;
    }
    ObsHit oh;
    const int n_ob = find_obstacle_v2(L, M, lane, oh, ovf);      // 0 without an obstacle
    if (M.contact_model == 1) {
        // lane = cylinder; its manifold lives in LDS while this wave holds the environment (Lds<N, true>::mfl)
        int cnt = 0;
        MPt p[4];
        f3 wa[4];
        float Rw[9];
        const int c = lane < 2 * N ? lane : 0;
        const int b = (c + 1) >> 1;
        if (lane < 2 * N) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                p[j].a = mk3(L.mfl[6 * j][c], L.mfl[6 * j + 1][c], L.mfl[6 * j + 2][c]);
                p[j].b = mk3(L.mfl[6 * j + 3][c], L.mfl[6 * j + 4][c], 0.f);
                p[j].lam = L.mfl[6 * j + 5][c];
                p[j].d = 0.f;
            }
            const float* Rb = L.R[b];
            cyl_world_rot(Rb, M.cyl_R[c], Rw);
            const f3 dl = mk3(-Rw[6], -Rw[7], -Rw[8]);
            const f3 centre = ld3(L.o[b]) + mulRv(Rb, ld3(M.cyl_c[c]));
            cnt = manifold_core(M, (int)L.mfn[c], p, Rw, centre, dl, wa);
        }
        int total;
        (void)lane_prefix3(cnt, lane, total);
        const int room = 4 * N - n_ob;               // the solve's contact slots, less the box's contacts
        if (total > room || n_ob > LT::kObs) {
            // More points than this solve has slots for (a snake at rest gathers up to four per cylinder: 128).  Bullet
            // has no such limit, and neither has the streamed-row solve of this chain (128 + 32 slots): nothing has
            // been written yet, the caller runs THIS substep through that solve instead (snk_device.hpp: substep()).
            // Counted (snk_contact_overflow), never silent.
            if (lane == 0) atomicAdd(ovf, 1ull);
            return -1;
        }
        const int mask = (1 << cnt) - 1;
        const int kept = cnt;
        int tk;
        const int base = manifold_base(M, 2 * N, kept, lane, tk);
        if (lane < 2 * N) {
            L.cylbase[c] = (unsigned char)base;
            L.cyln[c] = (unsigned char)kept;
            L.cylkeep[c] = (unsigned char)mask;
            L.mfn[c] = (unsigned char)cnt;
            f3 dA, dB;
            friction_dirs(M, Rw, dA, dB);
            st3(L.cdir[c][0], dA);
            st3(L.cdir[c][1], dB);
            int idx = base;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool on = (mask >> j) & 1;
                if (!on) p[j].lam = 0.f;             // a point without rows carries no impulse
                if (j < cnt) {
                    L.mfl[6 * j][c] = p[j].a.x; L.mfl[6 * j + 1][c] = p[j].a.y; L.mfl[6 * j + 2][c] = p[j].a.z;
                    L.mfl[6 * j + 3][c] = p[j].b.x; L.mfl[6 * j + 4][c] = p[j].b.y; L.mfl[6 * j + 5][c] = p[j].lam;
                }
                if (on) {
                    st3(L.ccP[idx], wa[j]);
                    L.ccdist[idx] = p[j].d;
                    L.ccbody[idx] = (unsigned char)b;
                    L.ccds[idx] = (unsigned char)c;
                    // where the normal row starts (warm starting); the solve leaves the row's final impulse here
                    L.app[kAppNormal + idx] = M.warm_start ? p[j].lam * M.warm_factor : 0.f;
                    idx++;
                }
            }
        }
        if (lane == 0) L.nplane = tk;
        if (n_ob) append_obstacle_v2(L, lane, oh, tk);
        return tk + n_ob;
    }
    const int slot = lane;
    const int c = slot >> 1;
    const int b = (c + 1) >> 1;
    const float* Rb = L.R[b];
    float Rw[9];
    cyl_world_rot(Rb, M.cyl_R[c], Rw);
    f3 dl = mk3(-Rw[6], -Rw[7], -Rw[8]);
    float lx, ly;
    rim_point(M, dl, lx, ly);
    float lz = (slot & 1) ? M.cyl_hl : -M.cyl_hl;
    f3 loc = mk3(lx + M.margin * dl.x, ly + M.margin * dl.y, lz + M.margin * dl.z);
    f3 P = ld3(L.o[b]) + mulRv(Rb, ld3(M.cyl_c[c])) + mulRv(Rw, loc);
    const float dist = P.z;
    bool active = dist < M.break_thr;
    unsigned long long bal = __ballot(active);
    const int room = 4 * N - n_ob;                   // the solve's contact slots, less the box's contacts
    if (__popcll(bal) > room || n_ob > LT::kObs) {   // (only with an obstacle: two points per cylinder are 4 N at most)
        if (lane == 0) atomicAdd(ovf, 1ull);         // over to the streamed-row solve, as above
        return -1;
    }
    const int np_ = __popcll(bal);
    if (active) {
        const int idx = __popcll(bal & ((1ull << lane) - 1ull));
        st3(L.ccP[idx], P);
        L.ccdist[idx] = dist;
        L.ccbody[idx] = (unsigned char)b;
        L.ccds[idx] = (unsigned char)c;
        L.app[kAppNormal + idx] = 0.f;       // no contact cache, nothing to warm-start from
    }
    if (lane < 2 * N) {      // lane = cylinder: the friction directions its contacts share
        float Rc[9];
        cyl_world_rot(L.R[(lane + 1) >> 1], M.cyl_R[lane], Rc);
        f3 dA, dB;
        friction_dirs(M, Rc, dA, dB);
        st3(L.cdir[lane][0], dA);
        st3(L.cdir[lane][1], dB);
    }
    if (lane < 2 * N) {      // lane = cylinder: where its (up to two) contacts sit in the compact list
        L.cylbase[lane] = (unsigned char)__popcll(bal & ((1ull << (2 * lane)) - 1ull));
        L.cyln[lane] = (unsigned char)((int)((bal >> (2 * lane)) & 1ull) + (int)((bal >> (2 * lane + 1)) & 1ull));
    }
    if (lane == 0) L.nplane = np_;
    if (n_ob) append_obstacle_v2(L, lane, oh, np_);
    return np_ + n_ob;
}

// The same persistent-manifold contacts for the streamed-row solve (chains up to 32 links: one cylinder per lane):
// the manifolds in global memory, geometry records written at the COMPACT index (clist is the identity), at most NC
// ground contacts (the same rule for which points keep their rows, the same counters).  geo[19] of a record: the
// impulse its normal row starts from (warm starting).
template <class LT>
__device__ int find_contacts_manifold_v1(LT& L, const DevModel& M, int lane, float* __restrict__ rows, float* __restrict__ mf,
                                         unsigned long long* __restrict__ ovf) {
    constexpr int N = LT::kN;
    static_assert(2 * N <= 64, "one cylinder per lane");
    int cnt = 0;
    MPt p[4];
    f3 wa[4];
    float Rw[9];
    const int c = lane < 2 * N ? lane : 0;
    const int b = (c + 1) >> 1;
    if (lane < 2 * N) {
        const float* Rb = L.R[b];
        cyl_world_rot(Rb, M.cyl_R[c], Rw);
        const f3 dl = mk3(-Rw[6], -Rw[7], -Rw[8]);
        const f3 centre = ld3(L.o[b]) + mulRv(Rb, ld3(M.cyl_c[c]));
        const int n0 = manifold_load_global(mf + (size_t)c * kMfFloats, p);
        cnt = manifold_core(M, n0, p, Rw, centre, dl, wa);
    }
    int total;
    (void)lane_prefix3(cnt, lane, total);
    const int room = LT::NC;
    const int mask = manifold_keep_mask(cnt, p, lane, total, room);
    const int kept = __popc(mask);
    int tk;
    const int base = manifold_base(M, 2 * N, kept, lane, tk);
    if (total > room && lane == 0) {
        atomicAdd(ovf, 1ull);
        atomicAdd(ovf + 1, (unsigned long long)(total - tk));
    }
    if (lane < 2 * N) {
        // the sensor pass finds a body's contacts through its cylinders: [first compact index, count] of cylinder `lane`
        // (the slot -> contact table of the stateless model is not used in this one; its array holds these pairs) and,
        // in the upper bits of the count word, which cached points they are (for the impulses' way back into the cache)
        L.cidx[2 * lane] = base;
        L.cidx[2 * lane + 1] = kept | (mask << 8);
        f3 dA, dB;
        friction_dirs(M, Rw, dA, dB);
        int idx = base;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool on = (mask >> j) & 1;
            if (!on) p[j].lam = 0.f;
            if (on) {
                float* geo = rows + LT::kGeoOff + (size_t)idx * LT::kGeo;
                st3(geo, wa[j]);
                geo[3] = p[j].d;
                st3(geo + 4, dA);
                st3(geo + 7, dB);
                st3(geo + 10, mk3(0.f, 0.f, 1.f));
                st3(geo + 13, mk3(0.f, 0.f, 0.f));
                geo[16] = (float)b; geo[17] = -1.0f; geo[18] = 1.0f;
                geo[19] = M.warm_start ? p[j].lam * M.warm_factor : 0.f;
                L.clist[idx] = idx;
                idx++;
            }
        }
        manifold_store_global(mf + (size_t)c * kMfFloats, cnt, p);
    }
    return tk;
}

// ------------------------------------------------------------------------------------
// one batch of rows, lane = row: M^-1 J^T by the ABA delta sweeps
// (btMultiBody::calcAccelerationDeltasMultiDof [U]) plus the row's denominator and
// right-hand side.  KIND 0 motor (-> L.Mm, kept for the limit rows), 1 normal (contact = lane),
// 4 friction pairs of 32 contacts: lane -> contact base + (lane & 31), direction lane >> 5
// (A in staging rows 0..31, B in rows 32..63, so one pass over the slots fills both halves).
// Staging row layout: [0..21] M^-1 J^T, [22] rhs impulse, [23] denominator, [24] 1/denominator.
// ------------------------------------------------------------------------------------
template <class LT, int KIND>
__device__ void build_batch_v2(LT& L, const DevModel& M, int lane, int nc, int base = 0) {
    constexpr int N = LT::kN;
    const int ci = KIND == 4 ? base + (lane & 31) : lane;      // contact (or motor) of this lane
    if (ci < (KIND == 0 ? N : nc)) {
        const bool motor = KIND == 0;
        int k;
        f3 P = mk3(0, 0, 0), d = mk3(0, 0, 0);
        float* Mrow = motor ? L.Mm[lane] : L.stM[lane];
        if (motor) {
            k = lane + 1;
        } else {
            k = L.ccbody[ci];
            P = ld3(L.ccP[ci]);
            // a ground contact's normal is +z, the obstacle's contacts (behind the nplane ground contacts) carry theirs
            d = KIND == 1 ? (ci < L.nplane ? mk3(0.f, 0.f, 1.f) : ld3(L.obn[ci - L.nplane])) : ld3(L.cdir[L.ccds[ci]][lane >> 5]);
        }
        f3 pN = mk3(0, 0, 0), pF = mk3(0, 0, 0);
#pragma unroll 4
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
#pragma unroll 4
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
        if (motor) S[3] = target;
    }
    lds_sync();
}


// per-lane constants of the (half, d) layout
struct LaneK {
    int h, d;
    bool isdof;
    float m22, m24;  // -1 at d==22 / +1 at d==24, else 0
    float m31;       // 1 at d==31 (the accumulated impulse's lane), else 0
    int spoff;       // staging column that feeds this lane's special value (22 rhs, 23 den)
    f3 oL, aL;       // for d < 3 and d >= 6:  J[d] = aL . ((P - oL) x dir);  3 <= d < 6: dir[d-3]
    int bL;          // joint index (body) of lane d, 0 for the base components
};

// Fill the halves of one register slot from the rows staged by the last batch.
// KIND 1 normal (contact 2s + h; both halves), 2 friction A (contact s -> lower half only),
// 3 friction B (contact s -> upper half only).
//   RJ: d < 22  J[d] / den * sJ,   d == 22  -rhs * sJ,   d == 31  accumulated impulse (0)
//   RM: d < 22  (M^-1 J^T)[d] * sM, d == 24  den * sM
// Friction rows are built in units of mu (sJ = 1/mu, sM = mu): the impulse variable is
// f / mu, so the cone radius is lambda_n itself and mu never appears in the solve loop;
// dI * den (the residual) is unchanged.  mu == 0 gives inert zero rows.
// The LDS reads of a slot (issued one slot ahead of the arithmetic that consumes them: a
// dependent ds_read round trip costs ~200 clocks, as much as the whole slot's arithmetic).
struct SlotRaw {
    float mval, dinv, sp, lam;
    f3 P, dir;
    int k;
    bool valid;
};
template <class LT, int KIND>
__device__ __forceinline__ SlotRaw fetch_slot(LT& L, const LaneK& K, int s, int count, int base, int nplane) {
    SlotRaw r;
    // KIND 1: slot s holds contacts 2s, 2s+1 (staging row = contact);  KIND 4: slot s = contact s,
    // direction A / B in the lower / upper half, staged in rows (s - base) and 32 + (s - base)
    const int ci = KIND == 1 ? 2 * s + K.h : s;
    r.valid = ci < count;
    const int rs = r.valid ? ci : 0;
    const int row = KIND == 1 ? rs : (r.valid ? 32 * K.h + (s - base) : 0);
    const float* st = L.stM[row];
    const int dd = K.isdof ? K.d : 0;
    r.mval = st[dd];
    r.dinv = st[24];
    r.sp = st[22 + K.spoff];
    r.P = ld3(L.ccP[rs]);
    r.dir = KIND == 1 ? (rs < nplane ? mk3(0.f, 0.f, 1.f) : ld3(L.obn[rs - nplane])) : ld3(L.cdir[L.ccds[rs]][K.h]);
    r.k = L.ccbody[rs];
    r.lam = (KIND == 1 && r.valid) ? L.app[kAppNormal + rs] : 0.f;     // where the normal row starts (warm starting)
    return r;
}
template <int KIND>
__device__ __forceinline__ void finish_slot(const SlotRaw& r, const LaneK& K, float sJ, float sM, float& RJ, float& RM,
                                            float& wsum) {
    float v = dot(K.aL, cross(r.P - K.oL, r.dir));
    v = K.d == 3 ? r.dir.x : (K.d == 4 ? r.dir.y : (K.d == 5 ? r.dir.z : v));
    const float jd = (K.bL <= r.k) ? v : 0.0f;
    float rj = (K.isdof ? jd * r.dinv : r.sp * K.m22) * sJ;
    float rm = (K.isdof ? r.mval : r.sp * K.m24) * sM;
    if (!r.valid) { rj = 0.f; rm = 0.f; }
    if (KIND == 1) {
        rj += K.m31 * r.lam;
        wsum += rm * r.lam;
    }
    RJ = rj;
    RM = rm;
}
// eight consecutive slots BASE .. BASE+7 of one kind, reads issued one slot ahead (two ahead
// measured the same and spills)
// friction rows are written in units of their contact's friction coefficient (finish_slot): the ground's for the first
// nplane contacts, the obstacle's for those behind them
struct MuScale {
    float fJ, fM, fJo, fMo;
    int nplane;
};
template <class LT, int KIND, int BASE, int DST>
__device__ __forceinline__ void load_slots8(LT& L, const LaneK& K, int count, int base, const MuScale& ms,
                                            float (&RJ)[kSlots], float (&RM)[kSlots], float& wsum) {
    SlotRaw cur = fetch_slot<LT, KIND>(L, K, BASE, count, base, ms.nplane);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        SlotRaw nxt = cur;
        if (i < 7) nxt = fetch_slot<LT, KIND>(L, K, BASE + i + 1, count, base, ms.nplane);
        const bool gr = KIND == 1 || BASE + i < ms.nplane;           // (KIND 4: slot = contact)
        finish_slot<KIND>(cur, K, KIND == 1 ? 1.0f : (gr ? ms.fJ : ms.fJo), KIND == 1 ? 1.0f : (gr ? ms.fM : ms.fMo),
                          RJ[DST + BASE + i], RM[DST + BASE + i], wsum);
        cur = nxt;
    }
}

// old with lane l replaced by a wave-uniform value (once per substep: a select is fine)
__device__ __forceinline__ float wrlane(float old, float v_uniform, int l) {
    return ((int)threadIdx.x == l) ? v_uniform : old;
}

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
    "s_nop 0\n\t"
// Wait states used below are the ones hipcc itself inserts on gfx950: VALU write -> DPP read 2,
// VALU write -> v_readlane 1, v_readlane (SGPR) -> VALU read 2, v_rsq -> VALU use 1,
// VALU write -> v_permlane32_swap 2, v_permlane32_swap -> VALU use 0.  Every s_nop is an issue
// slot of the wave, so none is spent beyond those (measured: -15 % per step).

// Two consecutive single rows living in the two halves of one slot (hand-written, 25 VALU):
// lower-half row first, then the upper-half row.  One multiply and one reduction give both
// sums  -(rhs - J.dv/den + a)  in lanes 31 / 63; they, the accumulated impulses (lanes 31 / 63
// of RJ) are read into SGPRs (the coupling scalar c stays in lane 62 of RJ, see below) and the clamps run on
// wave-uniform values:  a' = med3(-sum, LO, HI), dI = a' - a;  the upper row's sum first
// receives c * dI_lower.  v_cndmask gives every lane its own half's dI; the two contributions
// to delta-v cross halves with v_permlane32_swap.  RES: also max |dI * den| (den: lane 24 / 56
// of RM) into lsq.
// 22 VALU (round 2; 25 before): the coupling scalar sits in lane d = 30 of RJ (lanes 30 / 62) and reaches the upper
// row's partial sum through ONE v_fmac_f32_dpp (row_shr:1 puts RJ[62] under lane 63) -- no v_readlane of the scalar,
// no v_mov of the upper sum into a VGPR; the price is that the upper sum is read after the lower row's clamp
// (one scalar round trip longer, irrelevant: the kernel is VALU-issue-bound).
#define SNK_DUO_HEAD                                     \
    "v_mul_f32 %[t], %[RJ], %[dv]\n\t"                    \
    "v_readlane_b32 %[s2], %[RJ], 31\n\t"                 \
    "v_readlane_b32 %[s3], %[RJ], 63\n\t"                 \
    SNK_REDUCE_12                                          \
    "s_nop 1\n\t"                                         \
    SNK_REDUCE_22                                          \
    "s_nop 1\n\t"                                         \
    SNK_REDUCE_345                                         \
    "v_readlane_b32 %[s0], %[t], 31\n\t"                  \
    "s_nop 1\n\t"
// The accumulated impulse of a row lives in lane 31 (lower-half row) / 63 (upper-half row) of its RJ.  Adding the
// step's dI there is ONE DPP add and needs no mask register: T holds dI_lower in lanes 0..31 and dI_upper in lanes
// 32..63 (it comes out of a v_cndmask on the half mask), row_shr:15 has a valid source only in lane 15 of a row --
// the other lanes are not written (bound_ctrl off) -- and row_mask 0xa keeps rows 1 and 3: lanes 31 and 63 get
// T[16] + RJ[31] and T[48] + RJ[63].  (Rounds 1-2 multiplied by a 0/1 mask held in a VGPR for the whole solve; the
// solve has no register to spare: DESIGN.md 4.)
#define SNK_ACC3163(RJ_, T_) "v_add_f32_dpp " RJ_ ", " T_ ", " RJ_ " row_shr:15 row_mask:0xa bank_mask:0xf\n\t"
#define SNK_DUO_TAIL                                     \
    "v_cndmask_b32_e64 %[t], %[dS], %[dF], %[lowmask]\n\t" \
    "v_mul_f32 %[x], %[RM], %[t]\n\t"                     \
    "v_mul_f32 %[c2], %[RM], %[t]\n\t"                    \
    SNK_ACC3163("%[RJ]", "%[t]")                           \
    "s_nop 0\n\t"                                         \
    "v_permlane32_swap_b32 %[x], %[c2]\n\t"               \
    "v_add_f32 %[dv], %[dv], %[x]\n\t"                    \
    "v_add_f32 %[dv], %[dv], %[c2]\n\t"
// BOX = false: contact normals, bounds [0, inf) -> a single v_max; BOX = true: [-HI, HI]
template <bool RES, bool BOX>
__device__ __forceinline__ void duo_step(float& RJ, const float RM, float& dv, float HI, float E3163,
                                         unsigned long long lowmask, float& lsq) {
    float t, x, dF, dS, c2;
    float s0, s1, s2, s3;
    if (!BOX) {
        asm volatile(
            SNK_DUO_HEAD
            "v_max_f32_e64 %[dF], -%[s0], 0\n\t"
            "v_subrev_f32 %[dF], %[s2], %[dF]\n\t"
            "s_nop 0\n\t"
            "v_fmac_f32_dpp %[t], %[RJ], %[dF] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %[s1], %[t], 63\n\t"
            "s_nop 1\n\t"
            "v_max_f32_e64 %[x], -%[s1], 0\n\t"
            "v_subrev_f32 %[dS], %[s3], %[x]\n\t"
            SNK_DUO_TAIL
            : [t] "=&v"(t), [x] "=&v"(x), [dF] "=&v"(dF), [dS] "=&v"(dS), [c2] "=&v"(c2), [s0] "=&s"(s0), [s1] "=&s"(s1),
              [s2] "=&s"(s2), [s3] "=&s"(s3), [RJ] "+v"(RJ), [dv] "+v"(dv)
            : [RM] "v"(RM), [lowmask] "s"(lowmask));
    } else
    asm volatile(
        SNK_DUO_HEAD
        "v_med3_f32 %[dF], -%[s0], -%[HI], %[HI]\n\t"
        "v_subrev_f32 %[dF], %[s2], %[dF]\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %[t], %[RJ], %[dF] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[s1], %[t], 63\n\t"
        "s_nop 1\n\t"
        "v_med3_f32 %[x], -%[s1], -%[HI], %[HI]\n\t"
        "v_subrev_f32 %[dS], %[s3], %[x]\n\t"
        SNK_DUO_TAIL
        : [t] "=&v"(t), [x] "=&v"(x), [dF] "=&v"(dF), [dS] "=&v"(dS), [c2] "=&v"(c2), [s0] "=&s"(s0), [s1] "=&s"(s1),
          [s2] "=&s"(s2), [s3] "=&s"(s3), [RJ] "+v"(RJ), [dv] "+v"(dv)
        : [RM] "v"(RM), [HI] "v"(HI), [lowmask] "s"(lowmask));
    if (RES) asm volatile("v_max3_f32 %[lsq], %[lsq], |%[x]|, |%[c2]|" : [lsq] "+v"(lsq) : [x] "v"(x), [c2] "v"(c2));
}

// one DPP step of two interleaved half-reductions (each instruction is the other's wait state)
#define SNK_RED2(MODE)                                        \
    "v_add_f32_dpp %[t1], %[t1], %[t1] " MODE "\n\t"           \
    "v_add_f32_dpp %[t2], %[t2], %[t2] " MODE "\n\t"
// FOUR consecutive contact normals (two slots) in one step, 44 VALU like two duo steps, but with the two reductions
// interleaved (no s_nop) and only the four scalar round trips in sequence.  Row order r1 (slot 1 lower), r2 (slot 1
// upper), r3, r4 as in the sequential sweep; later rows see earlier ones through coupling scalars parked in RJ:
//   RJ1 lane 62: c(r2,r1);   RJ2 lanes 29 / 61: c(r3,r1), c(r4,r1);   lanes 28 / 60: c(r3,r2), c(r4,r2);   lane 62: c(r4,r3)
// each applied to lanes 31 / 63 of the partial sums by one v_fmac_f32_dpp (row_shr:1, 2, 3).
template <bool RES>
__device__ __forceinline__ void quad_step(float& RJ1, const float RM1, float& RJ2, const float RM2, float& dv, float E3163,
                                          unsigned long long lowmask, float& lsq) {
    float t1, t2, d1, d2, d3, d4;
    float a1, a2, a3, a4, s1, s2, s3, s4;
    asm volatile(
        // (between two DPP steps of the same sum stand the other sum's step and ONE more instruction: the two wait
        //  states a dependent DPP read needs; the four reads of the accumulated impulses are those instructions)
        "v_mul_f32 %[t1], %[RJ1], %[dv]\n\t"
        "v_mul_f32 %[t2], %[RJ2], %[dv]\n\t"
        "v_readlane_b32 %[a1], %[RJ1], 31\n\t"
        SNK_RED2("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        "v_readlane_b32 %[a2], %[RJ1], 63\n\t"
        SNK_RED2("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        "v_readlane_b32 %[a3], %[RJ2], 31\n\t"
        SNK_RED2("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        "v_readlane_b32 %[a4], %[RJ2], 63\n\t"
        SNK_RED2("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        "s_nop 0\n\t"
        SNK_RED2("row_bcast:15 row_mask:0xa bank_mask:0xf")
        "v_readlane_b32 %[s1], %[t1], 31\n\t"
        "s_nop 1\n\t"
        "v_max_f32_e64 %[d1], -%[s1], 0\n\t"
        "v_subrev_f32 %[d1], %[a1], %[d1]\n\t"
        "v_fmac_f32_dpp %[t1], %[RJ1], %[d1] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %[t2], %[RJ2], %[d1] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_readlane_b32 %[s2], %[t1], 63\n\t"
        "s_nop 1\n\t"
        "v_max_f32_e64 %[d2], -%[s2], 0\n\t"
        "v_subrev_f32 %[d2], %[a2], %[d2]\n\t"
        "v_fmac_f32_dpp %[t2], %[RJ2], %[d2] row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[s3], %[t2], 31\n\t"
        "s_nop 1\n\t"
        "v_max_f32_e64 %[d3], -%[s3], 0\n\t"
        "v_subrev_f32 %[d3], %[a3], %[d3]\n\t"
        "v_fmac_f32_dpp %[t2], %[RJ2], %[d3] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %[s4], %[t2], 63\n\t"
        "s_nop 1\n\t"
        "v_max_f32_e64 %[d4], -%[s4], 0\n\t"
        "v_subrev_f32 %[d4], %[a4], %[d4]\n\t"
        // per-half impulses, the two slots' contributions
        "v_cndmask_b32_e64 %[d1], %[d2], %[d1], %[lowmask]\n\t"
        "v_cndmask_b32_e64 %[d3], %[d4], %[d3], %[lowmask]\n\t"
        "v_mul_f32 %[t1], %[RM1], %[d1]\n\t"
        "v_mul_f32 %[t2], %[RM2], %[d3]\n\t"
        SNK_ACC3163("%[RJ1]", "%[d1]")
        SNK_ACC3163("%[RJ2]", "%[d3]")
        : [t1] "=&v"(t1), [t2] "=&v"(t2), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [d4] "=&v"(d4),
          [a1] "=&s"(a1), [a2] "=&s"(a2), [a3] "=&s"(a3), [a4] "=&s"(a4), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3),
          [s4] "=&s"(s4), [RJ1] "+v"(RJ1), [RJ2] "+v"(RJ2)
        : [RM1] "v"(RM1), [RM2] "v"(RM2), [dv] "v"(dv), [lowmask] "s"(lowmask));
    if (RES) asm volatile("v_max3_f32 %[lsq], %[lsq], |%[p1]|, |%[p2]|" : [lsq] "+v"(lsq) : [p1] "v"(t1), [p2] "v"(t2));
    asm volatile(
        "v_add_f32 %[p1], %[p1], %[p2]\n\t"
        "v_mov_b32 %[p2], %[p1]\n\t"
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %[p1], %[p2]\n\t"
        "v_add_f32 %[dv], %[dv], %[p1]\n\t"
        "v_add_f32 %[dv], %[dv], %[p2]\n\t"
        : [p1] "+v"(t1), [p2] "+v"(t2), [dv] "+v"(dv));
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
        "s_nop 0\n\t"
        "v_fma_f32 %[r2], %[s1], %[s1], 1\n\t"
        "v_fma_f32 %[r2], %[s3], %[s3], %[r2]\n\t"
        "v_rsq_f32 %[r2], %[r2]\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_e64 %[r2], %[s4], %[r2] clamp\n\t"
        "v_fma_f32 %[xA], %[r2], -%[s1], -%[xA]\n\t"
        "v_fma_f32 %[xB], %[r2], -%[s3], -%[xB]\n\t"
        "v_cndmask_b32_e64 %[t], %[xB], %[xA], %[lowmask]\n\t"
        "v_mul_f32 %[r2], %[RM], %[t]\n\t"
        "v_mul_f32 %[c2], %[RM], %[t]\n\t"
        SNK_ACC3163("%[RJ]", "%[t]")
        "s_nop 0\n\t"
        "v_permlane32_swap_b32 %[r2], %[c2]\n\t"
        "v_add_f32 %[dv], %[dv], %[r2]\n\t"
        "v_add_f32 %[dv], %[dv], %[c2]\n\t"
        : [t] "=&v"(t), [xA] "=&v"(xA), [xB] "=&v"(xB), [r2] "=&v"(r2), [c2] "=&v"(c2),
          [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [RJ] "+v"(RJ), [dv] "+v"(dv)
        : [RM] "v"(RM), [RJnorm] "v"(RJnorm), [lowmask] "s"(lowmask),
          [NLn] "n"(NL));
    if (RES) asm volatile("v_max3_f32 %[lsq], %[lsq], |%[r2]|, |%[c2]|" : [lsq] "+v"(lsq) : [r2] "v"(r2), [c2] "v"(c2));
}

// TWO consecutive friction pairs (contacts c0, c0+1) in one step, hand-written, 52 VALU against 2 x 28: the two
// reductions run interleaved (each DPP step is the other's wait state, no s_nop at all), and the second pair still
// sees the first pair's update exactly as in the sequential sweep: its two sums are corrected by
//     c(X2,A1) dI_A1 + c(X2,B1) dI_B1,   c(X2,Y1) = RJ_X2 . RM_Y1   (X, Y in {A, B}),
// four scalars per pair of contacts, computed once per substep (cone2_couplings) and kept in spare lanes of RJ2
// (d = 30: coupling with A1, d = 29: with B1) so that two v_fmac_f32_dpp (row_shr:1 / row_shr:2) add them to lanes
// 31 / 63 of the partial sums without any scalar traffic.  Same arithmetic per row as cone_step; what changes is
// the dependency chain (one reduction + one scalar round trip per two pairs instead of two).
// Since round 2's last third the A / B halves of each pair's wave-uniform arithmetic go through gfx950's packed fp32
// instructions: the accumulated impulses (a_A, a_B) are read into an SGPR PAIR and copied with one v_mov_b64, the two
// `new - a` updates are one v_pk_fma_f32 with the two sums in an SGPR pair -- 52 VALU instead of 56.  Inline asm cannot
// name the halves of a 64-bit operand, so these temporaries are PINNED registers (v250..v255, s84..s89), declared as
// clobbers: the compiler keeps nothing in them across a step.
#define SNK_STR_(x) #x
#define SNK_V1(a) "v" SNK_STR_(a)
#define SNK_S1(a) "s" SNK_STR_(a)
#define SNK_VP(a, b) "v[" SNK_STR_(a) ":" SNK_STR_(b) "]"
#define SNK_SP(a, b) "s[" SNK_STR_(a) ":" SNK_STR_(b) "]"
#define SNK_CONE2_BODY_(P0, P1, P2, P3, P4, P5, Q0, Q1, Q2, Q3, Q4, Q5) \
    "v_mul_f32 %[t1], %[RJ1], %[dv]\n\t" \
    "v_mul_f32 %[t2], %[RJ2], %[dv]\n\t" \
    "v_readlane_b32 " SNK_S1(Q0) ", %[RJ1], 31\n\t" \
    "v_readlane_b32 " SNK_S1(Q1) ", %[RJ1], 63\n\t" \
    SNK_RED2("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    "v_readlane_b32 " SNK_S1(Q2) ", %[RJ2], 31\n\t" \
    "v_readlane_b32 " SNK_S1(Q3) ", %[RJ2], 63\n\t" \
    SNK_RED2("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    "v_readlane_b32 %[l1], %[RJn], 31\n\t" \
    "v_readlane_b32 %[l2], %[RJn], 63\n\t" \
    SNK_RED2("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    "v_mov_b64 " SNK_VP(P0, P1) ", " SNK_SP(Q0, Q1) "\n\t" \
    SNK_RED2("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    "v_mov_b64 " SNK_VP(P2, P3) ", " SNK_SP(Q2, Q3) "\n\t" \
    SNK_RED2("row_bcast:15 row_mask:0xa bank_mask:0xf") \
    "v_readlane_b32 " SNK_S1(Q4) ", %[t1], 31\n\t" \
    "v_readlane_b32 " SNK_S1(Q5) ", %[t1], 63\n\t" \
    "s_nop 0\n\t" \
    "v_fma_f32 " SNK_V1(P4) ", " SNK_S1(Q4) ", " SNK_S1(Q4) ", 1\n\t" \
    "v_fma_f32 " SNK_V1(P4) ", " SNK_S1(Q5) ", " SNK_S1(Q5) ", " SNK_V1(P4) "\n\t" \
    "v_rsq_f32 " SNK_V1(P4) ", " SNK_V1(P4) "\n\t" \
    "s_nop 0\n\t" \
    "v_mul_f32_e64 " SNK_V1(P4) ", %[l1], " SNK_V1(P4) " clamp\n\t" \
    "v_pk_fma_f32 " SNK_VP(P0, P1) ", " SNK_VP(P4, P5) ", " SNK_SP(Q4, Q5) ", " SNK_VP(P0, P1) " op_sel_hi:[0,1,1] neg_lo:[0,1,1] neg_hi:[0,1,1]\n\t" \
    "v_fmac_f32_dpp %[t2], %[RJ2], " SNK_V1(P0) " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_fmac_f32_dpp %[t2], %[RJ2], " SNK_V1(P1) " row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_cndmask_b32_e64 " SNK_V1(P0) ", " SNK_V1(P1) ", " SNK_V1(P0) ", %[lowmask]\n\t" \
    "v_readlane_b32 " SNK_S1(Q4) ", %[t2], 31\n\t" \
    "v_readlane_b32 " SNK_S1(Q5) ", %[t2], 63\n\t" \
    "v_mul_f32 " SNK_V1(P1) ", %[RM1], " SNK_V1(P0) "\n\t" \
    SNK_ACC3163("%[RJ1]", SNK_V1(P0)) \
    "v_fma_f32 " SNK_V1(P4) ", " SNK_S1(Q4) ", " SNK_S1(Q4) ", 1\n\t" \
    "v_fma_f32 " SNK_V1(P4) ", " SNK_S1(Q5) ", " SNK_S1(Q5) ", " SNK_V1(P4) "\n\t" \
    "v_rsq_f32 " SNK_V1(P4) ", " SNK_V1(P4) "\n\t" \
    "s_nop 0\n\t" \
    "v_mul_f32_e64 " SNK_V1(P4) ", %[l2], " SNK_V1(P4) " clamp\n\t" \
    "v_pk_fma_f32 " SNK_VP(P2, P3) ", " SNK_VP(P4, P5) ", " SNK_SP(Q4, Q5) ", " SNK_VP(P2, P3) " op_sel_hi:[0,1,1] neg_lo:[0,1,1] neg_hi:[0,1,1]\n\t" \
    "v_cndmask_b32_e64 " SNK_V1(P2) ", " SNK_V1(P3) ", " SNK_V1(P2) ", %[lowmask]\n\t" \
    "v_mul_f32 " SNK_V1(P3) ", %[RM2], " SNK_V1(P2) "\n\t"
#define SNK_CONE2_TAIL_(P0, P1, P2, P3, P4, P5, Q0, Q1, Q2, Q3, Q4, Q5)                           \
    "v_add_f32 " SNK_V1(P1) ", " SNK_V1(P1) ", " SNK_V1(P3) "\n\t"           \
    "v_mov_b32 " SNK_V1(P3) ", " SNK_V1(P1) "\n\t"                 \
    SNK_ACC3163("%[RJ2]", SNK_V1(P2))                \
    "s_nop 0\n\t"                              \
    "v_permlane32_swap_b32 " SNK_V1(P1) ", " SNK_V1(P3) "\n\t"     \
    "v_add_f32 %[dv], %[dv], " SNK_V1(P1) "\n\t"         \
    "v_add_f32 %[dv], %[dv], " SNK_V1(P3) "\n\t"
// PIN 1: the pinned temporaries of a kernel built for three waves per SIMD (168 registers): v162..v167, s84..s89
#define SNK_PINS_HI 250, 251, 252, 253, 254, 255, 84, 85, 86, 87, 88, 89
#define SNK_PINS_LO 162, 163, 164, 165, 166, 167, 84, 85, 86, 87, 88, 89
#define SNK_PINS_128 122, 123, 124, 125, 126, 127, 84, 85, 86, 87, 88, 89      /* (tools/ubench_solve.hip: four waves) */
#define SNK_CONE2_X(M_, ...) M_(__VA_ARGS__)
#define SNK_CONE2_ASM(RESLINE, PINS, CLOB)                                                                               \
    asm volatile(SNK_CONE2_X(SNK_CONE2_BODY_, PINS) RESLINE SNK_CONE2_X(SNK_CONE2_TAIL_, PINS)                              \
                 : [t1] "=&v"(t1), [t2] "=&v"(t2), [l1] "=&s"(l1), [l2] "=&s"(l2), [RJ1] "+v"(RJ1), [RJ2] "+v"(RJ2),       \
                   [dv] "+v"(dv), [lsq] "+v"(lsq)                                                                         \
                 : [RM1] "v"(RM1), [RM2] "v"(RM2), [RJn] "v"(RJnorm), [lowmask] "s"(lowmask)                             \
                 : CLOB, "s84", "s85", "s86", "s87", "s88", "s89")
template <bool RES, int PIN = 0>
__device__ __forceinline__ void cone2_step(float& RJ1, const float RM1, float& RJ2, const float RM2, const float RJnorm,
                                           float& dv, float EPS, float E3163, unsigned long long lowmask, float& lsq) {
    float t1, t2;
    float l1, l2;
    // v[250:251] = (x1A, x1B), v[252:253] = (x2A, x2B), v254 = the pair's scale factor (v255: the pair's unused half)
    // s[84:85] = (a1A, a1B), s[86:87] = (a2A, a2B), s[88:89] = the current pair's two sums.
    // After the body v251 = RM1 dI_1, v253 = RM2 dI_2 (lane 24 / 56: den dI of the A / B rows: the residual); ONE asm
    // statement, so that nothing can be scheduled into the pinned registers on the way
#define SNK_CLOB_HI "v250", "v251", "v252", "v253", "v254", "v255"
#define SNK_CLOB_LO "v162", "v163", "v164", "v165", "v166", "v167"
#define SNK_CLOB_128 "v122", "v123", "v124", "v125", "v126", "v127"
    if constexpr (PIN == 0) {
        if (RES) SNK_CONE2_ASM("v_max3_f32 %[lsq], %[lsq], |v251|, |v253|\n\t", SNK_PINS_HI, SNK_CLOB_HI);
        else SNK_CONE2_ASM("", SNK_PINS_HI, SNK_CLOB_HI);
    } else if constexpr (PIN == 1) {
        if (RES) SNK_CONE2_ASM("v_max3_f32 %[lsq], %[lsq], |v163|, |v165|\n\t", SNK_PINS_LO, SNK_CLOB_LO);
        else SNK_CONE2_ASM("", SNK_PINS_LO, SNK_CLOB_LO);
    } else {
        if (RES) SNK_CONE2_ASM("v_max3_f32 %[lsq], %[lsq], |v123|, |v125|\n\t", SNK_PINS_128, SNK_CLOB_128);
        else SNK_CONE2_ASM("", SNK_PINS_128, SNK_CLOB_128);
    }
}

// The four coupling scalars of cone2_step for the friction slots (S, S+1) of two consecutive contacts, written into
// lanes d = 30 (with A1) and d = 29 (with B1) of both halves of RJ[S+1].
// SH: the coupling with the first slot's lower-half row goes to lane d = 31 - SH, with its upper-half row to d = 30 - SH
template <int SH = 1>
__device__ __forceinline__ void cone2_coupling(const float RM1, float& RJ2, int lane) {
    const swap2 sw = half_swap(RM1, RM1);                 // a = [RM_A1, RM_A1], b = [RM_B1, RM_B1]
    const float tA = half_reduce(RJ2 * sw.a);             // lane 31: c(A2,A1), lane 63: c(B2,A1)
    const float tB = half_reduce(RJ2 * sw.b);             // lane 31: c(A2,B1), lane 63: c(B2,B1)
    const float mA = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tA), 0x100 + SH, 0xf, 0xf, true));       // row_shl:SH
    const float mB = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tB), 0x100 + SH + 1, 0xf, 0xf, true));   // row_shl:SH+1
    const int d = lane & 31;
    RJ2 = d == 31 - SH ? mA : (d == 30 - SH ? mB : RJ2);
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
    unsigned long long sv;      // the caller's exec mask (the one-lane update runs under its own)
    if (CLAMP) {
        asm volatile(
            "v_sub_f32 %[u], %[TARGV], %[dv]\n\t"
            "v_add_f32 %[xs], %[ACCV], %[u]\n\t"
            "v_med3_f32 %[xs], %[xs], -%[PMIV], %[PMIV]\n\t"
            "v_sub_f32 %[u], %[xs], %[ACCV]\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %[s], %[u], %[LN]\n\t"
            "s_mov_b64 %[sv], exec\n\t"
            "s_mov_b64 exec, %[MASK]\n\t"
            "v_add_f32 %[ACCV], %[ACCV], %[u]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "v_fmac_f32 %[dv], %[s], %[RMj]\n\t"
            : [u] "=&v"(u), [xs] "=&v"(xs), [s] "=&s"(s), [sv] "=&s"(sv), [ACCV] "+v"(ACCV), [dv] "+v"(dv)
            : [RMj] "v"(RMj), [TARGV] "v"(TARGV), [PMIV] "v"(PMIV), [LN] "n"(6 + J), [MASK] "n"(1 << (6 + J)));
    } else {
        asm volatile(
            "v_sub_f32 %[u], %[TARGV], %[dv]\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 %[s], %[u], %[LN]\n\t"
            "s_mov_b64 %[sv], exec\n\t"
            "s_mov_b64 exec, %[MASK]\n\t"
            "v_add_f32 %[ACCV], %[ACCV], %[u]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "v_fmac_f32 %[dv], %[s], %[RMj]\n\t"
            : [u] "=&v"(u), [s] "=&s"(s), [sv] "=&s"(sv), [ACCV] "+v"(ACCV), [dv] "+v"(dv)
            : [RMj] "v"(RMj), [TARGV] "v"(TARGV), [LN] "n"(6 + J), [MASK] "n"(1 << (6 + J)));
        (void)xs;
    }
    return s;
}


// eight consecutive contact normals (slots BASE..BASE+3), four per step; the second step only when the group holds
// more than four contacts (rows past the active count are inert zeros, but a step on them costs what any step does:
// with 35 contacts on average, groups of four instead of eight save 5 % of an iteration)
template <bool RES, int BASE>
__device__ __forceinline__ void duos4(float (&RJ)[kSlots], float (&RM)[kSlots], float& dv, float E3163, float& lsq, int ncl) {
    quad_step<RES>(RJ[BASE], RM[BASE], RJ[BASE + 1], RM[BASE + 1], dv, E3163, kLowMask, lsq);
    if (ncl > 2 * (BASE - kSlotNormal) + 4)
        quad_step<RES>(RJ[BASE + 2], RM[BASE + 2], RJ[BASE + 3], RM[BASE + 3], dv, E3163, kLowMask, lsq);
}

// eight consecutive friction pairs (contacts 8G..8G+7), two contacts per step, the second half of the group only when it
// holds a contact; the normal impulses of contacts c0, c0+1 sit in slot kSlotNormal + c0/2, lanes 31 and 63
template <bool RES, int G>
__device__ __forceinline__ void cones8(float (&RJ)[kSlots], float (&RM)[kSlots], float& dv, float EPS,
                                       float E3163, float& lsq, int ncl) {
#define SNK_CONE2(I)                                                                                                  \
    cone2_step<RES>(RJ[kSlotFric + 8 * G + 2 * (I)], RM[kSlotFric + 8 * G + 2 * (I)], RJ[kSlotFric + 8 * G + 2 * (I) + 1],   \
                    RM[kSlotFric + 8 * G + 2 * (I) + 1], RJ[kSlotNormal + 4 * G + (I)], dv, EPS, E3163, kLowMask, lsq);
    // (a branch per step instead of per two steps: no gain on the default model, -1.3 % on the round-1 model's 64 contacts)
    SNK_CONE2(0)
    SNK_CONE2(1)
    if (ncl > 8 * G + 4) {
        SNK_CONE2(2)
        SNK_CONE2(3)
    }
#undef SNK_CONE2
}

// the 16 motor rows, forwards or backwards; returns max |dI * den| = max |y| of the sweep
// (ALIVE: every motor row has a usable denominator -- always, unless a joint has no inertia;
// otherwise dead rows are masked out of the residual through LDS)
template <bool FWD, bool CLAMP, bool ALIVE, class LT>
__device__ __forceinline__ float motors16(LT& L, const float (&RMm)[16], float& dv, const float TARGV, float& ACCV,
                                          float PMIV) {
    float res = 0.f;
#define SNK_MOTOR(J)                                                                              \
    {                                                                                              \
        float s_ = motor_step<J, CLAMP>(RMm[J], dv, TARGV, ACCV, PMIV);                            \
        res = fmaxf(res, fabsf(ALIVE ? s_ : s_ * (L.MmS[J][1] * L.MmS[J][2])));                    \
    }
    if (FWD) {
        SNK_MOTOR(0) SNK_MOTOR(1) SNK_MOTOR(2) SNK_MOTOR(3) SNK_MOTOR(4) SNK_MOTOR(5) SNK_MOTOR(6) SNK_MOTOR(7)
        SNK_MOTOR(8) SNK_MOTOR(9) SNK_MOTOR(10) SNK_MOTOR(11) SNK_MOTOR(12) SNK_MOTOR(13) SNK_MOTOR(14) SNK_MOTOR(15)
    } else {
        SNK_MOTOR(15) SNK_MOTOR(14) SNK_MOTOR(13) SNK_MOTOR(12) SNK_MOTOR(11) SNK_MOTOR(10) SNK_MOTOR(9) SNK_MOTOR(8)
        SNK_MOTOR(7) SNK_MOTOR(6) SNK_MOTOR(5) SNK_MOTOR(4) SNK_MOTOR(3) SNK_MOTOR(2) SNK_MOTOR(1) SNK_MOTOR(0)
    }
#undef SNK_MOTOR
    return res;
}

// all contact rows of one iteration: normals in contact order (one scalar branch per 8
// contacts; rows past the active count are inert zeros), then the friction pairs.
// RES: lsq collects max |dI * den| in lanes 24 / 56.
template <bool RES>
__device__ __forceinline__ void contact_rows(float (&RJ)[kSlots], float (&RM)[kSlots], float& dv, int ncl, bool cone,
                                             float EPS, float E3163, float& lsq) {
    if (ncl > 0) duos4<RES, kSlotNormal + 0>(RJ, RM, dv, E3163, lsq, ncl);
    if (ncl > 8) duos4<RES, kSlotNormal + 4>(RJ, RM, dv, E3163, lsq, ncl);
    if (ncl > 16) duos4<RES, kSlotNormal + 8>(RJ, RM, dv, E3163, lsq, ncl);
    if (ncl > 24) duos4<RES, kSlotNormal + 12>(RJ, RM, dv, E3163, lsq, ncl);
    if (ncl > 32) duos4<RES, kSlotNormal + 16>(RJ, RM, dv, E3163, lsq, ncl);
    if (ncl > 40) duos4<RES, kSlotNormal + 20>(RJ, RM, dv, E3163, lsq, ncl);
    if (ncl > 48) duos4<RES, kSlotNormal + 24>(RJ, RM, dv, E3163, lsq, ncl);
    if (ncl > 56) duos4<RES, kSlotNormal + 28>(RJ, RM, dv, E3163, lsq, ncl);
    if (cone) {
        if (ncl > 0) cones8<RES, 0>(RJ, RM, dv, EPS, E3163, lsq, ncl);
        if (ncl > 8) cones8<RES, 1>(RJ, RM, dv, EPS, E3163, lsq, ncl);
        if (ncl > 16) cones8<RES, 2>(RJ, RM, dv, EPS, E3163, lsq, ncl);
        if (ncl > 24) cones8<RES, 3>(RJ, RM, dv, EPS, E3163, lsq, ncl);
        if (ncl > 32) cones8<RES, 4>(RJ, RM, dv, EPS, E3163, lsq, ncl);
        if (ncl > 40) cones8<RES, 5>(RJ, RM, dv, EPS, E3163, lsq, ncl);
        if (ncl > 48) cones8<RES, 6>(RJ, RM, dv, EPS, E3163, lsq, ncl);
        if (ncl > 56) cones8<RES, 7>(RJ, RM, dv, EPS, E3163, lsq, ncl);
    } else {
        // pyramid friction (not Bullet's default here): the two directions are resolved one
        // after the other (a duo with its coupling scalar), bounds +-lambda_n (rows are in
        // units of mu), and -- as in Bullet -- skipped altogether while the normal impulse is zero
#pragma unroll
        for (int g = 0; g < 8; g++) {
            if (ncl > 8 * g) {
#pragma unroll
                for (int ci = 8 * g; ci < 8 * g + 8; ci++) {
                    float lim = rdlane(RJ[kSlotNormal + (ci >> 1)], (ci & 1) ? 63 : 31);
                    if (__builtin_amdgcn_readfirstlane(lim > 0.f ? 1 : 0))
                        duo_step<RES, true>(RJ[kSlotFric + ci], RM[kSlotFric + ci], dv, lim, E3163, kLowMask, lsq);
                }
            }
        }
    }
}

// (6) constraint pass for the joint-0 sensor [U], as a FUNCTION OF ITS OWN (never inlined).  The pass is full of
// lane-dependent regions (lane < nc, lane <= N, lane == 0); inlined into the substep, every value the substep keeps in a
// vector register across it -- delta-v above all -- is exposed to the register allocator placing a copy or a reload of
// it INSIDE one of those regions, which then moves the active lanes only.  Round 4 met a build with that signature
// (substep_kernel<16, true>: joints 10..15 stood still under the single-substep API) and parked delta-v in LDS; ADVICE r4's
// point was that this protects one variable.  Behind a call boundary the whole class is closed: whatever the caller has
// live is preserved either by the callee's prologue / epilogue or by the caller around the call -- in both cases under
// the call site's exec mask (every lane), never inside the pass.  The pass takes wave-uniform scalars only, re-derives
// its lane index, and leaves its two results in the record (lane 0): no all-lane value crosses it in either direction.
// (The one build of that time that can be rebuilt from history -- commit 89723cb without the LDS detour -- was
// disassembled in round 5: delta-v crosses the pass untouched in v203, and that library passes
// test_substep_api_servo_converges on the GPU; the failing object was an uncommitted intermediate.  The cause above is
// therefore the best-supported reading, not a proven one: DESIGN.md 4.)
template <class LT>
__device__ __noinline__ void sensor_pass_v2(LT& L, const DevModel& M, const int nc, const int nlim
#ifdef SNK_PROFILE
                                            , unsigned long long* prof_t
#endif
) {
    constexpr int N = LT::kN;
    const int lane = lane_id();
    // lane = contact: its force and its moment about the body's joint origin (staging rows are
    // free now); then lane = body sums the contacts of its cylinders (2b-1 and 2b; body 0: cylinder 0) in
    // contact order
    if (lane < nc) {
        const int ci = lane, k = L.ccbody[ci];
        const f3 nrm = ci < L.nplane ? mk3(0.f, 0.f, 1.f) : ld3(L.obn[ci - L.nplane]);
        f3 F = (nrm * L.app[kAppNormal + ci] +
                ld3(L.cdir[L.ccds[ci]][0]) * L.app[kAppFric + 2 * ci] +
                ld3(L.cdir[L.ccds[ci]][1]) * L.app[kAppFric + 2 * ci + 1]) * M.inv_dt;
        st3(&L.stM[ci][0], cross(ld3(L.ccP[ci]) - ld3(L.o[k]), F));
        st3(&L.stM[ci][3], F);
    }
    lds_sync();
    if (lane <= N) {
        const int b = lane;
        f3 eN = mk3(0, 0, 0), eF = mk3(0, 0, 0);
        const int c0 = b == 0 ? 0 : 2 * b - 1;
        const int ncyl = (b == 0 || b == N) ? 1 : 2;
        for (int cc = c0; cc < c0 + ncyl; cc++) {
            const int cb = L.cylbase[cc], cn = L.cyln[cc];
            for (int ci = cb; ci < cb + cn; ci++) {
                eN = eN + ld3(&L.stM[ci][0]);
                eF = eF + ld3(&L.stM[ci][3]);
            }
        }
        for (int ci = L.nplane; ci < nc; ci++)               // the obstacle's contacts, behind the ground's
            if (L.ccbody[ci] == b) {
                eN = eN + ld3(&L.stM[ci][0]);
                eF = eF + ld3(&L.stM[ci][3]);
            }
        st3(L.ext(b), eN);
        st3(L.ext(b) + 3, eF);
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
    SNK_STAMP(14)
    body_bias<LT, false>(L, M, lane);
    lds_sync();
    aba_main<LT, false>(L, M, lane);
    {
        f3 zb = mulRv(L.R[0], ld3(M.zbase));
        f3 v1 = ld3(L.base() + 10);
        float nv1 = sqrtf(dot(v1, v1));
        f3 a2 = ld3(&L.acc0[3]);
        const float fz = L.fz_park - dot(zb, a2 * M.m_root + v1 * (M.m_root * (M.lin_damp + M.lin_damp * nv1)));
        const f3 al2 = ld3(&L.acc0[0]);
        const f3 F3 = -((a2 + cross(al2, ld3(L.cw[0]))) * M.mass[0] + ld3(&L.p[0][3]));
        const float fz3 = L.fz3_park + F3.x * L.R[1][2] + F3.y * L.R[1][5] + F3.z * L.R[1][8];
        // both results straight into the record (the caller's lds_sync in front of the integration follows)
        if (lane == 0) { L.fz() = fz; L.fz3() = fz3; }
    }
    SNK_STAMP(15)
}

template <class LT>
__device__ __forceinline__ void substep_v2(LT& L, const DevModel& M, int lane_in, float mu, int& iters, int& ncontacts,
                                           const SensorHint& hint, unsigned long long* __restrict__ ovf) {
    int lane = lane_in;
    constexpr int N = LT::kN;
    constexpr int ND = N + 6;
    static_assert(N == 16, "v2 is laid out for the 16-link chain");
    const float dt = M.dt;
#ifdef SNK_PROFILE
    unsigned long long prof_t[17];
#endif
    SNK_STAMP(0)
    // (1) contacts of the current pose, (2) bias forces with gravity, joint damping torque
    const int nc = __builtin_amdgcn_readfirstlane(find_contacts_v2(L, M, lane, ovf));
    ncontacts = nc;
    if (nc < 0) return;            // does not fit this solve: nothing has been touched (substep() takes the other one)
    SNK_STAMP(1)
    if (lane < N) {
        float qd = L.qd()[lane];
        L.qd_old[lane] = qd;
        L.tauj[lane] = -M.joint_damp * qd;   // PyBullet adds URDF joint damping as a torque [U]
    }
    body_bias<LT, true>(L, M, lane);
    lds_sync();
    aba_main<LT, true>(L, M, lane);
    SNK_STAMP(2)
    // joint-0 force sensor, first pass [U] (parked in LDS: nothing but rows may live across the solve)
    {
        f3 zb = mulRv(L.R[0], ld3(M.zbase));
        f3 v_old = ld3(L.base() + 10);
        float nv0 = sqrtf(dot(v_old, v_old));
        f3 a1 = ld3(&L.acc0[3]);
        float fz1 = -dot(zb, (a1 - mk3(0.f, 0.f, M.gz)) * M.m_root + v_old * (M.m_root * (M.lin_damp + M.lin_damp * nv0)));
        // ... and of the reaction through the first motor joint (Bullet joint 3, snake_gait_test.py:33-40): what body 0
        // does not use up of the forces on it, F = -(m_0 (a_0 + alpha_0 x c_0) + own force bias), z of body 1
        const f3 al0 = ld3(&L.acc0[0]);
        const f3 F3 = -((a1 + cross(al0, ld3(L.cw[0]))) * M.mass[0] + ld3(&L.p[0][3]));
        if (lane == 0) { L.fz_park = fz1; L.fz3_park = F3.x * L.R[1][2] + F3.y * L.R[1][5] + F3.z * L.R[1][8]; }
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

    SNK_STAMP(3)
    // (4) rows -> registers.  The lane=row builder needs ~60 registers of its own, so it only
    // runs while at most the 128 friction-row registers are live: motors go to LDS first,
    // then friction A, friction B, normals through the 64-row staging, motors are loaded last.
    float RJ[kSlots], RM[kSlots];
    float RMm[16], TARGV;
    bool malive;                  // every motor row has a positive denominator
    float wsum = 0.f;             // sum over the normal rows of M^-1 J^T x (the impulse the row starts from), per half
    {
        LaneK K;
        K.h = lane >> 5;
        K.d = lane & 31;
        K.isdof = K.d < ND;
        K.m22 = K.d == 22 ? -1.0f : 0.0f;
        K.m24 = K.d == 24 ? 1.0f : 0.0f;
        K.m31 = K.d == 31 ? 1.0f : 0.0f;
        K.spoff = K.d == 24 ? 1 : 0;
        K.bL = (K.d >= 6 && K.isdof) ? K.d - 5 : 0;
        K.oL = ld3(L.o[K.bL]);
        K.aL = (K.d >= 6 && K.isdof) ? ld3(L.ax[K.bL])
                                     : mk3(K.d == 0 ? 1.f : 0.f, K.d == 1 ? 1.f : 0.f, K.d == 2 ? 1.f : 0.f);
        MuScale ms;
        // (wave-uniform values, pinned to scalar registers: the row builder's vector registers are all spoken for)
        auto sgpr = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
        ms.fJ = sgpr(mu > 0.f ? 1.0f / mu : 0.f); ms.fM = sgpr(mu > 0.f ? mu : 0.f);
        const float mu_ob = fminf(M.mu_link * M.mu_obs, 10.0f);
        ms.fJo = sgpr(mu_ob > 0.f ? 1.0f / mu_ob : 0.f); ms.fMo = sgpr(mu_ob > 0.f ? mu_ob : 0.f);
        ms.nplane = __builtin_amdgcn_readfirstlane(L.nplane);
        build_batch_v2<LT, 0>(L, M, lane, nc);
        SNK_STAMP(4)
        // friction pairs: two batches of 32 contacts x {A, B}
#pragma unroll
        for (int s = 0; s < kSlots - kSlotFric; s++) { RJ[kSlotFric + s] = 0.f; RM[kSlotFric + s] = 0.f; }
        build_batch_v2<LT, 4>(L, M, lane, nc, 0);
        SNK_STAMP(5)
        if (nc > 0) load_slots8<LT, 4, 0, kSlotFric>(L, K, nc, 0, ms, RJ, RM, wsum);
        if (nc > 8) load_slots8<LT, 4, 8, kSlotFric>(L, K, nc, 0, ms, RJ, RM, wsum);
        if (nc > 16) load_slots8<LT, 4, 16, kSlotFric>(L, K, nc, 0, ms, RJ, RM, wsum);
        if (nc > 24) load_slots8<LT, 4, 24, kSlotFric>(L, K, nc, 0, ms, RJ, RM, wsum);
        lds_sync();
        SNK_STAMP(6)
        build_batch_v2<LT, 4>(L, M, lane, nc, 32);      // no active lanes when nc <= 32
        SNK_STAMP(7)
        if (nc > 32) load_slots8<LT, 4, 32, kSlotFric>(L, K, nc, 32, ms, RJ, RM, wsum);
        if (nc > 40) load_slots8<LT, 4, 40, kSlotFric>(L, K, nc, 32, ms, RJ, RM, wsum);
        if (nc > 48) load_slots8<LT, 4, 48, kSlotFric>(L, K, nc, 32, ms, RJ, RM, wsum);
        if (nc > 56) load_slots8<LT, 4, 56, kSlotFric>(L, K, nc, 32, ms, RJ, RM, wsum);
        lds_sync();
        SNK_STAMP(8)
        build_batch_v2<LT, 1>(L, M, lane, nc);
        SNK_STAMP(9)
#pragma unroll
        for (int s = 0; s < kSlotFric; s++) { RJ[kSlotNormal + s] = 0.f; RM[kSlotNormal + s] = 0.f; }
        // a normal slot holds contacts 2s, 2s+1: 8 slots per 16 contacts
        if (nc > 0) load_slots8<LT, 1, 0, kSlotNormal>(L, K, nc, 0, ms, RJ, RM, wsum);
        if (nc > 16) load_slots8<LT, 1, 8, kSlotNormal>(L, K, nc, 0, ms, RJ, RM, wsum);
        if (nc > 32) load_slots8<LT, 1, 16, kSlotNormal>(L, K, nc, 0, ms, RJ, RM, wsum);
        if (nc > 48) load_slots8<LT, 1, 24, kSlotNormal>(L, K, nc, 0, ms, RJ, RM, wsum);
        // motors: column 6+j of M^-1 (divided by the row's denominator) in both halves;
        // target velocity change of motor j in lane 6+j
#pragma unroll
        for (int j = 0; j < 16; j++) RMm[j] = K.isdof ? L.Mm[j][K.d] * L.MmS[j][2] : 0.f;
        {
            const bool mot = K.d >= 6 && K.isdof;
            const int jm = mot ? K.d - 6 : 0;
            const float dinv_m = mot ? L.MmS[jm][2] : 0.f;
            TARGV = (mot && dinv_m > 0.f) ? L.MmS[jm][3] : 0.f;
            malive = __all(!mot || dinv_m > 0.f) != 0;
        }
        lds_sync();
    }

    SNK_STAMP(10)
    // coupling scalar of each duo, once per substep:  lane 62 of RJ <- RJ_upper . RM_lower
    // (the upper row is resolved after the lower one).  Pyramid friction resolves the two
    // directions of a contact one after the other, so its slots are duos as well.
    {
#pragma unroll
        for (int s = 0; s < kSlotFric; s++) {
            swap2 sw = half_swap(RM[s], RM[s]);             // a = [RM_lower, RM_lower], b = [RM_upper, RM_upper]
            float t = half_reduce(RJ[s] * sw.a);
            RJ[s] = wrlane(RJ[s], rdlane(t, 63), 62);       // d = 30 of the upper half: duo_step's row_shr:1 source
        }
        // ... and the 2 x 2 block between the two slots of a quad_step (four normals per step): lanes d = 29, 28
#pragma unroll
        for (int g = 0; g < 8; g++) {
            if (nc > 8 * g) {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int s = kSlotNormal + 4 * g + 2 * i;
                    cone2_coupling<2>(RM[s], RJ[s + 1], lane);
                }
            }
        }
        if (M.cone == 0) {
#pragma unroll
            for (int s = kSlotFric; s < kSlots; s++) {
                swap2 sw = half_swap(RM[s], RM[s]);
                float t = half_reduce(RJ[s] * sw.a);
                RJ[s] = wrlane(RJ[s], rdlane(t, 63), 62);
            }
        } else {
            // cone friction resolves two contacts per step (cone2_step): the 2 x 2 coupling block of each pair of
            // contacts, groups of 8 contacts behind one scalar branch (rows past the active count are zero anyway)
#pragma unroll
            for (int g = 0; g < 8; g++) {
                if (nc > 8 * g) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int s = kSlotFric + 8 * g + 2 * i;
                        cone2_coupling(RM[s], RJ[s + 1], lane);
                    }
                }
            }
        }
    }

    SNK_STAMP(11)
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

    SNK_STAMP(12)
    // The solve runs at a higher wave priority than the phases around it.  Two waves share a SIMD; the one in its solve
    // issues VALU instructions back to back, the one in a setup phase waits on LDS most of the time -- when both have an
    // instruction ready, the solve's goes first (oldest-first arbitration gave the older WAVE that preference whatever
    // it was doing).  One s_setprio on either side of the loop: 352 k -> 367 k env-steps/s; the other way round
    // (setup first) 350 k.
    __builtin_amdgcn_s_setprio(3);
    // (5) projected Gauss-Seidel on the register-resident rows
    float dv;
    {
        const int d = lane & 31;
        dv = d == 22 ? 1.0f : (d == 31 ? -1.0f : 0.0f);
        // warm starting (snk_params::warm_start): every normal row starts at the impulse find_contacts_v2 left for its
        // contact (factor x what its cached point carried; zero without warm starting) -- finish_slot put it into lane 31
        // / 63 of the row's RJ -- and delta-v at the sum of M^-1 J^T of those (wsum, collected while the slots were
        // loaded).  No branch on the switch here: a conditional update of 32 row registers in front of the solve makes
        // the compiler keep two of them in scratch memory for the whole loop (round-3 measurement in the ISA).
        {
            const swap2 sw = half_swap(wsum, wsum);     // a = the lower half's rows' part in both halves, b = the upper's
            dv += sw.a + sw.b;
        }
        const float E3163 = d == 31 ? 1.0f : 0.0f;
        const float EPS = 1e-30f;
        const float mi = M.max_motor_imp;
        const bool mclamp = mi < 1e30f;
        float ACCV = 0.f;     // accumulated motor impulses, motor j in lane 6+j
        // (a wave-uniform value, pinned to a scalar register: the solve has no vector register to spare)
        const float thr = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sqrtf(M.resid_thr))));
        const int n_iter = M.n_iter;
        const bool cone = M.cone != 0;
        int it = 0;
        // Every row register gets a NEW live range here (an empty asm that "rewrites" it).  Where the row builder's peak
        // demand had pushed a row into scratch memory, the allocator otherwise keeps it there for good and reloads it at
        // every use -- inside this loop, 50 times per substep (round-3 ISA: up to 20 such reloads per iteration) -- although
        // the loop itself leaves twenty registers unused.  Split here, the row is reloaded once, in front of the loop.
#pragma unroll
        for (int s2 = 0; s2 < kSlots; s2++) asm volatile("" : "+v"(RJ[s2]), "+v"(RM[s2]));
#pragma unroll
        for (int j2 = 0; j2 < 16; j2++) asm volatile("" : "+v"(RMm[j2]));
        for (; it < n_iter; it++) {
            // Bullet leaves the sweep when max_rows |dI * den| <= threshold.  `exceeded` becomes 1
            // as soon as a group of rows shows a larger residual; from then on the residual-free
            // variants (RES = false) run.
            int exceeded = 0;
            int ncl = nc;
            asm volatile("" : "+s"(ncl));   // keeps the group-active compares from being hoisted and spilled
            // ... and once more at the top of every iteration (no instruction, only a point where every row must be in a
            // register): with the split in front of the loop alone two rows stayed in scratch memory and were reloaded
            // in every iteration, behind a full s_waitcnt each, although twenty registers are unused inside the loop
            // (tools/dbg/loop_spills.sh: 2 -> 0 scratch operations in the loop; 350.4 k -> 353.0 k env-steps/s).
#pragma unroll
            for (int s2 = 0; s2 < kSlots; s2++) asm volatile("" : "+v"(RJ[s2]), "+v"(RM[s2]));
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
            // non-contact rows: list = [limits..., motors 0..15], walked forwards on odd
            // iterations and backwards on even ones.  The motors' residual is always tracked
            // (one v_max per row); it decides which body runs for the ~190 contact rows.
            float mres;
            // the per-lane bound on y = dI * den of a clamped motor (max_motor_impulse finite: not the reference's
            // forces = inf), rebuilt from LDS in every sweep that needs it rather than held in a register for all
            auto pmiv = [&]() {
                const bool mot = d >= 6 && d < ND;
                const float dinv_m = mot ? L.MmS[mot ? d - 6 : 0][2] : 0.f;
                return (dinv_m > 0.f) ? mi / dinv_m : 1e30f;
            };
#define SNK_MOTORS(FWD)                                                                             \
    mres = malive ? (mclamp ? motors16<FWD, true, true>(L, RMm, dv, TARGV, ACCV, pmiv())             \
                            : motors16<FWD, false, true>(L, RMm, dv, TARGV, ACCV, 0.f))              \
                  : (mclamp ? motors16<FWD, true, false>(L, RMm, dv, TARGV, ACCV, pmiv())            \
                            : motors16<FWD, false, false>(L, RMm, dv, TARGV, ACCV, 0.f));
            if (it & 1) {
                limit_rows(true);
                SNK_MOTORS(true)
            } else {
                SNK_MOTORS(false)
                limit_rows(false);
            }
#undef SNK_MOTORS
            if (__builtin_amdgcn_readfirstlane(mres > thr ? 1 : 0)) exceeded = 1;
            // contact rows: the residual is tracked in every step (one v_max3; a second,
            // residual-free copy of the ~190 row steps costs more in register copies at the
            // join and in instruction cache than it saves)
            {
                float l_ = 0.f;
                contact_rows<true>(RJ, RM, dv, ncl, cone, EPS, E3163, l_);
                float m = fmaxf(rdlane(l_, 24), rdlane(l_, 56));
                if (__builtin_amdgcn_readfirstlane(m > thr ? 1 : 0)) exceeded = 1;
            }
            if (!exceeded || it >= n_iter - 1) { it++; break; }
        }
        iters = it;
        // The lane index is re-derived here (v_mbcnt) instead of being kept in a register across the solve: the row
        // registers leave none to spare (the callers do the same after the substep returns).
        lane = lane_id();
        // accumulated impulses -> LDS (friction back from units of mu)
        if ((lane & 31) == 31) {
            const int h = lane >> 5;
#pragma unroll
            for (int s = 0; s < kSlotFric; s++) L.app[kAppNormal + 2 * s + h] = RJ[kSlotNormal + s];
            const int npl = L.nplane;
            const float mu_ob = fminf(M.mu_link * M.mu_obs, 10.0f);
#pragma unroll
            for (int s = 0; s < kSlots - kSlotFric; s++) L.app[kAppFric + 2 * s + h] = RJ[kSlotFric + s] * (s < npl ? mu : mu_ob);
        }
        if (lane >= 6 && lane < ND) L.app[kAppMotor + lane - 6] = ACCV * L.MmS[lane - 6][2];   // y / den = impulse
    }
    lds_sync();
    if (M.contact_model == 1 && lane < 2 * N) {
        // the normal impulses go back into the contact cache (btManifoldPoint::m_appliedImpulse [U])
        int idx = L.cylbase[lane];
        const int mask = L.cylkeep[lane];
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((mask >> j) & 1) L.mfl[6 * j + 5][lane] = L.app[kAppNormal + idx++];
    }

    SNK_STAMP(13)
    __builtin_amdgcn_s_setprio(0);
    // (6) constraint pass for the joint-0 sensor [U] -- only when this substep can be the last of
    // its env-step (obs[55] is not observable otherwise): the servo error after it is within the
    // tolerance, or the counter reaches its cap, or the mean height can cross its threshold.  The
    // first two are evaluated exactly as the loop does (with a 1e-3 safety factor on the
    // tolerance); for the third, no sampled point can move further in one substep than
    // dt * (|v| + L_chain * (|omega| + sum |qd|)): rigid rotations about the base and the joints.
    const bool sensor = sensor_pass_needed(L, M, lane, dv, hint);
    // delta-v crosses the sensor pass in LDS (a free column of the staging rows), not in a register -- round 4's first
    // answer to a build that lost its lanes 16..21 on the way; round 5 put the pass itself behind a call boundary
    // (sensor_pass_v2 above), which protects every such value.  The detour stays: it costs one LDS write and read per
    // substep and takes a register out of the callee-saved set around the call.
    L.stM[lane][20] = dv;
    if (sensor) {
#ifdef SNK_PROFILE
        sensor_pass_v2(L, M, nc, nlim, prof_t);
#else
        sensor_pass_v2(L, M, nc, nlim);
#endif
    }
    // (7) apply the solver's delta-v (lower half's copy), motor torques, integrate positions
    lds_sync();
    dv = L.stM[lane][20];
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
        }
    }
    lds_sync();
    fk_vel(L, M, lane);
#ifdef SNK_PROFILE
    SNK_STAMP(16)
    lds_sync();
    if (lane < 16) L.taum()[lane] = (float)(prof_t[lane + 1] - prof_t[lane]);
    lds_sync();
#endif
}

}  // namespace snk
