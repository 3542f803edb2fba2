// snk_selfcol.hpp -- link-link (self) collision of the snake's cylinders: what URDF_USE_SELF_COLLISION
// (/root/reference/snake.py:93) switches on in PyBullet [U].  SURVEY.md 8(f)-2.
//
// Every pair of cylinder links except direct parent-child pairs (consecutive cylinders of the chain; the flag's
// default excludes a link's parent) is tested each substep:
//   broad phase   lane = cylinder a, one pass per offset b - a = 2, 3, ...: bounding spheres
//   narrow phase  distance between the two convex CORE shapes (the 32-gon hull PyBullet imports a URDF <cylinder> as,
//                 or the implicit cylinder; without margin) by GJK, as btGjkPairDetector does; the collision margins
//                 come off the distance and the witness points move onto the inflated surfaces.  Cores that overlap
//                 (more than both margins deep, where Bullet switches to EPA): a second GJK on cores shrunk by
//                 kShrink with the margin enlarged by as much; if even those overlap, the line of centres.
// One point per pair per step (stateless; Bullet caches up to four per pair), kept when closer than the breaking
// threshold, at most kMaxSelf per environment (what does not fit is counted: snk_contact_overflow), ordered by (b - a, a) -- the same rules as oracle/snake_oracle.cpp
// (find_self_contacts), restated independently here in float32.
//
// Only the streamed-row solve builds rows for these contacts.  The register-resident 16-link solve has no two-body rows:
// it runs the broad phase and the separating-axis cull (any_self_pair_v2 below), and a substep in which some pair of links
// may be within the breaking threshold goes through the streamed-row solve instead -- never under the reference's
// command range (|target| <= 30 deg: 15 mm of clearance, tools/self_collision_clearance.py), but a snake bent further
// (a larger scaling_factor, states set from outside) does fold onto itself.
#pragma once

namespace snk {

constexpr float kShrink = 0.006f;

struct Cvx {
    f3 c;           // world centre of the shape
    float R[9];     // world rotation of its frame
    int box;        // 0: one of the snake's cylinders; 1: the obstacle box (half extents below)
    f3 half;
};

// support point of the core shape (radius and half length reduced by `shrink`) in world direction dw
__device__ __forceinline__ f3 support_core(const DevModel& M, const Cvx& s, f3 dw, float shrink) {
    const f3 dl = mulRtv(s.R, dw);
    const float rad = M.cyl_r - shrink, hl = M.cyl_hl - shrink;
    f3 v;
    if (s.box) {          // btBoxShape keeps its collision margin INSIDE the nominal box: the core is smaller by it
        const f3 h = mk3(s.half.x - M.margin - shrink, s.half.y - M.margin - shrink, s.half.z - M.margin - shrink);
        v = mk3(dl.x < 0.f ? -h.x : h.x, dl.y < 0.f ? -h.y : h.y, dl.z < 0.f ? -h.z : h.z);
    } else if (M.hull_sides > 0) {
        const float sc = rad / M.cyl_r;
        float best = -3.0e38f;
        v = mk3(0.f, 0.f, 0.f);
        for (int k = 0; k < 2 * M.hull_sides; k++) {
            const f3 c = mk3(M.hull_xy[k >> 1][0] * sc, M.hull_xy[k >> 1][1] * sc, (k & 1) ? -hl : hl);
            const float val = dot(dl, c);
            if (val > best) { best = val; v = c; }
        }
    } else {
        const float rr = sqrtf(dl.x * dl.x + dl.y * dl.y);
        v = rr != 0.f ? mk3(rad * dl.x / rr, rad * dl.y / rr, 0.f) : mk3(rad, 0.f, 0.f);
        v.z = dl.z < 0.f ? -hl : hl;
    }
    return s.c + mulRv(s.R, v);
}

// The simplex of GJK: up to four vertices of the Minkowski difference (W = A - B) with their witness points and
// barycentric weights.  Every index below is a compile-time constant (unrolled selects instead of dynamic array
// indexing), so the whole structure stays in registers: the first version indexed these arrays dynamically, the
// compiler put them in scratch memory, and the ~100 dependent scratch round trips per iteration cost 0.4 ms per
// physics substep (configs[3]: 27.2 k -> 31 k env-steps/s with this form).
struct Simplex {
    int n;
    f3 W[4], A[4], B[4];
    float lam[4];
};
struct Vtx {
    f3 w, a, b;
};
__device__ __forceinline__ Vtx sx_get(const Simplex& S, int i) {      // i: runtime, 0..3
    Vtx v;
    v.w = S.W[0]; v.a = S.A[0]; v.b = S.B[0];
    if (i == 1) { v.w = S.W[1]; v.a = S.A[1]; v.b = S.B[1]; }
    if (i == 2) { v.w = S.W[2]; v.a = S.A[2]; v.b = S.B[2]; }
    if (i == 3) { v.w = S.W[3]; v.a = S.A[3]; v.b = S.B[3]; }
    return v;
}
__device__ __forceinline__ void sx_put(Simplex& S, int k, const Vtx& v, float lam) {   // k: runtime, 0..3
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i == k) { S.W[i] = v.w; S.A[i] = v.a; S.B[i] = v.b; S.lam[i] = lam; }
}

__device__ __forceinline__ void closest_segment(f3 a, f3 b, float* l2) {
    const f3 ab = b - a;
    const float t = -dot(a, ab), dd = dot(ab, ab);
    if (t <= 0.f || dd <= 0.f) { l2[0] = 1.f; l2[1] = 0.f; }
    else if (t >= dd) { l2[0] = 0.f; l2[1] = 1.f; }
    else { l2[1] = t / dd; l2[0] = 1.f - l2[1]; }
}
// closest point of triangle abc to the origin, barycentric (Ericson, Real-Time Collision Detection 5.1.5)
__device__ __forceinline__ void closest_triangle(f3 a, f3 b, f3 c, float& l0, float& l1, float& l2) {
    const f3 ab = b - a, ac = c - a, ap = -a, bp = -b, cp = -c;
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    const float d5 = dot(ab, cp), d6 = dot(ac, cp);
    const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d1 <= 0.f && d2 <= 0.f) { l0 = 1.f; l1 = 0.f; l2 = 0.f; }
    else if (d3 >= 0.f && d4 <= d3) { l0 = 0.f; l1 = 1.f; l2 = 0.f; }
    else if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { const float v = d1 / (d1 - d3); l0 = 1.f - v; l1 = v; l2 = 0.f; }
    else if (d6 >= 0.f && d5 <= d6) { l0 = 0.f; l1 = 0.f; l2 = 1.f; }
    else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { const float w = d2 / (d2 - d6); l0 = 1.f - w; l1 = 0.f; l2 = w; }
    else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        l0 = 0.f; l1 = 1.f - w; l2 = w;
    } else {
        const float den = 1.0f / (va + vb + vc);
        const float v = vb * den, w = vc * den;
        l0 = 1.f - v - w; l1 = v; l2 = w;
    }
}

// keeps, in order, the vertices v0, v1, v2 (m of them) that carry positive weight, in slots 0 ...
__device__ __forceinline__ void simplex_keep(Simplex& S, const Vtx& v0, const Vtx& v1, const Vtx& v2, float l0, float l1,
                                             float l2, int m) {
    int k = 0;
    if (l0 > 0.f) { sx_put(S, k, v0, l0); k++; }
    if (m > 1 && l1 > 0.f) { sx_put(S, k, v1, l1); k++; }
    if (m > 2 && l2 > 0.f) { sx_put(S, k, v2, l2); k++; }
    S.n = k;
}

// closest point v of the simplex to the origin, the simplex reduced to the supporting sub-simplex; false when the
// origin lies inside the tetrahedron
__device__ __forceinline__ bool simplex_closest(Simplex& S, f3& v) {
    const Vtx p0 = sx_get(S, 0), p1 = sx_get(S, 1), p2 = sx_get(S, 2), p3 = sx_get(S, 3);
    if (S.n == 1) {
        S.lam[0] = 1.f;
    } else if (S.n == 2) {
        float l[2];
        closest_segment(p0.w, p1.w, l);
        simplex_keep(S, p0, p1, p2, l[0], l[1], 0.f, 2);
    } else if (S.n == 3) {
        float l0, l1, l2;
        closest_triangle(p0.w, p1.w, p2.w, l0, l1, l2);
        simplex_keep(S, p0, p1, p2, l0, l1, l2, 3);
    } else {
        // tetrahedron: inside, or the closest of the faces the origin is outside of; faces (0,1,2|3) (0,2,3|1)
        // (0,3,1|2) (1,3,2|0)
        float best = 3.0e38f, bl0 = 0.f, bl1 = 0.f, bl2 = 0.f;
        Vtx f0 = p0, f1 = p1, f2 = p2;
        bool outside_any = false;
        auto face = [&](const Vtx& a, const Vtx& b, const Vtx& c, const Vtx& d) {
            const f3 nn = cross(b.w - a.w, c.w - a.w);
            const float sp = -dot(a.w, nn), sd = dot(d.w - a.w, nn);
            if (sp * sd >= 0.f && sd != 0.f) return;
            outside_any = true;
            float l0, l1, l2;
            closest_triangle(a.w, b.w, c.w, l0, l1, l2);
            const f3 q = a.w * l0 + b.w * l1 + c.w * l2;
            const float dd = dot(q, q);
            if (dd < best) { best = dd; bl0 = l0; bl1 = l1; bl2 = l2; f0 = a; f1 = b; f2 = c; }
        };
        face(p0, p1, p2, p3);
        face(p0, p2, p3, p1);
        face(p0, p3, p1, p2);
        face(p1, p3, p2, p0);
        if (!outside_any) { v = mk3(0.f, 0.f, 0.f); return false; }
        simplex_keep(S, f0, f1, f2, bl0, bl1, bl2, 3);
    }
    v = mk3(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < S.n) v = v + S.W[i] * S.lam[i];
    return true;
}

// distance between the (shrunk) cores, witness points pa / pb; -1 when they overlap
__device__ __noinline__ float gjk_distance(const DevModel& M, const Cvx& a, const Cvx& b, float shrink, f3& pa, f3& pb) {
    Simplex S;
    S.n = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { S.W[i] = mk3(0, 0, 0); S.A[i] = mk3(0, 0, 0); S.B[i] = mk3(0, 0, 0); S.lam[i] = 0.f; }
    f3 v = a.c - b.c;
    if (dot(v, v) == 0.f) v.x = 1.f;
    // relative duality gap on the squared distance: 1e-5 leaves the distance of a 15-mm gap good to 1e-7 m; smooth
    // (implicit) cylinders converge linearly, hulls terminate on a vertex after a handful of steps
    const float eps = 1e-5f;
    float result = 0.f;
    bool overlap = false;
    for (int it = 0; it < 32; it++) {
        Vtx nw;
        nw.a = support_core(M, a, -v, shrink);
        nw.b = support_core(M, b, v, shrink);
        nw.w = nw.a - nw.b;
        const float vv = dot(v, v), vw = dot(v, nw.w);
        if (S.n > 0 && vv - vw <= eps * vv) break;
        bool dup = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const f3 d = S.W[i] - nw.w;
            if (i < S.n && dot(d, d) <= 1e-12f) dup = true;
        }
        if (dup) break;
        sx_put(S, S.n, nw, 0.f);
        S.n++;
        if (!simplex_closest(S, v)) { overlap = true; break; }
        if (dot(v, v) <= 1e-12f) { overlap = true; break; }
    }
    if (overlap) return -1.0f;
    pa = mk3(0.f, 0.f, 0.f);
    pb = mk3(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < S.n) { pa = pa + S.A[i] * S.lam[i]; pb = pb + S.B[i] * S.lam[i]; }
    result = sqrtf(dot(v, v));
    return result;
}

// btPlaneSpace1 [U]: two unit directions orthogonal to n
__device__ __forceinline__ void plane_space(f3 n, f3& p, f3& q) {
    if (fabsf(n.z) > 0.7071067811865475244f) {
        const float a = n.y * n.y + n.z * n.z, k = 1.0f / sqrtf(a);
        p = mk3(0.f, -n.z * k, n.y * k);
        q = mk3(a * k, -n.x * p.z, n.x * p.y);
    } else {
        const float a = n.x * n.x + n.y * n.y, k = 1.0f / sqrtf(a);
        p = mk3(-n.y * k, n.x * k, 0.f);
        q = mk3(-n.z * p.y, n.z * p.x, a * k);
    }
}
__device__ __forceinline__ f3 aniso_scale(const DevModel& M, const float* Rw, f3 d) {
    const f3 l = mulRtv(Rw, d);
    return mulRv(Rw, mk3(l.x * M.aniso[0], l.y * M.aniso[1], l.z * M.aniso[2]));
}

// Separating-axis culls in front of the GJK (round 3).  A lower bound on the distance between the NOMINAL shapes along one
// axis; a pair goes to the narrow phase only when that bound, less both margins, is within the breaking threshold -- the
// narrow phase would reject every pair culled here (dist = GJK distance - 2 margins >= bound - 2 margins > threshold), so
// the contact set is the oracle's, which has no such cull.  A 32-gon hull lies inside its cylinder.  (Under the relative
// breaking threshold, 1.2 mm, the neighbours across one joint -- 15 to 29 mm apart -- never pass; their bounding spheres
// always overlap, and 62 of them went through the GJK every substep: 12 % of the 32-link kernel.)
__device__ __forceinline__ float cyl_extent(const DevModel& M, f3 axis, f3 u) {      // |u| = 1
    const float c = dot(axis, u);
    return M.cyl_hl * fabsf(c) + M.cyl_r * sqrtf(fmaxf(0.f, 1.0f - c * c));
}
__device__ __forceinline__ bool cyl_cyl_may_touch(const DevModel& M, f3 ca, f3 axa, f3 cb, f3 axb) {
    const f3 d = ca - cb;
    const float n2 = dot(d, d);
    if (!(n2 > 1e-12f)) return true;
    const float n = sqrtf(n2);
    const f3 u = d * (1.0f / n);
    const float bound = n - cyl_extent(M, axa, u) - cyl_extent(M, axb, u);
    return bound - 2.0f * M.margin <= M.break_thr + 1e-5f;
}
// cylinder against the box (centre bc, rotation Rb row-major, half extents hb): the three face normals of the box
__device__ __forceinline__ bool cyl_box_may_touch(const DevModel& M, f3 ca, f3 axa, f3 bc, const float* Rb, f3 hb) {
    const f3 d = ca - bc;
    float bound = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const f3 e = mk3(Rb[i], Rb[3 + i], Rb[6 + i]);          // column i: the box's axis i in the world
        const float hi = i == 0 ? hb.x : (i == 1 ? hb.y : hb.z);
        bound = fmaxf(bound, fabsf(dot(d, e)) - (hi - M.margin) - cyl_extent(M, axa, e));
    }
    return bound - 2.0f * M.margin <= M.break_thr + 1e-5f;
}

// Does any pair of cylinder links (parent-child pairs excepted) pass the broad phase and the separating-axis cull?
// Wave-uniform answer.  Centres and axes are staged in LDS (the row staging area, idle at this point of a substep);
// lane = (half, cylinder a): the lower half takes the even offsets b - a, the upper half the odd ones, no cross-lane
// traffic and one ballot at the end (the first version walked the offsets with three ds_bpermutes and a ballot each:
// 1.1 % of the kernel in dependent latency).
template <class LT>
__device__ __forceinline__ bool any_self_pair_v2(LT& L, const DevModel& M, int lane) {
    constexpr int NCYL = 2 * LT::kN;
    static_assert(NCYL == 32, "one cylinder per lane of a half");
    const float rb = sqrtf(M.cyl_r * M.cyl_r + M.cyl_hl * M.cyl_hl) + M.margin;
    const float reach = 2.0f * rb + M.break_thr;
    const int a = lane & 31, h = lane >> 5;
    const int ba = (a + 1) >> 1;
    const f3 ca = ld3(L.o[ba]) + mulRv(L.R[ba], ld3(M.cyl_c[a]));
    const f3 axa = mulRv(L.R[ba], mk3(M.cyl_R[a][2], M.cyl_R[a][5], M.cyl_R[a][8]));
    if (h == 0) {
        st3(&L.stM[a][0], ca);
        st3(&L.stM[a][3], axa);
    }
    lds_sync();
    bool hit = false;
#pragma unroll 5
    for (int t = 0; t < 15; t++) {
        const int bc = a + 2 + h + 2 * t;
        const bool valid = bc < NCYL;
        const int bb = valid ? bc : a;
        const f3 cb = ld3(&L.stM[bb][0]);
        const f3 d = ca - cb;
        bool cand = valid && dot(d, d) <= reach * reach;
        if (cand) cand = cyl_cyl_may_touch(M, ca, axa, cb, ld3(&L.stM[bb][3]));
        hit = hit || cand;
    }
    const bool any = __any(hit) != 0;
    lds_sync();                 // (the staging area is written again further down)
    return any;
}

// Link-link contacts of the current pose, appended behind the ground contacts: geometry records at slots
// NC .. NC + count - 1 of the environment's global block.  Returns their number (<= kMaxSelf).
template <class LT>
__device__ int find_self_contacts_v1(LT& L, const DevModel& M, int lane, float mu_ground, float* __restrict__ rows,
                                     unsigned long long* __restrict__ ovf) {
    constexpr int N = LT::kN;
    constexpr int NCYL = 2 * N;
    const float rb = sqrtf(M.cyl_r * M.cyl_r + M.cyl_hl * M.cyl_hl) + M.margin;
    const float reach = 2.0f * rb + M.break_thr;
    const float mu_self = fminf(M.mu_link * M.mu_link, 10.0f);
    // the solve bounds every friction pair by mu_ground * lambda_n; rows of a link-link contact are written in
    // units that make that the right bound for its own coefficient (build_rows_v1 scales by geo[18])
    const float rho = mu_ground > 0.f ? mu_self / mu_ground : 0.f;
    auto frame = [&](int c, Cvx& s) {
        const int b = (c + 1) >> 1;
        const float* Rb = L.R[b];
        const float* Rc = M.cyl_R[c];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                s.R[3 * i + j] = Rb[3 * i] * Rc[j] + Rb[3 * i + 1] * Rc[3 + j] + Rb[3 * i + 2] * Rc[6 + j];
        s.c = ld3(L.o[b]) + mulRv(Rb, ld3(M.cyl_c[c]));
        s.box = 0;
        s.half = mk3(0.f, 0.f, 0.f);
    };
    // lane = cylinder a; pass = offset delta: the pairs (a, a + delta).  The other cylinder's centre comes from lane
    // a + delta through the LDS crossbar (ds_bpermute: no LDS memory).  All the neighbours-across-one-joint
    // (delta = 2, always inside each other's bounding spheres) share one pass, so their GJK runs lane-parallel.
    static_assert(NCYL <= 64, "one cylinder per lane");
    const int a = lane < NCYL ? lane : NCYL - 1;
    const int ba = (a + 1) >> 1;
    const f3 ca = ld3(L.o[ba]) + mulRv(L.R[ba], ld3(M.cyl_c[a]));
    const f3 axa = mulRv(L.R[ba], mk3(M.cyl_R[a][2], M.cyl_R[a][5], M.cyl_R[a][8]));      // the cylinder's axis in the world
    // The obstacle box (static): lane = cylinder, one point per (cylinder, box) pair, normal from the box to the link,
    // friction mu_link x mu_obstacle, directions scaled by the link's anisotropy only.  Its records FOLLOW the link-link
    // ones (row order: ground, link-link, obstacle -- the oracle's), but the narrow phase runs first: when the room for
    // these contacts (kMaxSelf) runs out, the obstacle's are kept and link-link contacts go (in pair order, last first).
    bool ob_hit = false;
    f3 ob_P = mk3(0, 0, 0), ob_n = mk3(0, 0, 1);
    float ob_dist = 0.f;
    float ob_R[9];
#pragma unroll
    for (int i = 0; i < 9; i++) ob_R[i] = 0.f;
    int n_ob = 0;
    f3 ob_PB = mk3(0, 0, 0);           // the contact's point on the box (a free box, obstacle 2, needs it for its rows)
    if (M.obstacle) {
        Cvx Bx;
        Bx.c = mk3(M.obs_c[0], M.obs_c[1], M.obs_c[2]);
#pragma unroll
        for (int i = 0; i < 9; i++) Bx.R[i] = (i % 4 == 0) ? 1.f : 0.f;
        if (N <= 16 && M.obstacle == 2) {         // the free box: where it is now (LDS: Lds<N, false>::box, bR)
            Bx.c = ld3(L.box);
#pragma unroll
            for (int i = 0; i < 9; i++) Bx.R[i] = L.bR[i];
        }
        Bx.box = 1;
        Bx.half = mk3(M.obs_h[0], M.obs_h[1], M.obs_h[2]);
        const float rbox = sqrtf(dot(Bx.half, Bx.half));
        const f3 d = ca - Bx.c;
        const float reach_ob = rb + rbox + M.break_thr;
        const bool cand = lane < NCYL && dot(d, d) <= reach_ob * reach_ob && cyl_box_may_touch(M, ca, axa, Bx.c, Bx.R, Bx.half);
        if (__any(cand)) {
            if (cand) {
                Cvx A;
                frame(a, A);
#pragma unroll
                for (int i = 0; i < 9; i++) ob_R[i] = A.R[i];
                f3 pa, pb;
                float mg = M.margin;
                float dd = gjk_distance(M, A, Bx, 0.f, pa, pb);
                if (dd < 0.f) {
                    dd = gjk_distance(M, A, Bx, kShrink, pa, pb);
                    mg = M.margin + kShrink;
                }
                if (dd < 0.f) {
                    const float nn = sqrtf(dot(d, d));
                    ob_n = nn > 0.f ? d * (1.0f / nn) : mk3(-1.f, 0.f, 0.f);
                    ob_P = A.c;
                    ob_PB = ob_P;
                    ob_dist = -2.0f * mg;
                } else {
                    ob_n = (pa - pb) * (1.0f / dd);
                    ob_dist = dd - 2.0f * mg;
                    ob_P = pa - ob_n * mg;
                    ob_PB = pb + ob_n * mg;
                }
                ob_hit = ob_dist < M.break_thr;
            }
        }
    }
    const unsigned long long ob_bal = __ballot(ob_hit);
    n_ob = __popcll(ob_bal);
    const int ob_kept = n_ob < LT::kMaxSelf ? n_ob : LT::kMaxSelf;
    const int self_room = LT::kMaxSelf - ob_kept;
    int nself = 0;
    for (int delta = 2; delta < (M.self_collision ? NCYL : 0); delta++) {
        const int bc = a + delta;
        const bool valid = lane < NCYL && bc < NCYL;
        const int src = (bc < NCYL ? bc : NCYL - 1) << 2;
        const f3 cb = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(ca.x))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(ca.y))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(ca.z))));
        bool cand = false;
        if (valid) {
            const f3 d = ca - cb;
            cand = dot(d, d) <= reach * reach;
        }
        if (!__any(cand)) continue;
        {
            const f3 axb = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(axa.x))),
                               __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(axa.y))),
                               __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(axa.z))));
            cand = cand && cyl_cyl_may_touch(M, ca, axa, cb, axb);
        }
        if (!__any(cand)) continue;
        bool hit = false;
        f3 P = mk3(0, 0, 0), PB = mk3(0, 0, 0), nrm = mk3(0, 0, 1);
        float dist = 0.f;
        Cvx A, B;
        if (cand) {
            frame(a, A);
            frame(bc, B);
            f3 pa, pb;
            float mg = M.margin;
            float dd = gjk_distance(M, A, B, 0.f, pa, pb);
            if (dd < 0.f) {
                dd = gjk_distance(M, A, B, kShrink, pa, pb);
                mg = M.margin + kShrink;
            }
            if (dd < 0.f) {
                const f3 d = A.c - B.c;
                const float nn = sqrtf(dot(d, d));
                nrm = nn > 0.f ? d * (1.0f / nn) : mk3(0.f, 0.f, 1.f);
                P = (A.c + B.c) * 0.5f;
                PB = P;
                dist = -2.0f * mg;
            } else {
                nrm = (pa - pb) * (1.0f / dd);
                dist = dd - 2.0f * mg;
                P = pa - nrm * mg;
                PB = pb + nrm * mg;
            }
            hit = dist < M.break_thr;
        }
        const unsigned long long bal = __ballot(hit);
        if (hit) {
            const int idx = nself + __popcll(bal & ((1ull << lane) - 1ull));
            if (idx < self_room) {
                float* geo = rows + LT::kGeoOff + (size_t)(LT::NC + idx) * LT::kGeo;
                f3 dA, dB;
                plane_space(nrm, dA, dB);
                dA = aniso_scale(M, B.R, aniso_scale(M, A.R, dA));
                dB = aniso_scale(M, B.R, aniso_scale(M, A.R, dB)) * M.fricB;
                st3(geo, P);
                geo[3] = dist;
                st3(geo + 4, dA);
                st3(geo + 7, dB);
                st3(geo + 10, nrm);
                st3(geo + 13, PB);
                geo[16] = (float)((a + 1) >> 1);
                geo[17] = (float)((bc + 1) >> 1);
                geo[18] = rho;
                geo[19] = 0.f;        // (no contact cache for these pairs: nothing to warm-start from)
            }
        }
        nself += __popcll(bal);
    }
    const int self_kept = nself < self_room ? nself : self_room;
    if ((nself > self_kept || n_ob > ob_kept) && lane == 0)      // counted, never silent (snk_contact_overflow)
        atomicAdd(ovf + 2, (unsigned long long)((nself - self_kept) + (n_ob - ob_kept)));
    if (ob_hit) {
        const int k = __popcll(ob_bal & ((1ull << lane) - 1ull));
        if (k < ob_kept) {
            const float mu_ob = fminf(M.mu_link * M.mu_obs, 10.0f);
            const float rho_ob = mu_ground > 0.f ? mu_ob / mu_ground : 0.f;
            float* geo = rows + LT::kGeoOff + (size_t)(LT::NC + self_kept + k) * LT::kGeo;
            f3 dA, dB;
            plane_space(ob_n, dA, dB);
            dA = aniso_scale(M, ob_R, dA);
            dB = aniso_scale(M, ob_R, dB) * M.fricB;
            st3(geo, ob_P);
            geo[3] = ob_dist;
            st3(geo + 4, dA);
            st3(geo + 7, dB);
            st3(geo + 10, ob_n);
            st3(geo + 13, ob_PB);
            geo[16] = (float)((a + 1) >> 1);
            geo[17] = (N <= 16 && M.obstacle == 2) ? (float)LT::kBoxBody : -1.0f;      // a free box is the contact's second body
            geo[18] = rho_ob;
            geo[19] = 0.f;
        }
    }
    return self_kept + ob_kept;

}

}  // namespace snk

