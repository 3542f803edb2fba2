// snk_freebox.hpp -- obstacle 2: the box of snake/block.urdf as the reference loads it (snake_gait_test.py:51,
// Snake.add_obstacle snake.py:83-84: useFixedBase=0), a FREE 200-kg body resting on the ground.
// Included behind snk_pgs_v2.hpp (it uses the manifold helpers defined there); used by the streamed-row kernels only.
#pragma once

namespace snk {

// ----------------------------------------------------------------------------------------------------------------
// obstacle 2: the box of snake/block.urdf as the reference loads it -- a FREE body (a btMultiBody without links [U]).
// Its state lives in LDS while a wave holds the environment (Lds<N, false>::box ...); everything below is evaluated
// wave-uniformly (every lane computes the same few dozen flops, lane 0 stores).
// ----------------------------------------------------------------------------------------------------------------
// world rotation and world inverse inertia of the box for this substep
template <class LT>
__device__ __forceinline__ void box_frame_v1(LT& L, const DevModel& M, int lane) {
    const float qx = L.box[3], qy = L.box[4], qz = L.box[5], qw = L.box[6];
    const float s2 = 2.0f / (qx * qx + qy * qy + qz * qz + qw * qw);
    const float xs = qx * s2, ys = qy * s2, zs = qz * s2;
    const float wx = qw * xs, wy = qw * ys, wz = qw * zs, xx = qx * xs, xy = qx * ys, xz = qx * zs, yy = qy * ys, yz = qy * zs, zz = qz * zs;
    const float R[9] = {1 - (yy + zz), xy - wz, xz + wy, xy + wz, 1 - (xx + zz), yz - wx, xz - wy, yz + wx, 1 - (xx + yy)};
    const float d[6] = {M.obs_iinv[0], 0.f, 0.f, M.obs_iinv[1], 0.f, M.obs_iinv[2]};
    float W[6];
    rotSym(R, d, W);
    lds_sync();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) L.bR[i] = R[i];
#pragma unroll
        for (int i = 0; i < 6; i++) L.bIw[i] = W[i];
    }
    lds_sync();
}

// The box against the ground [U]: btBoxShape is polyhedral, so the same convex-plane algorithm and persistent manifold
// as a link's hull -- refresh, then one new support corner per step (localGetSupportingVertex: the nominal corner,
// the box keeps its margin inside), <= 4 cached points, threshold DevModel::obs_thr.  Appends its records behind the
// `first` contacts already there (geometry slots NC + first ...): body kBoxBody against the world, normal +z, friction
// directions btPlaneSpace1(+z) unscaled (no anisotropic friction on the box), coefficient mu_obstacle x the plane's.
// Returns the number of points (<= 4, less if the room for these contacts is used up: counted).
template <class LT>
__device__ __forceinline__ int find_box_ground_v1(LT& L, const DevModel& M, int lane, float mu_ground, int first,
                                                  float* __restrict__ rows, unsigned long long* __restrict__ ovf) {
    MPt p[4];
    f3 wa[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        p[j].a = mk3(L.bman[6 * j], L.bman[6 * j + 1], L.bman[6 * j + 2]);
        p[j].b = mk3(L.bman[6 * j + 3], L.bman[6 * j + 4], 0.f);
        p[j].lam = L.bman[6 * j + 5];
        p[j].d = 0.f;
    }
    int n = L.bmn;
    n = n < 0 ? 0 : (n > 4 ? 4 : n);
    const float thr = M.obs_thr;
    const f3 centre = ld3(L.box);
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = L.bR[i];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        wa[j] = centre + mulRv(R, p[j].a);
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
    const f3 dl = mk3(-R[6], -R[7], -R[8]);                  // world "down" in box coordinates
    MPt np;
    np.a = mk3(dl.x >= 0.f ? M.obs_h[0] : -M.obs_h[0], dl.y >= 0.f ? M.obs_h[1] : -M.obs_h[1], dl.z >= 0.f ? M.obs_h[2] : -M.obs_h[2]);
    const f3 wnew = centre + mulRv(R, np.a);
    np.d = wnew.z;
    np.b = mk3(wnew.x, wnew.y, 0.f);
    np.lam = 0.f;
    {
        const bool add = np.d < thr;
        int nearest = -1;
        float shortest = thr * thr, lam_near = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f3 d = p[j].a - np.a;
            const float dd = dot(d, d);
            const bool nr = j < n && dd < shortest;
            shortest = nr ? dd : shortest; nearest = nr ? j : nearest; lam_near = nr ? p[j].lam : lam_near;
        }
        const int evict = manifold_sort_cached(p, np);
        np.lam = nearest >= 0 ? lam_near : 0.f;
        const int where = nearest >= 0 ? nearest : (n < 4 ? n : evict);
        n = (add && nearest < 0 && n < 4) ? n + 1 : n;
#pragma unroll
        for (int j = 0; j < 4; j++) { mpt_sel(p[j], np, add && j == where); f3_sel(wa[j], wnew, add && j == where); }
    }
    int kept = n;
    if (first + kept > LT::kMaxSelf) {
        kept = LT::kMaxSelf - first;
        kept = kept < 0 ? 0 : kept;
        if (lane == 0) atomicAdd(ovf + 2, (unsigned long long)(n - kept));
    }
    lds_sync();
    if (lane == 0) {
        L.bmn = n;
        const float mu_bg = fminf(M.mu_obs * (M.mu_link > 0.f ? mu_ground / M.mu_link : 0.f), 10.0f);
        const float rho = mu_ground > 0.f ? mu_bg / mu_ground : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j >= kept) p[j].lam = j < n ? 0.f : p[j].lam;       // a point without rows carries no impulse
            L.bman[6 * j] = p[j].a.x; L.bman[6 * j + 1] = p[j].a.y; L.bman[6 * j + 2] = p[j].a.z;
            L.bman[6 * j + 3] = p[j].b.x; L.bman[6 * j + 4] = p[j].b.y; L.bman[6 * j + 5] = p[j].lam;
            if (j < kept) {
                float* geo = rows + LT::kGeoOff + (size_t)(LT::NC + first + j) * LT::kGeo;
                st3(geo, wa[j]);
                geo[3] = p[j].d;
                st3(geo + 4, mk3(0.f, -1.f, 0.f));
                st3(geo + 7, mk3(M.fricB, 0.f, 0.f));
                st3(geo + 10, mk3(0.f, 0.f, 1.f));
                st3(geo + 13, mk3(0.f, 0.f, 0.f));
                geo[16] = (float)LT::kBoxBody; geo[17] = -1.0f; geo[18] = rho;
                geo[19] = M.warm_start ? p[j].lam * M.warm_factor : 0.f;
            }
        }
    }
    lds_sync();
    return kept;
}

}  // namespace snk
