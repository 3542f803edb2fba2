// snk_model.hpp -- host-side model compiler: snk_params -> flat device model.
//
// Replaces what Bullet's URDF importer does for loadURDF(snake.urdf) (reference call site
// snake.py:92-93) -- but parametrically from the module constants of snake/snake.urdf
// (SURVEY.md Appendix B), and with the 2n+1 fixed joints MERGED: the stepper works on n+1
// composite rigid bodies joined by n revolute joints (a serial chain).  Merging is exact
// for rigid-body dynamics; Bullet's per-link damping (non-linear in |v|) is kept exact by
// carrying the sub-link masses/offsets of every composite.
//
//   body 0      = {kdl_dummy_root, base, INPUT_IF_1, COLLAR_1}         frame = root link
//   body k<n    = {OUTPUT_BODY_k, INPUT_IF_{k+1}, COLLAR_{k+1}}        frame = OUTPUT_BODY_k
//   body n      = {OUTPUT_BODY_n}
//   joint k     : body k-1 -> body k, axis = +y of body k, at the origin of body k
//   cylinder c  : c=0 INPUT_IF_1 (body 0); c=2k-1 OUTPUT_BODY_k, c=2k INPUT_IF_{k+1} (body k)
#pragma once
#include <cmath>
#include <cstring>

#include "../../include/snk.h"

namespace snk {

constexpr int kMaxN = 32;
constexpr int kMaxB = kMaxN + 1;
constexpr int kMaxCyl = 2 * kMaxN;
constexpr int kMaxSub = 4;

// Flat, float, uniform data read by every wave (scalar loads).
struct DevModel {
    int n, nb, ncyl;
    // integrator / solver
    float dt, inv_dt, gz, lin_damp, ang_damp, joint_damp, max_vel;
    float kp, kd, max_motor_imp, jlo, jhi, limit_erp, limit_max;
    float mu_link, aniso[3], contact_erp, slop, break_thr, margin, cyl_r, cyl_hl;
    float resid_thr;
    int n_iter, cone;
    // task
    float scaling, servo_tol, height_thr, energy_dt, alpha, beta, gamma;
    float term_angle, coll_force, coll_pen, done_pen;
    int gait, max_counter, term_index, act_dim, obs_dim, state_dim, rec_floats;
    // composite bodies (body frame)
    float mass[kMaxB];
    float com[kMaxB][3];
    float Ib[kMaxB][6];     // about body origin: xx xy xz yy yz zz
    float Irot[kMaxB][6];   // sum of sub-link inertias about their own COMs
    int nsub[kMaxB];
    float sub_m[kMaxB][kMaxSub];
    float sub_c[kMaxB][kMaxSub][3];
    float pfix[kMaxB][3];   // origin of body b in body b-1 frame (b >= 1)
    float Rfix[kMaxB][9];   // rotation body b (q=0) -> body b-1
    // collision cylinders
    int cyl_body[kMaxCyl];
    float cyl_c[kMaxCyl][3];   // centre in body frame
    float cyl_R[kMaxCyl][9];   // cylinder link frame -> body frame
    // contact model: hull_sides > 0 = the prism PyBullet imports a URDF <cylinder> as [U]; hull_xy[s] =
    // r (sin, cos)(2 pi s / hull_sides), the importer's vertex order; contact_model 1 = persistent manifold;
    // cyl_zoff = cylinder centre in its link's frame (snake.urdf:807,863), the manifold keeps link coordinates
    int hull_sides, contact_model, self_collision;
    int warm_start;                      // snk_params::warm_start; warm_factor = warmstarting_factor
    int poison;                          // SNK_POISON=1 at snk_create: LDS images start as NaNs (snk_device.hpp: poison)
    int hist;                            // snk_contact_histogram_enable: count every substep's contact points (one atomic each)
    float warm_factor;
    float fricB;                         // snk_params::friction_directions: 1.0 (two tangents), 0.0 (the second tangent gets no row)
    // snk_params::contact_order: the ground manifolds' place in the solver's sweep.  cyl_rank[c] = position of cylinder c,
    // cyl_at[r] = the cylinder at position r (identity for contact_order 0)
    int contact_order;
    unsigned char cyl_rank[kMaxCyl], cyl_at[kMaxCyl];
    int obstacle;                        // a static box on the ground (snake/block.urdf), contacts through the streamed-row solve
    float obs_c[3], obs_h[3], mu_obs;    // its centre, half extents, lateral friction
    // obstacle 2 (a free body): mass, inverse inertia diagonal in box axes, breaking threshold of its manifold with the plane
    float obs_minv, obs_iinv[3], obs_thr;
    float cyl_zoff;
    float hull_xy[32][2];
    // sensors
    float m_root;
    float zbase[3];   // z axis of the `base` link in body-0 frame (joint-0 force component)
    float hbase[3];   // COM of the `base` link in body-0 frame (height sample)
};

namespace detail {
inline void rpy(double r, double p, double y, double* R) {
    double cr = cos(r), sr = sin(r), cp = cos(p), sp = sin(p), cy = cos(y), sy = sin(y);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}
inline void mv(const double* R, const double* v, double* o) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
struct Sub {
    double m, c[3], I[9];   // mass, COM in body frame, inertia about own COM in body axes
};
inline void rot_diag(const double* R, const double* d, double* out) {   // R diag(d) R^T
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += R[3 * i + k] * d[k] * R[3 * j + k];
            out[3 * i + j] = s;
        }
}
}  // namespace detail

// Host copy in double, for tests (snk_model_describe) and for filling DevModel.
struct HostModel {
    int n;
    double mass[kMaxB], com[kMaxB][3], Ib[kMaxB][9], Irot[kMaxB][9];
    int nsub[kMaxB];
    detail::Sub sub[kMaxB][kMaxSub];
    double pfix[kMaxB][3], Rfix[kMaxB][9];
    int cyl_body[kMaxCyl];
    double cyl_c[kMaxCyl][3], cyl_R[kMaxCyl][9];
    double zbase[3], hbase[3], m_root;
};

inline void build_host_model(const snk_params& P, HostModel& H) {
    using namespace detail;
    memset(&H, 0, sizeof(H));
    const int n = P.n_modules;
    H.n = n;
    const double m_link = 0.103;                                        // snake.urdf:814,870
    const double I_file[3] = {5.4796e-05, 5.4796e-05, 3.4814e-05};      // snake.urdf:815,871
    const double mg = P.collision_margin;
    // [U] Bullet: inertia = box formula on the collision compound's AABB (three margins deep);
    // a link with no collision shape gets a margin-sized box.
    const double hx = 0.026 + 3 * mg, hz = 0.033 / 2 + 3 * mg, lx = 2 * hx, lz = 2 * hz;
    double I_cyl[3] = {m_link / 12 * (lx * lx + lz * lz), m_link / 12 * (lx * lx + lz * lz), m_link / 12 * (2 * lx * lx)};
    const double le = 2 * mg;
    double I_empty[3] = {P.default_mass / 12 * 2 * le * le, P.default_mass / 12 * 2 * le * le, P.default_mass / 12 * 2 * le * le};
    if (P.inertia_from_file) {
        for (int i = 0; i < 3; i++) { I_cyl[i] = I_file[i]; I_empty[i] = 1.0; }
    }
    double Rbase[9], Rz[9], Rid[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    rpy(0, -1.57079632679, 0, Rbase);                                   // snake.urdf:11
    rpy(0, 0, -1.57075, Rz);                                            // snake.urdf:877
    const double z_rev = 0.0366, z_out = 0.0273, z_cyl = 0.0183;        // urdf:836,877,807

    auto add = [&](int b, double m, const double* c, const double* R, const double* Idiag) {
        Sub& s = H.sub[b][H.nsub[b]++];
        s.m = m;
        for (int i = 0; i < 3; i++) s.c[i] = c[i];
        rot_diag(R, Idiag, s.I);
    };
    // body 0
    {
        double c0[3] = {0, 0, 0}, cb[3] = {0, 0, 0.026}, t[3], v[3] = {0, 0, z_rev}, cin[3];
        add(0, P.default_mass, c0, Rid, I_empty);                       // kdl_dummy_root (urdf:7)
        add(0, P.default_mass, cb, Rbase, I_empty);                     // base (urdf:14)
        mv(Rbase, v, t);
        for (int i = 0; i < 3; i++) cin[i] = cb[i] + t[i];
        add(0, m_link, cin, Rbase, I_cyl);                              // INPUT_IF_1 (urdf:796-817)
        add(0, P.default_mass, cb, Rbase, I_empty);                     // COLLAR_1 (urdf:818-828)
        H.m_root = P.default_mass;
        double ez[3] = {0, 0, 1};
        mv(Rbase, ez, H.zbase);
        for (int i = 0; i < 3; i++) H.hbase[i] = cb[i];
    }
    for (int b = 1; b <= n; b++) {
        double c0[3] = {0, 0, 0};
        add(b, m_link, c0, Rid, I_cyl);                                 // OUTPUT_BODY_b (urdf:851-873)
        if (b < n) {
            double cin[3] = {0, 0, z_out + z_rev}, ccol[3] = {0, 0, z_out};
            add(b, m_link, cin, Rz, I_cyl);                             // INPUT_IF_{b+1}
            add(b, P.default_mass, ccol, Rz, I_empty);                  // COLLAR_{b+1}
        }
    }
    // composites
    for (int b = 0; b <= n; b++) {
        double m = 0, mc[3] = {0, 0, 0};
        for (int s = 0; s < H.nsub[b]; s++) {
            m += H.sub[b][s].m;
            for (int i = 0; i < 3; i++) mc[i] += H.sub[b][s].m * H.sub[b][s].c[i];
        }
        H.mass[b] = m;
        for (int i = 0; i < 3; i++) H.com[b][i] = mc[i] / m;
        for (int i = 0; i < 9; i++) { H.Ib[b][i] = 0; H.Irot[b][i] = 0; }
        for (int s = 0; s < H.nsub[b]; s++) {
            const Sub& S = H.sub[b][s];
            double c2 = S.c[0] * S.c[0] + S.c[1] * S.c[1] + S.c[2] * S.c[2];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    H.Irot[b][3 * i + j] += S.I[3 * i + j];
                    H.Ib[b][3 * i + j] += S.I[3 * i + j] + S.m * ((i == j ? c2 : 0) - S.c[i] * S.c[j]);
                }
        }
    }
    // joints
    for (int b = 1; b <= n; b++) {
        if (b == 1) {
            double v[3] = {0, 0, z_rev}, t[3];
            mv(Rbase, v, t);
            H.pfix[b][0] = t[0]; H.pfix[b][1] = t[1]; H.pfix[b][2] = 0.026 + t[2];
            memcpy(H.Rfix[b], Rbase, sizeof(Rbase));
        } else {
            H.pfix[b][0] = 0; H.pfix[b][1] = 0; H.pfix[b][2] = z_out + z_rev;
            memcpy(H.Rfix[b], Rz, sizeof(Rz));
        }
    }
    // cylinders
    for (int c = 0; c < 2 * n; c++) {
        int b = (c + 1) / 2;
        H.cyl_body[c] = b;
        if (c == 0) {
            double v[3] = {0, 0, z_cyl}, t[3];
            mv(Rbase, v, t);
            H.cyl_c[c][0] = t[0]; H.cyl_c[c][1] = t[1]; H.cyl_c[c][2] = 0.026 + t[2];
            memcpy(H.cyl_R[c], Rbase, sizeof(Rbase));
        } else if (c & 1) {   // OUTPUT_BODY_b
            H.cyl_c[c][0] = 0; H.cyl_c[c][1] = 0; H.cyl_c[c][2] = z_cyl;
            memcpy(H.cyl_R[c], Rid, sizeof(Rid));
        } else {              // INPUT_IF_{b+1}
            H.cyl_c[c][0] = 0; H.cyl_c[c][1] = 0; H.cyl_c[c][2] = z_out + z_cyl;
            memcpy(H.cyl_R[c], Rz, sizeof(Rz));
        }
    }
}

inline void build_dev_model(const snk_params& P, const HostModel& H, DevModel& D) {
    memset(&D, 0, sizeof(D));
    const int n = H.n;
    D.n = n; D.nb = n + 1; D.ncyl = 2 * n;
    D.dt = (float)P.dt; D.inv_dt = (float)(1.0 / P.dt); D.gz = (float)P.gravity_z;
    D.lin_damp = (float)P.lin_damping; D.ang_damp = (float)P.ang_damping;
    D.joint_damp = (float)P.joint_damping; D.max_vel = (float)P.max_coord_vel;
    D.kp = (float)P.kp; D.kd = (float)P.kd; D.max_motor_imp = (float)P.max_motor_impulse;
    D.jlo = (float)P.joint_lo; D.jhi = (float)P.joint_hi;
    D.limit_erp = (float)P.limit_erp; D.limit_max = (float)P.limit_max_impulse;
    D.mu_link = (float)P.mu_link;
    for (int i = 0; i < 3; i++) D.aniso[i] = (float)P.aniso[i];
    D.contact_erp = (float)P.contact_erp; D.slop = (float)P.linear_slop;
    D.margin = (float)P.collision_margin;
    // the link collider's cylinder (snake.urdf:806-811, 862-867): radius, half length, centre along the link's z.  ONE set
    // of numbers for the model, the hull's vertices and the contact threshold (tests/test_urdf_and_live.py holds them
    // against the URDF-derived golden values)
    const double kCylR = 0.026, kCylHL = 0.033 / 2, kCylZ = 0.0183;
    D.cyl_r = (float)kCylR; D.cyl_hl = (float)kCylHL; D.cyl_zoff = (float)kCylZ;
    {
        // [U] btCollisionDispatcher::getNewManifold with CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD (the dispatcher's
        // default): gContactBreakingThreshold x the smaller shape's angular-motion disc = |centre| + bounding-sphere
        // radius of its AABB.  A link collider is a compound with the cylinder's hull (margin included) 0.0183 m from
        // the link's inertial frame (snake.urdf:807,813,863,869); the plane's and the box's discs are larger.
        const double ax = kCylR + P.collision_margin, az = kCylHL + P.collision_margin;
        const double disc = kCylZ + std::sqrt(ax * ax + ax * ax + az * az);
        D.break_thr = (float)(P.relative_breaking_threshold ? P.breaking_threshold * disc : P.breaking_threshold);
    }
    D.warm_start = (P.warm_start && P.contact_model == 1) ? 1 : 0;
    D.warm_factor = (float)P.warmstarting_factor;
    D.hull_sides = P.hull_sides; D.contact_model = P.contact_model;
    // (link-link rows are built by the streamed-row solve; a 16-link substep with a pair of links within reach of each
    //  other goes there: snk_pgs_v2.hpp, find_contacts_v2)
    D.self_collision = P.self_collision ? 1 : 0;
    D.obstacle = P.obstacle;
    for (int i = 0; i < 3; i++) { D.obs_c[i] = (float)P.obstacle_pos[i]; D.obs_h[i] = (float)P.obstacle_half[i]; }
    D.mu_obs = (float)P.mu_obstacle;
    {
        // [U] no URDF_USE_INERTIA_FROM_FILE: btBoxShape::calculateLocalInertia on the nominal box (block.urdf:7 otherwise)
        const double lx = 2 * P.obstacle_half[0], ly = 2 * P.obstacle_half[1], lz = 2 * P.obstacle_half[2], m = P.obstacle_mass;
        double I[3] = {m / 12 * (ly * ly + lz * lz), m / 12 * (lx * lx + lz * lz), m / 12 * (lx * lx + ly * ly)};
        if (P.inertia_from_file) { I[0] = 1.0; I[1] = 100.0; I[2] = 1.0; }
        D.obs_minv = (float)(m > 0 ? 1.0 / m : 0.0);
        for (int i = 0; i < 3; i++) D.obs_iinv[i] = (float)(I[i] > 0 ? 1.0 / I[i] : 0.0);
        const double hn = std::sqrt(P.obstacle_half[0] * P.obstacle_half[0] + P.obstacle_half[1] * P.obstacle_half[1] +
                                    P.obstacle_half[2] * P.obstacle_half[2]);
        // its manifold with the plane: the box's own angular-motion disc (|half extents|), smaller than the plane's [U]
        D.obs_thr = (float)(P.relative_breaking_threshold ? P.breaking_threshold * hn : P.breaking_threshold);
    }
    for (int s = 0; s < P.hull_sides && s < 32; s++) {
        const double th = 2.0 * 3.14159265358979323846 * s / P.hull_sides;
        D.hull_xy[s][0] = (float)(kCylR * sin(th));
        D.hull_xy[s][1] = (float)(kCylR * cos(th));
    }
    D.resid_thr = (float)P.residual_threshold;
    D.n_iter = P.n_iterations;
    // (Bullet enters the implicit-cone branch only under SOLVER_USE_2_FRICTION_DIRECTIONS [U]: with one direction the
    //  friction rows go through the box-bounded loop, which skips a row while its contact carries no normal impulse)
    D.cone = (P.cone_friction && P.friction_directions == 2) ? 1 : 0;
    D.fricB = P.friction_directions == 1 ? 0.0f : 1.0f;
    D.contact_order = P.contact_order;
    {
        // The oracle's rule (oracle/snake_oracle.cpp: find_contacts): 1 = reversed; 2 = link order after Bullet's unstable
        // quickSort on equal keys; k >= 3 = the cylinder links sorted by a splitmix-style hash of (k, link).  `link` is the index of the cylinder's link in the unmerged URDF tree with the
        // root at 0 -- INPUT_INTERFACE_k = 3 k - 1, OUTPUT_BODY_k = 3 k + 1 -- so that both sides sort the same keys.
        const int nc2 = 2 * n;
        unsigned long long key[kMaxCyl];
        // contact_order 2: where btAlignedObjectArray::quickSort (Hoare partition, pivot = the middle element, `i <= j` swap:
        // on all-equal keys every partition reverses its range and recurses into the halves) leaves element c of a list
        // of nc2 equal keys -- the island manager's sort of the plane-link manifolds, restated (oracle: qs_equal_keys)
        int qs_perm[kMaxCyl], qs_pos[kMaxCyl];
        for (int c = 0; c < nc2; c++) qs_perm[c] = c;
        {
            int stack[2 * kMaxCyl][2], sp = 0;
            if (nc2 > 1) { stack[sp][0] = 0; stack[sp][1] = nc2 - 1; sp++; }
            while (sp > 0) {
                sp--;
                const int lo = stack[sp][0], hi = stack[sp][1];
                int i = lo, j = hi;
                do {
                    if (i <= j) { const int t = qs_perm[i]; qs_perm[i] = qs_perm[j]; qs_perm[j] = t; i++; j--; }
                } while (i <= j);
                // (the two recursive calls work on disjoint ranges: their order does not matter)
                if (i < hi) { stack[sp][0] = i; stack[sp][1] = hi; sp++; }
                if (lo < j) { stack[sp][0] = lo; stack[sp][1] = j; sp++; }
            }
        }
        for (int c = 0; c < nc2; c++) qs_pos[qs_perm[c]] = c;
        for (int c = 0; c < nc2; c++) {
            const int link = (c & 1) ? 3 * (c + 1) / 2 + 1 : 3 * c / 2 + 2;
            if (P.contact_order == 0) key[c] = (unsigned long long)c;
            else if (P.contact_order == 1) key[c] = (unsigned long long)(nc2 - 1 - c);
            else if (P.contact_order == 2) key[c] = (unsigned long long)qs_pos[c];
            else {
                unsigned long long z = (unsigned long long)P.contact_order * 0x9E3779B97F4A7C15ull + (unsigned long long)(link + 1) * 0xBF58476D1CE4E5B9ull;
                z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
                key[c] = z;
            }
        }
        for (int c = 0; c < kMaxCyl; c++) { D.cyl_rank[c] = (unsigned char)c; D.cyl_at[c] = (unsigned char)c; }
        for (int c = 0; c < nc2; c++) {
            int r = 0;
            for (int o = 0; o < nc2; o++) r += (key[o] < key[c] || (key[o] == key[c] && o < c)) ? 1 : 0;
            D.cyl_rank[c] = (unsigned char)r;
            D.cyl_at[r] = (unsigned char)c;
        }
    }
    D.scaling = (float)P.scaling_factor; D.servo_tol = (float)P.servo_tol;
    D.height_thr = (float)P.height_threshold; D.energy_dt = (float)P.energy_dt;
    D.alpha = (float)P.alpha; D.beta = (float)P.beta; D.gamma = (float)P.gamma;
    D.term_angle = (float)P.term_angle; D.coll_force = (float)P.collision_force;
    D.coll_pen = (float)P.collision_penalty; D.done_pen = (float)P.done_penalty;
    D.gait = P.gait; D.max_counter = P.max_counter; D.term_index = P.term_index;
    D.act_dim = (P.gait == 0 || P.gait == 1) ? n / 2 : n;
    D.obs_dim = 3 * n + 8;
    D.state_dim = 13 + 2 * n;
    D.rec_floats = (n <= 16) ? 64 : 128;     // state + aux, padded to whole 256-B lines
    auto sym6 = [](const double* M, float* o) {
        o[0] = (float)M[0]; o[1] = (float)M[1]; o[2] = (float)M[2];
        o[3] = (float)M[4]; o[4] = (float)M[5]; o[5] = (float)M[8];
    };
    for (int b = 0; b <= n; b++) {
        D.mass[b] = (float)H.mass[b];
        for (int i = 0; i < 3; i++) D.com[b][i] = (float)H.com[b][i];
        sym6(H.Ib[b], D.Ib[b]);
        sym6(H.Irot[b], D.Irot[b]);
        D.nsub[b] = H.nsub[b];
        for (int s = 0; s < H.nsub[b]; s++) {
            D.sub_m[b][s] = (float)H.sub[b][s].m;
            for (int i = 0; i < 3; i++) D.sub_c[b][s][i] = (float)H.sub[b][s].c[i];
        }
        for (int i = 0; i < 3; i++) D.pfix[b][i] = (float)H.pfix[b][i];
        for (int i = 0; i < 9; i++) D.Rfix[b][i] = (float)H.Rfix[b][i];
    }
    for (int c = 0; c < 2 * n; c++) {
        D.cyl_body[c] = H.cyl_body[c];
        for (int i = 0; i < 3; i++) D.cyl_c[c][i] = (float)H.cyl_c[c][i];
        for (int i = 0; i < 9; i++) D.cyl_R[c][i] = (float)H.cyl_R[c][i];
    }
    D.m_root = (float)H.m_root;
    for (int i = 0; i < 3; i++) { D.zbase[i] = (float)H.zbase[i]; D.hbase[i] = (float)H.hbase[i]; }
}

}  // namespace snk
