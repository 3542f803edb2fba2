/*
 * snake_oracle.h -- C API of the CPU ORACLE for the SnakeGymEnv step/reset path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the arithmetic of the reference path lives in the third-party
 * `pybullet` wheel (bullet3, version not pinned by the reference; absent from this
 * image and from /root/reference).  This oracle restates Bullet's published
 * btMultiBody pipeline from knowledge of the public bullet3 sources; every such
 * rule is tagged [U] in snake_oracle.cpp and is a field of orc_params.  The
 * reference holds no golden vectors for this path.
 *
 * Reference call sites restated (paths under /root/reference):
 *   snake.py:86-101   Snake.reset            -> orc_reset / orc_hard_reset
 *   snake.py:209-217  Snake.getObservation   -> orc_get_obs
 *   snake.py:219-225  applyActions           -> motor targets in orc_substep
 *   snake.py:228-235  checkFeedback          -> inside orc_env_step
 *   snake.py:237-245  checkSnakeHeight       -> orc_mean_height
 *   snake.py:247-269  createAction           -> inside orc_env_step
 *   snake.py:274-306  Snake.step             -> inside orc_env_step
 *   snake.py:286      pybullet.stepSimulation-> orc_substep
 *   snake.py:336-341  calculateEnergy        -> inside orc_env_step
 *   SnakeGymEnv.py:33-50,82-103 step / reward / termination -> orc_env_step
 *   ppo/multiprocessing_env.py:11-16 worker auto-reset       -> orc_env_step(vec_mode=1)
 */
#ifndef SNAKE_ORACLE_H
#define SNAKE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_params {
    /* model (snake/snake.urdf constants) */
    int32_t n_modules;        /* 16 (reference) or 32 (BASELINE config 4)                */
    int32_t inertia_from_file;/* 0: Bullet default, inertia from collision AABB [U]      */
    double  default_mass;     /* mass given to links without <inertial> [U] = 1          */
    double  collision_margin; /* gUrdfDefaultCollisionMargin [U] = 0.001                 */
    int32_t hull_sides;       /* 32 (default) = PyBullet's 32-gon hull import of a URDF <cylinder> [U] (snake.py:93 passes
                               * no URDF_USE_IMPLICIT_CYLINDER); 0 = implicit cylinder (the round-1 model)       */
    int32_t contact_model;    /* 1 (default) = Bullet's persistent manifold [U]: one new support point per cylinder
                               *     per step merged into a cache of <= 4, refreshed / dropped at the breaking threshold;
                               * 0 = stateless: both end-cap points of every cylinder, every step (the round-1 model) */
    int32_t max_contacts;     /* 0 = no limit; > 0: only the first max_contacts points (in manifold order) get
                               * rows -- mirrors the product's structural limit of 4n contacts (tests only; the
                               * product counts what it drops: snk_contact_overflow)                           */
    int32_t self_collision;   /* 1: link-link contacts of non-adjacent cylinder links (URDF_USE_SELF_COLLISION,
                               * snake.py:93 [U]) for any chain length; 0 (default of this test tool): none.
                               * For the 16-link snake the rows are speculative only and never carry an impulse
                               * inside the joint limits (tools/self_collision_clearance.py, and a test);
                               * the 32-link tests switch it on (the product evaluates it for 32 links)        */
    int32_t max_self_contacts;/* 0 = no limit; > 0 mirrors the product's cap on link-link + obstacle contacts  */
    int32_t obstacle;         /* the box of snake/block.urdf (snake.py:83-84,94; snake_gait_test.py:51).  1: STATIC;
                               * 2: as the reference loads it (useFixedBase=0): a FREE body of obstacle_mass resting on
                               * the ground -- a btMultiBody without links [U]: gravity, link damping, its own
                               * persistent manifold with the plane (one support corner per step, <= 4 cached),
                               * two-body rows with the snake's links                                            */
    int32_t pair_manifold;    /* 0 (default): link-link and link-box contacts are ONE stateless point per pair per step
                               * (deviations 1-2, shared with the kernels).  1: Bullet's own bookkeeping
                               * for those pairs too [U] -- a btPersistentManifold of <= 4 points per pair: refreshed
                               * from both poses (distance along the stored normal; removed when separated or drifted
                               * beyond the breaking threshold), then this step's GJK point merged in (getCacheEntry /
                               * replaceContactPoint / addManifoldPoint with sortCachedPoints).  Oracle only: measures
                               * what the deviation is worth (tests/test_oracle_pair_manifold.py)                */
    double  obstacle_pos[3];  /* centre [2, 0, 0.1]                                                            */
    double  obstacle_half[3]; /* half extents [0.1, 0.4, 0.1] (block.urdf:16)                                  */
    double  mu_obstacle;      /* 0.5 [U]                                                                       */
    double  obstacle_mass;    /* 200 (block.urdf:6); inertia from the box shape [U] (inertia_from_file: 1, 100, 1) */
    /* world / integrator */
    double  dt;               /* PyBullet default fixedTimeStep 1/240 [U] (F2)           */
    double  gravity_z;        /* snake.py:8  -9.8                                        */
    double  lin_damping;      /* btMultiBody m_linearDamping  0.04 [U]                   */
    double  ang_damping;      /* btMultiBody m_angularDamping 0.04 [U]                   */
    double  joint_damping;    /* snake.urdf:838 damping=.1                               */
    double  max_coord_vel;    /* btMultiBody m_maxCoordinateVelocity 100 [U]             */
    /* motors (snake.py:219-221, PyBullet defaults [U]) */
    double  kp;               /* positionGain default 0.1                                */
    double  kd;               /* velocityGain default 1.0                                */
    double  max_motor_impulse;/* forces=[inf] -> inf                                     */
    double  joint_lo, joint_hi;/* snake.urdf:839 +-1.57                                  */
    double  limit_erp;        /* btContactSolverInfo m_erp 0.2 [U]                       */
    double  limit_max_impulse;/* btMultiBodyConstraint m_maxAppliedImpulse 100 [U]       */
    /* contact */
    double  mu_link;          /* snake.py:104-106 lateralFriction=2                      */
    double  aniso[3];         /* snake.py:25 [1, 0.1, 0.01]                              */
    double  contact_erp;      /* solver m_erp2 [U] (PyBullet sets 0.08? default 0.2)     */
    double  linear_slop;      /* PyBullet m_linearSlop 1e-5 [U]                          */
    double  breaking_threshold;/* gContactBreakingThreshold 0.02 [U]                     */
    int32_t relative_breaking_threshold; /* 1 (default): btCollisionDispatcher's CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD
                               * [U], on by default in Bullet: a manifold's threshold is breaking_threshold x the smaller
                               * angular-motion disc of its two shapes (|AABB centre| + AABB half diagonal of a link's
                               * compound: 0.0603 m -> 1.206 mm); 0: breaking_threshold itself                     */
    int32_t cone_friction;    /* 1: implicit cone on the 2 friction rows [U]; 0: pyramid */
    /* solver */
    int32_t n_iterations;     /* numSolverIterations 50 [U]                              */
    double  residual_threshold;/* m_leastSquaresResidualThreshold 1e-7 [U]; 0 = never exit */
    int32_t warm_start;       /* 0 (default): btMultiBodyConstraintSolver starts every contact row at zero impulse [U]
                               * (setupMultiBodyContactConstraint has its warm start disabled); 1: SOLVER_USE_WARMSTARTING
                               * as the rigid-body solver does it -- a cached point's normal row starts at
                               * warmstarting_factor x the impulse it carried last step, delta-v starts at
                               * sum M^-1 J^T of those; friction rows start at zero.  contact_model 1 only             */
    double  warmstarting_factor; /* btContactSolverInfo m_warmstartingFactor 0.85 [U] (the struct's default; PyBullet's
                               * world constructor is read as setting 0.1 [U]: an error-bar row, profiles/r05_u_rows.json) */
    int32_t friction_directions; /* 2 (default): SOLVER_USE_2_FRICTION_DIRECTIONS [U] -- both btPlaneSpace1 tangents get a
                               * row (and, with cone_friction, the implicit cone over the pair).  1: Bullet's multibody
                               * solver WITHOUT that flag [U]: one row per contact along the first btPlaneSpace1 tangent
                               * ((0,-1,0) for the ground), box bounds +-mu lambda_n, no cone branch.  Oracle only
                               * (VERDICT r4 item 6): on flat ground that leaves world x without any friction -- a body
                               * pushed along x would never stop -- which PyBullet visibly does not do             */
    int32_t contact_erp_rule; /* 0 (default): contact rows use contact_erp (m_erp2 0.08) whatever the depth.  1: the
                               * choice as setupMultiBodyContactConstraint is read to make it [U]: m_erp (= limit_erp,
                               * 0.2) unless split impulse is on AND the penetration is deeper than
                               * m_splitImpulsePenetrationThreshold (-0.04 m), only then m_erp2.  (Joint-limit rows:
                               * fillMultiBodyConstraint has the same test commented out and always takes m_erp: that is
                               * limit_erp already.)  Oracle only, an error-bar row                                   */
    /* task (snake.py / SnakeGymEnv.py) */
    double  scaling_factor;   /* snake.py:63  pi/6                                       */
    int32_t gait;             /* snake.py:62  1 -> odd slots                             */
    double  servo_tol;        /* snake.py:232 0.05                                       */
    int32_t max_counter;      /* snake.py:303 40                                         */
    double  height_threshold; /* snake.py:238 0.1                                        */
    double  energy_dt;        /* snake.py:9   1/100                                      */
    double  alpha, beta, gamma;/* SnakeGymEnv.py:14-16  1, 0.01, 0.1                     */
    double  term_angle;       /* SnakeGymEnv.py:100 0.5 on obs[9]                        */
    int32_t term_index;       /* 9                                                       */
    double  collision_force;  /* SnakeGymEnv.py:94  10                                   */
    double  collision_penalty;/* -10                                                     */
    double  done_penalty;     /* SnakeGymEnv.py:40  -5                                   */
    /* row order (round 6): error-bar rows, oracle only */
    int32_t noncontact_order; /* 0 (default): violated joint limits by joint index, then the motors by joint index.
                               * 1 [U]: the order btMultiBodyDynamicsWorld::solveConstraints hands the solver -- the world's
                               * list [limit_1..limit_n, motor_1..motor_n] after btAlignedObjectArray::quickSort on equal
                               * island ids (not stable: a fixed non-identity permutation, orc_quicksort_equal_keys)    */
    int32_t contact_order;    /* 0 (default): ground manifolds in link order.  1: reversed.  2: link order after the island
                               * manager's unstable quickSort on equal island ids (orc_quicksort_equal_keys over the 2n
                               * plane-link manifolds: the restatable candidate).  k >= 3: the fixed permutation of the links
                               * a hash of (k, link) gives.  Bullet's island-manager order is unknown [U]               */
} orc_params;

typedef struct orc_env orc_env;

void     orc_default_params(orc_params* p);
orc_env* orc_create(const orc_params* p);
void     orc_destroy(orc_env* e);
void     orc_set_plane_friction(orc_env* e, double mu);          /* BASELINE config 5 */

int32_t  orc_num_links(const orc_env* e);    /* 3*n+2 (root + Bullet links 0..3n)       */
int32_t  orc_num_dofs(const orc_env* e);     /* 6 + n                                    */
int32_t  orc_obs_dim(const orc_env* e);      /* 3n + 8                                   */
int32_t  orc_state_dim(const orc_env* e);    /* 13 + 2n                                  */

/* state = [pos3, quat xyzw 4, omega_world 3, vel_world 3, q n, qd n] */
void     orc_get_state(const orc_env* e, double* s);
void     orc_set_state(orc_env* e, const double* s);
/* carried observables: motor torques [n], joint-0 reaction Fz, prev obs x */
void     orc_get_aux(const orc_env* e, double* tau_n, double* fz, double* prev_x);
void     orc_set_aux(orc_env* e, const double* tau_n, double fz, double prev_x);

void     orc_hard_reset(orc_env* e);                  /* snake.py:88-95  */
void     orc_reset(orc_env* e, double* obs);          /* SnakeGymEnv.py:28-31 (soft) */
void     orc_get_obs(const orc_env* e, double* obs);  /* snake.py:209-217 */
double   orc_mean_height(orc_env* e);                 /* snake.py:237-245 (value) */
/* getJointState(robot, 3)[2][2] of the last substep (snake_gait_test.py:33-40,126: "> 20: hit the wall") */
double   orc_joint3_reaction_fz(const orc_env* e);

/* obstacle 2: the box's state [pos3, quat xyzw 4, omega_world 3, vel_world 3] and its manifold with the plane
 * [count, 4 x (point on the box in box coordinates 3, point on the ground 3, applied normal impulse)] = 29 */
void     orc_get_box_state(const orc_env* e, double* s13);
void     orc_set_box_state(orc_env* e, const double* s13);
void     orc_get_box_manifold(const orc_env* e, double* m29);
void     orc_set_box_manifold(orc_env* e, const double* m29);

/* contact cache of contact_model 1, per cylinder in link order:
 * [count, 4 x (point on the link in link coordinates 3, point on the ground in world coordinates 3, the normal impulse
 *  the point carried in the last substep: btManifoldPoint::m_appliedImpulse)] */
int32_t  orc_manifold_floats(const orc_env* e);                 /* 2n * 29 */
void     orc_get_manifold(const orc_env* e, double* out);
void     orc_set_manifold(orc_env* e, const double* in);

/* one physics substep (pybullet.stepSimulation with POSITION_CONTROL targets) */
void     orc_substep(orc_env* e, const double* targets_n);
int32_t  orc_last_iterations(const orc_env* e);
int32_t  orc_last_num_contacts(const orc_env* e);

/* SnakeGymEnv.step.  vec_mode=1 adds the SubprocVecEnv worker's reset-on-done
 * (returned obs is the post-reset one).  action[A] is clipped in place.        */
void     orc_env_step(orc_env* e, double* action, int32_t vec_mode,
                      double* obs, double* reward, int32_t* done, int32_t* substeps);

/* --- introspection for tests --- */
/* per link: world COM position (3) */
void     orc_link_com_world(orc_env* e, double* out_L3);
/* per revolute joint: world axis (3) and origin (3) */
void     orc_joint_axes_world(orc_env* e, double* axis_n3, double* origin_n3);
/* per link: mass, com(3, link frame), inertia diag(3) */
void     orc_link_inertials(const orc_env* e, double* out_L7);
void     orc_link_parents(const orc_env* e, int32_t* out_L);
/* forward dynamics only: generalized acceleration [6+n] = [omega_dot_w, vdot_w, qdd]
 * for joint torques tau[n] (gravity+damping per flags), no constraints          */
void     orc_forward_dynamics(orc_env* e, const double* tau_n, int32_t with_gravity,
                              int32_t with_damping, double* acc);
/* y = M^-1 x for x in generalized force space [6+n] (ABA delta pass)            */
void     orc_minv_mul(orc_env* e, const double* x, double* y);
/* total linear momentum (3), angular momentum about world origin (3), kinetic energy */
void     orc_momentum(orc_env* e, double* lin3, double* ang3, double* kinetic);
/* contacts of the current pose: returns count; per contact [px,py,pz, dist, link] */
int32_t  orc_contacts(orc_env* e, double* out, int32_t max_contacts);
/* world frames of the collision cylinders in link order: [centre 3, rotation 9 row-major] each; returns the count */
int32_t  orc_cylinder_frames(orc_env* e, double* out);
/* contacts with both participants: per contact [P 3, dist, link, linkB (-1 ground, -2 obstacle box), n 3, PB 3];
 * like orc_contacts: of the current pose, as the next substep would find them, cache and last list untouched */
int32_t  orc_contacts_full(orc_env* e, double* out, int32_t max_contacts);
/* ... and the contacts the LAST substep solved (aligned with orc_last_normal_impulses) */
int32_t  orc_last_contacts_full(const orc_env* e, double* out, int32_t max_contacts);
/* impulses of the last substep: normal impulses per contact */
int32_t  orc_last_normal_impulses(const orc_env* e, double* out, int32_t max_contacts);

/* CPU-baseline driver for bench.py: n_envs envs x (warmup + steps) batched env-steps of the
 * serpenoid-gait stream on n_threads threads, timed inside (no Python in the loop).  Returns seconds for the
 * `steps` timed batched steps; *substeps_out = physics substeps executed in them; agg4 (optional) = episode ends,
 * summed reward, summed per-step x displacement, summed contact counts over the timed steps. */
double   orc_bench_gait(const orc_params* p, int32_t n_envs, const double* phases,
                        const double* mu_plane_or_null, int32_t warmup, int32_t steps,
                        int32_t n_threads, int64_t* substeps_out, double* agg4_or_null);

#ifdef __cplusplus
}
#endif
#endif
