/*
 * snk.h -- C ABI of the MI355X batched snake stepper (libsnk.so).
 *
 * Drop-in boundary for ONE path of vinits5/bullet-envs: SnakeGymEnv.step()/reset(),
 * whose arithmetic the reference delegates to pybullet.stepSimulation on the CPU.
 * Everything here is plain C: pointers, sizes, int status codes; no torch types.
 * Paths cited below are under /root/reference.
 *
 * One handle = N independent environments resident on one GPU.  Calls on a handle
 * are serialised by the caller (the reference is single-threaded per env process,
 * multiprocessing_env.py:7-29).  Functions return 0 on success, non-zero on error;
 * snk_last_error() gives the message.  No exceptions cross this boundary.
 */
#ifndef SNK_H
#define SNK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Every knob the reference path depends on.  [U] = PyBullet/Bullet default taken from
 * knowledge of bullet3 (not verifiable here). */
#define SNK_ABI_VERSION 6       /* bumped whenever snk_params' layout or an entry point's meaning changes */

typedef struct snk_params {
    /* layout guard (round 6): snk_default_params fills both, snk_create refuses a struct whose size or
       version is not this library's -- a caller compiled against another header gets an error message, not garbage
       parameters.  New fields are appended at the END of the struct from here on. */
    uint32_t struct_size;       /* sizeof(snk_params) of the header the caller was compiled against */
    uint32_t abi_version;       /* SNK_ABI_VERSION of that header                                    */
    /* model: snake/snake.urdf (constants generated parametrically, not parsed) */
    int32_t n_modules;          /* 16 = snake.urdf; 32 = BASELINE config 4 (only these two)      */
    int32_t inertia_from_file;  /* 0: inertia from collision AABB [U] (snake.py:93 passes no
                                   URDF_USE_INERTIA_FROM_FILE); 1: urdf:815,871 values       */
    double  default_mass;       /* links without <inertial> (urdf:7,14,818): mass 1 [U]       */
    double  collision_margin;   /* 0.001 [U]                                                  */
    int32_t hull_sides;         /* 32 (default): the 32-gon prism PyBullet builds for a URDF <cylinder> unless
                                   URDF_USE_IMPLICIT_CYLINDER [U] (snake.py:93 passes no such flag); 0: implicit
                                   cylinder (the round-1 model); 0 or 3..32                           */
    int32_t contact_model;      /* 1 (default): Bullet's persistent manifold [U] -- one new support point per
                                   cylinder per step merged into a cache of <= 4, refreshed / dropped at the
                                   breaking threshold; 0: stateless -- both end-cap points of
                                   every cylinder, every step (the round-1 model)                      */
    int32_t self_collision;     /* 1 (default): link-link contacts between non-adjacent cylinder links, what
                                   URDF_USE_SELF_COLLISION (snake.py:93) switches on [U].  Their two-body rows are
                                   built by the streamed-row solve; a 16-link substep in which some pair of links may
                                   be within the breaking threshold goes through it (never inside the reference's
                                   command range)                                              */
    int32_t obstacle;           /* the box of snake/block.urdf that Snake.add_obstacle (snake.py:83-84, commented out
                                   at :94) and snake_gait_test.py:51 put in front of the snake.  0 (default): none.
                                   1: STATIC (16 links: up to 8 contacts with it, out of the solve's 64 slots).
                                   2: as the reference loads it (useFixedBase=0): a FREE body of obstacle_mass resting
                                   on the ground -- its own six velocity components, gravity, damping, a persistent
                                   manifold with the plane, two-body rows with the snake's links; 16 links only, on
                                   the streamed-row kernels                                       */
    double  obstacle_pos[3];    /* centre of the box: [2, 0, 0.1] (snake.py:94, snake_gait_test.py:51)         */
    double  obstacle_half[3];   /* half extents: [0.1, 0.4, 0.1] (snake/block.urdf:16)                         */
    double  mu_obstacle;        /* 0.5 [U]: Bullet's default lateral friction for a link without <contact>     */
    double  obstacle_mass;      /* 200 (snake/block.urdf:6); inertia from the box shape [U] (inertia_from_file 1:
                                   block.urdf:7's 1, 100, 1)                                                   */
    /* pybullet world */
    double  dt;                 /* 1/240 [U]: setTimeSteps is never called (snake.py:271-272) */
    double  gravity_z;          /* snake.py:8,91   -9.8                                       */
    double  lin_damping;        /* 0.04 [U]                                                   */
    double  ang_damping;        /* 0.04 [U]                                                   */
    double  joint_damping;      /* snake.urdf:838  0.1                                        */
    double  max_coord_vel;      /* 100 [U]                                                    */
    /* setJointMotorControlArray(POSITION_CONTROL) defaults, snake.py:219-221 */
    double  kp;                 /* 0.1 [U]                                                    */
    double  kd;                 /* 1.0 [U]                                                    */
    double  max_motor_impulse;  /* forces=[inf] (snake.py:26-27) -> +inf                      */
    double  joint_lo, joint_hi; /* snake.urdf:839  -1.57, 1.57                                */
    double  limit_erp;          /* 0.2 [U]                                                    */
    double  limit_max_impulse;  /* 100 [U]                                                    */
    /* changeDynamics, snake.py:103-107 */
    double  mu_link;            /* lateralFriction = 2                                        */
    double  aniso[3];           /* anisotropicFriction = [1, 0.1, 0.01] (snake.py:25)         */
    double  contact_erp;        /* 0.08 [U]                                                   */
    double  linear_slop;        /* 1e-5 [U]                                                   */
    double  breaking_threshold; /* 0.02 [U] gContactBreakingThreshold                         */
    int32_t relative_breaking_threshold; /* 1 (default) [U]: btCollisionDispatcher's default flag
                                   CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD -- a manifold's threshold is
                                   breaking_threshold x the smaller angular-motion disc of its two shapes, computed
                                   from the link collider's AABB (0.0603 m -> 1.206 mm); 0: breaking_threshold itself */
    int32_t cone_friction;      /* 1 [U]                                                      */
    int32_t n_iterations;       /* 50 [U]                                                     */
    double  residual_threshold; /* 1e-7 [U]; 0 disables the early exit                        */
    int32_t warm_start;         /* 0 (default) [U]: btMultiBodyConstraintSolver::setupMultiBodyContactConstraint has its
                                   warm start disabled, every row starts at zero impulse.  1: SOLVER_USE_WARMSTARTING as
                                   the rigid-body solver does it -- a cached point's normal row starts at
                                   warmstarting_factor x the impulse it carried in the last substep (the contact cache
                                   keeps it: btManifoldPoint::m_appliedImpulse), delta-v at the sum of M^-1 J^T of those;
                                   friction rows start at zero.  Needs contact_model 1                            */
    double  warmstarting_factor;/* 0.85 [U] btContactSolverInfo::m_warmstartingFactor            */
    int32_t friction_directions;/* 2 (default) [U]: SOLVER_USE_2_FRICTION_DIRECTIONS -- both btPlaneSpace1 tangents of a
                                   contact get a friction row (with cone_friction: the implicit cone over the pair).
                                   1: Bullet's multibody solver WITHOUT that flag -- one row per contact, along the first
                                   tangent ((0,-1,0) for the ground's normal), bounds +-mu lambda_n.  A switch since
                                   round 5 because the oracle prices it above the error bar's other rows (+14 % forward
                                   motion under the bench gait: profiles/r05_u_rows.json); the default is the reading
                                   under which a body pushed along world x comes to rest, as it does in PyBullet      */
    /* Snake / SnakeGymEnv */
    double  scaling_factor;     /* snake.py:63   pi/6                                         */
    int32_t gait;               /* snake.py:62   1: actions drive odd motor slots             */
    double  servo_tol;          /* snake.py:232  0.05                                         */
    int32_t max_counter;        /* snake.py:303  40  (=> at most 41 substeps)                 */
    double  height_threshold;   /* snake.py:238  0.1                                          */
    double  energy_dt;          /* snake.py:9,339  1/100                                      */
    double  alpha, beta, gamma; /* SnakeGymEnv.py:14-16  1, 0.01, 0.1                         */
    double  term_angle;         /* SnakeGymEnv.py:100  0.5                                    */
    int32_t term_index;         /* SnakeGymEnv.py:100  obs index 9                            */
    double  collision_force;    /* SnakeGymEnv.py:94   10                                     */
    double  collision_penalty;  /* SnakeGymEnv.py:94   -10                                    */
    double  done_penalty;       /* SnakeGymEnv.py:40   -5                                     */
    /* (appended in round 6: new fields go to the END of the struct) */
    int32_t contact_order;      /* 0 (default): the solver sweeps the ground manifolds in link order.  [U]: Bullet hands its
                                   solver the manifolds in the island manager's order, which is not knowable here, and
                                   with 50 unconverged sweeps the order is part of the answer -- the oracle prices other
                                   orders at -3 % .. -24 % of forward motion under the bench gait, the largest entry of
                                   the error bar (profiles/r06_u_rows.json), hence a switch: 1 = link order reversed;
                                   2 = link order after the island manager's unstable quickSort on equal island ids (the
                                   one candidate that can be restated: btAlignedObjectArray::quickSort over the 2n plane-link
                                   manifolds); k >= 3 = the fixed permutation of the cylinder links that sorting by a hash
                                   of (k, link) gives (the same every substep, as a list of persistent manifolds keeps its
                                   order).  A
                                   cylinder's <= 4 points stay together and in their own order.  Needs contact_model 1 */
    int32_t reserved0;          /* 0 */
} snk_params;

typedef struct snk_handle snk_handle;

/* Fills the reference's defaults (Snake.defaultParams snake.py:55-63, SnakeGymEnv.py:13-17). */
void snk_default_params(snk_params* p);

/* Replaces: Snake.__init__ + SnakeGymEnv.__init__ -> robot.reset(hardReset=True)
 * (snake.py:14-32,88-95; SnakeGymEnv.py:5-26) for n_envs worlds at once, and the N worker
 * processes of SubprocVecEnv.__init__ (ppo/multiprocessing_env.py:97-117).
 * device = HIP device ordinal.  State after create = hard reset. */
int snk_create(const snk_params* p, int32_t n_envs, int32_t device, snk_handle** out);
int snk_destroy(snk_handle* h);

int32_t snk_num_envs(const snk_handle* h);
int32_t snk_obs_dim(const snk_handle* h);    /* 3n+8  (snake.py:163-164)            */
int32_t snk_act_dim(const snk_handle* h);    /* n/2 for gait 0/1 (SnakeGymEnv.py:74-76) */
int32_t snk_state_dim(const snk_handle* h);  /* 13+2n: pos3 quat_xyzw4 omega3 vel3 q qd */
int32_t snk_record_floats(const snk_handle* h); /* floats per env in the HBM state record */

/* Replaces SnakeGymEnv.reset() (SnakeGymEnv.py:28-31 -> snake.py:96-99,119-127) for every
 * env whose mask byte is non-zero (mask == NULL: all).  obs_dev [n_envs x obs_dim] f32,
 * device pointer; rows of unmasked envs are left untouched.  stream = hipStream_t or NULL. */
int snk_reset(snk_handle* h, const uint8_t* mask_dev, float* obs_dev, void* stream);

/* Replaces SubprocVecEnv.step (ppo/multiprocessing_env.py:119-128) = per env
 * SnakeGymEnv.step (SnakeGymEnv.py:33-50) -> Snake.step (snake.py:274-306) with its
 * 0..41 stepSimulation calls, reward, termination, -5 and auto-reset.
 *   actions_dev  [n_envs x act_dim] f32, device; clipped to [-1,1] IN PLACE like
 *                checkBound (SnakeGymEnv.py:82-88)
 *   obs_dev      [n_envs x obs_dim] f32   rew_dev [n_envs] f32   done_dev [n_envs] u8
 *   substeps_dev [n_envs] i32 (may be NULL): Snake.counter
 *   vec_mode     1: worker semantics, obs of a done env is the post-reset one
 *                (multiprocessing_env.py:13-15); 0: SnakeGymEnv.step semantics, terminal obs
 * Asynchronous on `stream`; results are ready after the stream is synchronised.
 * Internally two launches: a one-block plan kernel and the step kernel, whose resident waves share
 * the env-steps through an in-launch queue.  Results do not depend on that schedule.
 * Every wait inside the kernel is bounded; if one ever runs out the kernel drains, and this and
 * every later call on the handle return non-zero ("env-step scheduler: ...").  Environment:
 * SNK_QUANTUM=<substeps per slice> (default 1; 0 = the unscheduled kernel), SNK_FORCE_STREAMED=1 (16 links on the
 * streamed-row kernels of the 32-link chain: diagnostics and cross-checks), both read by snk_create. */
int snk_step(snk_handle* h, float* actions_dev, float* obs_dev, float* rew_dev,
             uint8_t* done_dev, int32_t* substeps_dev, int32_t vec_mode, void* stream);

/* The same step with ONE output buffer: packed_dev [n_envs x row_stride] f32 (row_stride >= obs_dim + 2), row e =
 * [obs of env e (obs_dim floats) | reward (f32) | done (u32: 0 or 1) | untouched padding].  This is the block a rank of a
 * sharded vector env sends to the trainer rank -- the reference's workers send (ob, reward, done, info) per env through
 * their Pipe (ppo/multiprocessing_env.py:11-16, 125-128) -- written by the step kernel itself, so that no copy kernel
 * stands between the physics and the gather (bullet-envs_amd/device_env.py: ShardedVecEnv). */
int snk_step_packed(snk_handle* h, float* actions_dev, float* packed_dev, int32_t row_stride,
                    int32_t* substeps_dev, int32_t vec_mode, void* stream);

/* Host-buffer convenience forms (upload, run, download, synchronise). */
int snk_reset_host(snk_handle* h, const uint8_t* mask, float* obs);
int snk_step_host(snk_handle* h, float* actions, float* obs, float* rew, uint8_t* done,
                  int32_t* substeps, int32_t vec_mode);

/* Replaces pybullet.stepSimulation (snake.py:286) preceded by setJointMotorControlArray
 * (snake.py:221): k physics substeps with motor targets [n_envs x n] (radians), host buffer.
 * info (may be NULL) [n_envs x 2] i32: solver iterations and contact count of the last substep.
 * For single-substep parity tests. */
int snk_substep_host(snk_handle* h, const float* targets, int32_t k, int32_t* info);

/* State access (host buffers), for parity tests and checkpointing.  These calls (and snk_get_obs,
 * snk_mean_height, snk_link_positions, snk_set/get_ground_friction) synchronise the device first, so a step
 * still running on any stream is complete before the records are read or overwritten.
 * state [n_envs x state_dim]; aux [n_envs x (n+2)] = motor torques n, joint-0 Fz, prev obs x. */
int snk_get_state(snk_handle* h, float* state, float* aux);
int snk_set_state(snk_handle* h, const float* state, const float* aux);
/* contact_model 1 only: the persistent contact manifolds (part of the simulator state, like Bullet's contact cache,
 * which a soft reset does not clear [U]).  Host buffers [n_envs x 2n x 29]: per cylinder (in link order)
 * [count, 4 x (point on the link in link coordinates 3, point on the ground in world coordinates 3, the normal impulse
 * the point carried in the last substep)].
 * snk_manifold_floats = 2n * 29, or 0 for a contact_model 0 handle or a null handle (then get/set fail). */
int32_t snk_manifold_floats(const snk_handle* h);
int snk_get_manifold(snk_handle* h, float* out);
int snk_set_manifold(snk_handle* h, const float* in);
/* getObservation (snake.py:209-217) of every env, host buffer [n_envs x obs_dim]. */
int snk_get_obs(snk_handle* h, float* obs);
/* checkSnakeHeight's mean z (snake.py:237-245), host buffer [n_envs]. */
int snk_mean_height(snk_handle* h, float* out);
/* Snake.getLinkPositions (snake.py:138-146), the test-mode telemetry of SnakeGymEnv.step's info
 * (SnakeGymEnv.py:43-44): world COM of Bullet links 0,3,...,3n of every env, host buffer
 * [n_envs x 3(n+1)] laid out [x_0..x_n, y_0..y_n, z_0..z_n]. */
int snk_link_positions(snk_handle* h, float* out);

/* Reaction force through the first motor joint (Bullet joint 3: INPUT_IF_1 -> OUTPUT_BODY_1), z component in the
 * child link's frame, of every env's last physics substep: what snake_gait_test.py:33-40,126 reads
 * (getJointState(robot, 3)[2][2], "> 20: the snake has hit the wall").  Host buffer [n_envs].  Like the joint-0 force
 * of the observation it is evaluated on the last substep of an env-step (and by every snk_substep_host substep). */
int snk_joint3_reaction_fz(snk_handle* h, float* out);

/* obstacle 2 only: the free box of every env (part of the simulator state; a soft reset leaves it where it is, like
 * resetBasePositionAndOrientation of the SNAKE, snake.py:126-127).  Host buffers: state [n_envs x 13] = pos3,
 * quat xyzw 4, omega_world 3, vel_world 3; manifold [n_envs x 29] = its contact cache with the plane, [count, 4 x
 * (point on the box in box coordinates 3, point on the ground 3, applied normal impulse)].  Either may be NULL. */
int snk_get_box(snk_handle* h, float* state, float* manifold);
int snk_set_box(snk_handle* h, const float* state, const float* manifold);

/* Where this build's structural limits were met, counted on the device since snk_create (Bullet has no such limits;
 * DESIGN.md 3).  Ground contacts have none left: the streamed-row solve has a slot for every point its chain's manifolds
 * can hold (8n), and the register-resident 16-link solve hands the substeps that outgrow its 64 slots to it.
 *   out[0] 16-link handles on the register-resident solve: physics substeps in which an environment held more contact
 *          points than that solve's 64 slots (a snake at rest gathers up to four per cylinder), touched the
 *          obstacle with more than eight cylinders, or had two of its links within reach of each other (that
 *          solve has no two-body rows).  Those substeps are
 *          solved by the streamed-row solve of the same chain instead, in the same launch, with every point -- a
 *          count of slower substeps, not of lost contacts.  Streamed-row handles: always 0,
 *   out[1] manifold points that got no rows: always 0 (kept as a tripwire: the finders still count against the slots),
 *   out[2] link-link / obstacle contacts beyond the room for them (32 in all; obstacle contacts are kept before
 *          link-link ones): a 32-link chain can reach it; a 16-link one only when tangled beyond its joint limits with
 *          more than 32 pairs of links touching at once (states set from outside: tests/test_gpu_accuracy_distribution.py
 *          meets 6 in 1024 random foldings) -- a substep with more than the register-resident solve's eight box contacts
 *          goes through the other solve, which has room for every cylinder.
 * Host buffer of 3. */
int snk_contact_overflow(snk_handle* h, uint64_t* out);

/* How many contact points the environments held, physics substep by physics substep, since snk_create (or the last
 * reset of these counters): out[k] = substeps that ran with k contact points (ground + link-link + obstacle), the last
 * bin collecting everything beyond it.  snk_contact_histogram_bins() values (host buffer).  This is the distribution
 * that sizes the register-resident solve's row slots; counted by the step and the substep kernels alike
 * -- while enabled (snk_contact_histogram_enable).  reset != 0 zeroes the counters after reading. */
int32_t snk_contact_histogram_bins(void);
int snk_contact_histogram(snk_handle* h, uint64_t* out, int32_t reset);
/* The counting is OFF after snk_create (it costs one atomic per physics substep: 0.7 % of the headline rate) and is
 * switched with this call; the counters keep what they hold. */
int snk_contact_histogram_enable(snk_handle* h, int32_t on);

/* BASELINE config 5: per-env lateral friction of the ground plane (reference: plane.urdf = 1). */
int snk_set_ground_friction(snk_handle* h, const float* mu /* host [n_envs] */);
int snk_get_ground_friction(snk_handle* h, float* mu /* host [n_envs] */);

/* Test hook: sets the step queue's ticket counters (in-launch scheduling) to `base`, so that a test
 * can put the wrap-around of the 32-bit tickets inside its next step.  Results never depend on it. */
int snk_debug_set_tickets(snk_handle* h, uint32_t base);

/* Test hook for the failure path: raises this handle's alarm from the HOST -- the host-mapped word a wave of the step
 * kernel sets when one of its bounded waits runs out (nothing waits, nothing hangs).  Afterwards the handle behaves as
 * after a real alarm: snk_step / snk_step_packed / snk_step_host / snk_reset(_host) / snk_substep_host and the state
 * accessors (get/set state, manifold, box, obs, mean height, link positions, joint-3 force, set_ground_friction) return
 * non-zero with snk_last_error() = "env-step scheduler: a bounded wait ran out ..."; snk_destroy succeeds.  The
 * reference's failure story is SubprocVecEnv.close() draining and joining its workers
 * (ppo/multiprocessing_env.py:140-150): here too the only way on is to destroy the handle and create a new one. */
int snk_debug_raise_alarm(snk_handle* h);

/* Device-side self test of the wave primitives (DPP reductions); 0 = pass. */
int snk_selftest(int32_t device);

/* Dominant-kernel timing with HIP events on the launch stream: enable(capacity) arms a pool
 * of `capacity` event pairs (0 disarms); each following snk_step launch is bracketed by one
 * pair, without any host synchronisation; read() waits for them, returns the mean launch
 * duration in ms and how many launches it covers, and re-arms the pool. */
int snk_timing_enable(snk_handle* h, int32_t capacity);
int snk_timing_read(snk_handle* h, double* mean_ms, int32_t* count);

/* What the kernels derive from a parameter set, without a device (host arithmetic only), for tests: out[6] =
 * [contact breaking threshold of a (ground | link | box, link) manifold in metres (relative_breaking_threshold 1:
 *  breaking_threshold x the link collider's angular-motion disc), collision cylinder radius, half length, centre along
 *  the link's z (snake.urdf:806-811), collision margin, breaking threshold of the (ground, box) manifold]. */
int snk_params_derived(const snk_params* p, double* out);

/* Merged-model introspection for tests: per composite body [mass, com3, I_origin6] and
 * rest-pose world origins.  out_bodies [(n+1) x 10], out_origins [(n+1) x 3] (host). */
int snk_model_describe(const snk_handle* h, double* out_bodies, double* out_origins);

const char* snk_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
