"""GPU parity tests: the HIP path (through the C ABI, host-buffer forms) against the CPU
oracle on the same seeded inputs.  Run with `-m gpu` on an MI355X.

Tolerances (float32 GPU vs float64 oracle, same algorithm, different formulation):
  single substep, free flight (no contacts):        |dq| <= 2e-6,  |dqd| <= 2e-4 * (1 + |qd|)
  single substep, on the ground (<= 64 contacts):   |dq| <= 2e-4,  |dqd| <= 2e-2 * (1 + |qd|)
The ground tolerance is float32 round-off through the 16-joint articulated chain and 50
unconverged Gauss-Seidel sweeps over ~200 rows: the oracle itself compiled in float32
(liboracle32.so) differs from the float64 oracle by 1.5e-4 / 1.2e-2 on the same inputs, and
the test also requires the GPU to be no worse than twice that.
"""
import numpy as np
import pytest

from conftest import f32_gate      # noqa: E402

from conftest import ROUND1, random_state

pytestmark = pytest.mark.gpu


def _mk(pkg, oracle_mod, n_envs, n=16, **over):
    st = pkg.Stepper(n_envs, device=0, n_modules=n, **over)
    orcs = [oracle_mod.OracleEnv(n_modules=n, **over) for _ in range(n_envs)]
    return st, orcs


def test_wave_primitives(pkg):
    lib = pkg.load()
    rc = lib.snk_selftest(0)
    assert rc == 0, lib.snk_last_error().decode()


def test_model_matches_oracle_rest_pose(pkg, oracle_mod):
    st = pkg.Stepper(1)
    bodies, origins = st.model_describe()
    e = oracle_mod.OracleEnv()
    inert = e.link_inertials()
    assert abs(bodies[:, 0].sum() - inert[:, 0].sum()) < 1e-12
    _, org = e.joint_axes_world()
    assert np.allclose(origins[1:], org, atol=1e-12)
    assert abs(st.mean_height()[0] - 0.026) < 1e-6


@pytest.mark.parametrize("case", ["air", "ground"])
def test_single_substep_parity(pkg, oracle_mod, case):
    n, B = 16, 64
    rng = np.random.default_rng(123 if case == "air" else 321)
    over = dict(residual_threshold=0.0)
    st, orcs = _mk(pkg, oracle_mod, B, **over)
    S = np.zeros((B, 13 + 2 * n))
    for i in range(B):
        if case == "air":
            S[i] = random_state(rng, n, z=1.0, qamp=0.5, vamp=0.5)
        else:
            S[i] = random_state(rng, n, z=0.026, qamp=0.3, vamp=0.3, flat=True)
            S[i, 9] *= 0.1
            S[i, 7:9] *= 0.1
    S32 = S.astype(np.float32)
    T = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32)
    st.set_state(S32)
    info = st.substep(T, 1)
    G, Gaux = st.get_state()
    err_q, err_v = 0.0, 0.0
    ev_gpu, ev_cal = [], []
    for i in range(B):
        o = orcs[i]
        o.set_state(S32[i].astype(np.float64))
        o.substep(T[i].astype(np.float64))
        ref = o.get_state()
        tau, fz, _ = o.get_aux()
        assert info[i, 1] == o.last_num_contacts
        assert info[i, 0] == o.last_iterations
        dq = np.abs(G[i, :13 + n] - ref[:13 + n])
        dv = np.abs(G[i, 13 + n:] - ref[13 + n:]) / (1.0 + np.abs(ref[13 + n:]))
        dw = np.abs(G[i, 7:13] - ref[7:13]) / (1.0 + np.abs(ref[7:13]))
        err_q = max(err_q, dq[:7].max(), dq[13:].max())
        err_v = max(err_v, dv.max(), dw.max())
        ev_gpu.append(max(dv.max(), dw.max()))
        # motor torque and joint-0 force sensor
        # (impulse/dt: a 240x amplification of the solver's float32 round-off)
        assert np.abs(Gaux[i, :n] - tau).max() < 2e-3 * (1.0 + np.abs(tau).max())
        assert abs(Gaux[i, n] - fz) < 5e-3 * (1 + abs(fz))
    # calibration: float32 build of the oracle against the float64 one, same inputs
    o32 = oracle_mod.OracleEnv(f32=True, n_modules=n, **over)
    cq, cv = 0.0, 0.0
    for i in range(B):
        o = orcs[i]
        o.set_state(S32[i].astype(np.float64))
        o32.set_state(S32[i].astype(np.float64))
        o.substep(T[i].astype(np.float64))
        o32.substep(T[i].astype(np.float64))
        ra, rb = o.get_state(), o32.get_state()
        cq = max(cq, np.abs(ra[:7] - rb[:7]).max(), np.abs(ra[13:13 + n] - rb[13:13 + n]).max())
        cvi = max((np.abs(ra[13 + n:] - rb[13 + n:]) / (1 + np.abs(ra[13 + n:]))).max(),
                  (np.abs(ra[7:13] - rb[7:13]) / (1 + np.abs(ra[7:13]))).max())
        ev_cal.append(cvi)
        cv = max(cv, cvi)
    print("case", case, "GPU-f32 vs oracle-f64: max |dq|", err_q, "max rel |dqd|", err_v,
          "| oracle-f32 vs oracle-f64:", cq, cv)
    # ground: the error is heavy-tailed (stick-slip states amplify float32 round-off by 1e5: the
    # float32 oracle reaches 1.2e-2 on these 64 states and 6.6 on one of 512 others), and every
    # rebuild re-associates FMAs.  So the criteria are distribution-based -- median and 90th
    # percentile within 3x of the float32 oracle's on the same states -- plus loose caps on the
    # maxima; tools/acc_distribution.py prints the same statistics for 512 states.
    tol_q, tol_v = (2e-6, 2e-4) if case == "air" else (5e-4, 1e-1)
    assert err_q < tol_q
    assert err_v < tol_v
    f32_gate("single substep [%s] rel velocity median of %d" % (case, B), np.median(ev_gpu), np.median(ev_cal), 1.5, 1e-6)
    f32_gate("single substep [%s] rel velocity p90" % case, np.percentile(ev_gpu, 90), np.percentile(ev_cal, 90), 2.0, 1e-5)
    f32_gate("single substep [%s] worst |dq|" % case, err_q, cq, 2.0, 1e-6)
    f32_gate("single substep [%s] worst rel velocity" % case, err_v, cv, 2.0, 1e-5)


def _substep_compare(pkg, oracle_mod, S, T, k=1, n=16, **over):
    B = S.shape[0]
    st = pkg.Stepper(B, n_modules=n, **over)
    S32 = S.astype(np.float32)
    st.set_state(S32)
    info = st.substep(T.astype(np.float32), k)
    G, Gaux = st.get_state()
    refs, its, ncs = [], [], []
    for i in range(B):
        o = oracle_mod.OracleEnv(n_modules=n, **over)
        o.set_state(S32[i].astype(np.float64))
        for _ in range(k):
            o.substep(T[i].astype(np.float32).astype(np.float64))
        refs.append(o.get_state()); its.append(o.last_iterations); ncs.append(o.last_num_contacts)
    return G, Gaux, info, np.array(refs), np.array(its), np.array(ncs)


def test_rare_branch_pyramid_friction(pkg, oracle_mod):
    """cone_friction=0: the two friction directions are resolved one after the other."""
    n, B = 16, 16
    rng = np.random.default_rng(77)
    S = np.array([random_state(rng, n, z=0.026, qamp=0.3, vamp=0.3, flat=True) for _ in range(B)])
    S[:, 7:9] *= 0.1; S[:, 9] *= 0.1
    T = rng.uniform(-0.5, 0.5, (B, n))
    G, _, info, R, its, ncs = _substep_compare(pkg, oracle_mod, S, T, cone_friction=0, residual_threshold=0.0)
    assert np.array_equal(info[:, 1], ncs) and np.array_equal(info[:, 0], its)
    # the box-clamped rows switch on/off with the normal impulse: individual states are far more
    # sensitive to float32 round-off than with the cone (the oracle built in float32 is off by
    # 2.8e-4 / 9.5e-4 on one of these 16 states, <4e-6 on most), so the bound is per state:
    # twice the float32 oracle's own error, with a floor
    S32 = S.astype(np.float32)
    for i in range(B):
        o = oracle_mod.OracleEnv(n_modules=n, f32=True, cone_friction=0, residual_threshold=0.0)
        o.set_state(S32[i].astype(np.float64))
        o.substep(T[i].astype(np.float32).astype(np.float64))
        r32 = o.get_state()
        cal_p, cal_q = np.abs(r32[:7] - R[i, :7]).max(), np.abs(r32[13:29] - R[i, 13:29]).max()
        assert np.abs(G[i, :7] - R[i, :7]).max() < min(max(5e-5, 2 * cal_p), 5e-4)
        assert np.abs(G[i, 13:29] - R[i, 13:29]).max() < min(max(2e-4, 2 * cal_q), 2e-3)
    assert (np.abs(G[:, 29:] - R[:, 29:]) / (1 + np.abs(R[:, 29:]))).max() < 5e-2


@pytest.mark.parametrize("n", [16, 32])
def test_one_friction_direction(pkg, oracle_mod, n):
    """friction_directions=1 (round 5): Bullet's multibody solver without
    SOLVER_USE_2_FRICTION_DIRECTIONS [U] -- one friction row per contact along the first btPlaneSpace1 tangent, bounds
    +-mu lambda_n, no cone branch.  One substep from ground states with lateral motion, GPU against the oracle under the
    same switch, calibrated per state against the float32 build of the oracle (box-bounded rows switch on and off with the
    normal impulse, as in the pyramid test above); and the switch does something: the result differs from the default's."""
    B = 16 if n == 16 else 8
    rng = np.random.default_rng(79)
    S = np.array([random_state(rng, n, z=0.026, qamp=0.3, vamp=0.3, flat=True) for _ in range(B)])
    S[:, 7:9] *= 0.1; S[:, 9] *= 0.1
    T = rng.uniform(-0.5, 0.5, (B, n))
    over = dict(friction_directions=1, residual_threshold=0.0, self_collision=0)
    G, _, info, R, its, ncs = _substep_compare(pkg, oracle_mod, S, T, n=n, **over)
    over["n_modules"] = n
    assert np.array_equal(info[:, 1], ncs) and np.array_equal(info[:, 0], its)
    S32 = S.astype(np.float32)
    moved = 0.0
    for i in range(B):
        o = oracle_mod.OracleEnv(f32=True, **over)
        o.set_state(S32[i].astype(np.float64))
        o.substep(T[i].astype(np.float32).astype(np.float64))
        r32 = o.get_state()
        cal_p, cal_q = np.abs(r32[:7] - R[i, :7]).max(), np.abs(r32[13:13 + n] - R[i, 13:13 + n]).max()
        assert np.abs(G[i, :7] - R[i, :7]).max() < min(max(5e-5, 2 * cal_p), 5e-4), i
        assert np.abs(G[i, 13:13 + n] - R[i, 13:13 + n]).max() < min(max(2e-4, 2 * cal_q), 2e-3), i
        d = oracle_mod.OracleEnv(**dict(over, friction_directions=2))
        d.set_state(S32[i].astype(np.float64))
        d.substep(T[i].astype(np.float32).astype(np.float64))
        moved = max(moved, np.abs(d.get_state()[7:13] - R[i, 7:13]).max())
    assert (np.abs(G[:, 13 + n:] - R[:, 13 + n:]) / (1 + np.abs(R[:, 13 + n:]))).max() < 5e-2
    assert moved > 1e-3          # base twist after one substep: the second tangent's friction is really gone


def test_rare_branch_joint_limits(pkg, oracle_mod):
    """Joints beyond +-1.57 create limit rows (kept in LDS, generic path)."""
    n, B = 16, 8
    rng = np.random.default_rng(78)
    S = np.array([random_state(rng, n, z=2.0, qamp=0.3, vamp=0.3) for _ in range(B)])
    for i in range(B):
        S[i, 13 + (2 * i) % n] = 1.6 + 0.01 * i          # above the upper limit
        S[i, 13 + (2 * i + 5) % n] = -1.62               # below the lower limit
        S[i, 13 + n + (2 * i) % n] = 0.5
    T = rng.uniform(-0.5, 0.5, (B, n))
    # finite motor force so that the limit rows visibly act against the motors
    G, _, info, R, its, ncs = _substep_compare(pkg, oracle_mod, S, T, max_motor_impulse=0.05, residual_threshold=0.0)
    # residual_threshold = 0 leaves the sweep only when every row's impulse change is exactly
    # zero: in float32 the clamped rows can get there (all later sweeps are then no-ops), the
    # float64 oracle keeps polishing at 1e-17 -- so the GPU may stop earlier, never later
    # (in the air no ground contacts; links folded past the joint limit touch their neighbours: URDF_USE_SELF_COLLISION,
    #  the same link-link contacts on both sides)
    assert np.all(info[:, 1] == ncs) and np.all(info[:, 0] <= its) and np.all(info[:, 0] >= 20)
    assert np.abs(G[:, 13:29] - R[:, 13:29]).max() < 2e-5
    assert (np.abs(G[:, 29:] - R[:, 29:]) / (1 + np.abs(R[:, 29:]))).max() < 2e-3


def test_rare_branch_early_exit(pkg, oracle_mod):
    """Bullet's residual early exit: at rest in the air with targets == q the first sweep already
    has a zero residual; the GPU must leave after the same number of iterations."""
    n, B = 16, 4
    S = np.zeros((B, 13 + 2 * n)); S[:, 2] = 3.0; S[:, 6] = 1.0
    rng = np.random.default_rng(79)
    S[:, 13:13 + n] = rng.uniform(-0.3, 0.3, (B, n))
    T = S[:, 13:13 + n].copy()
    G, _, info, R, its, ncs = _substep_compare(pkg, oracle_mod, S, T, joint_damping=0.0, lin_damping=0.0, ang_damping=0.0,
                                               gravity_z=0.0)
    assert np.all(its < 50) and np.array_equal(info[:, 0], its), (info[:, 0], its)
    assert np.abs(G - R).max() < 1e-5
    # and with gravity the motors must hold the pose: many iterations, same count on both sides
    G, _, info, R, its, ncs = _substep_compare(pkg, oracle_mod, S, T)
    assert np.array_equal(info[:, 0], its)


@pytest.mark.parametrize("n,model", [(16, "round1"), (32, "round1"), (16, "default"), (32, "default")])
def test_partial_contact_sets(pkg, oracle_mod, n, model):
    """Snake pitched so that only some cylinders are within the contact threshold: contact
    counts that are not multiples of the group sizes (8 rows per branch in the 16-link solve,
    4 contacts per loop trip in the 32-link one) and in both halves.  round1: the stateless two-point manifold at the
    absolute 0.02-m threshold; default: hulls + persistent manifold at the 1.2-mm relative threshold, three substeps
    from an empty cache (one new point per cylinder per step), shallower pitch angles."""
    states = []
    if model == "round1":
        over, k, z0 = dict(ROUND1), 1, 0.002
        angs = (0.02, 0.04, 0.08, 0.15, 0.3) if n == 16 else (0.01, 0.02, 0.04, 0.08, 0.15)
    else:
        over, k, z0 = dict(), 3, 0.0008
        angs = (0.0006, 0.0012, 0.0025, 0.005, 0.01) if n == 16 else (0.0003, 0.0006, 0.0012, 0.0025, 0.005)
    for ang in angs:
        s = np.zeros(13 + 2 * n)
        # pitch about y lifts the tail (the chain extends along -x): nose stays near the ground
        s[3:7] = [0, np.sin(ang / 2), 0, np.cos(ang / 2)]
        s[2] = z0
        states.append(s)
    S = np.array(states)
    T = np.zeros((len(states), n))
    G, _, info, R, its, ncs = _substep_compare(pkg, oracle_mod, S, T, k=k, n=n, residual_threshold=0.0, **over)
    assert np.array_equal(info[:, 1], ncs), (info[:, 1], ncs)
    assert len(set(ncs.tolist())) >= 3 and any(c % 8 for c in ncs) and any(c % 4 for c in ncs)   # really partial sets
    assert np.abs(G[:, :7] - R[:, :7]).max() < 2e-4 and np.abs(G[:, 13:13 + n] - R[:, 13:13 + n]).max() < 2e-4
    assert (np.abs(G[:, 13 + n:] - R[:, 13 + n:]) / (1 + np.abs(R[:, 13 + n:]))).max() < 2e-2


@pytest.mark.parametrize("n", [16, 32])
def test_substep_api_servo_converges(pkg, oracle_mod, n):
    """snk_substep_host from the rest pose with EVERY joint commanded, one substep per call: after ten substeps every joint
    is where the oracle's is (the position motors close 10 % of the error per substep), and ten substeps in one call end
    on the same bits.  (Round 4: a build of substep_kernel<16, true> lost delta-v's lanes 16..21 across the sensor pass --
    joints 10..15 stood still under this API while the fused kernels and every random-state parity test were fine.)"""
    st = pkg.Stepper(2, n_modules=n)
    st.reset()
    T = np.zeros((2, n), np.float32)
    T[0, :] = 0.2
    T[1, 1::2] = np.linspace(-0.5, 0.5, n // 2)
    for _ in range(10):
        st.substep(T, 1)
    S, _ = st.get_state()
    one = pkg.Stepper(2, n_modules=n)
    one.reset()
    one.substep(T, 10)
    S1, _ = one.get_state()
    assert np.array_equal(S, S1)
    for i in range(2):
        e = oracle_mod.OracleEnv(n_modules=n)
        e.reset()
        for _ in range(10):
            e.substep(T[i].astype(np.float64))
        q = e.get_state()[13:13 + n]
        assert np.abs(S[i, 13:13 + n] - q).max() < 2e-3, (i, S[i, 13:13 + n], q)
        assert np.abs(S[i, 13:13 + n] - T[i] * (1 - 0.9 ** 10)).max() < 2e-2      # every joint tracks, the last one too


@pytest.mark.parametrize("n,streamed", [(16, False), (16, True), (32, False)])
def test_three_kernels_one_substep(pkg, monkeypatch, n, streamed):
    """The same physics substep through the three instantiations that contain it -- the scheduled fused kernel, the
    unscheduled fused kernel (SNK_QUANTUM=0) and the single-substep API -- from 512 random ground states with every joint
    commanded: bit-identical states.  With max_counter = 0 a fused env-step is exactly one substep (snake.py:303's cap).
    Each instantiation is compiled on its own; a register copy placed inside a lane-dependent region of ONE of them loses
    lanes there and nowhere else (round 4's frozen joints), which only a cross-check like this one sees."""
    if streamed:
        monkeypatch.setenv("SNK_FORCE_STREAMED", "1")
    B = 512
    rng = np.random.default_rng(99)
    # snakes lying flat (pitch joints at zero and commanded to stay there, yaw joints bent and commanded at random): no
    # env-step ends in a termination, which the substep API would not act on
    S = np.stack([random_state(rng, n, z=0.026, qamp=0.3, vamp=0.3, flat=True) for _ in range(B)]).astype(np.float32)
    S[:, 0:3] = [0, 0, 0]
    S[:, 7:13] *= 0.1
    S[:, 13:13 + n:2] = 0
    S[:, 13 + n::2] = 0
    act = rng.uniform(-0.9, 0.9, (B, n)).astype(np.float32)       # gait 2: one action per joint
    act[:, 0::2] = 0
    over = dict(n_modules=n, gait=2, max_counter=0)
    outs = []
    for quantum in ("1", "0"):
        monkeypatch.setenv("SNK_QUANTUM", quantum)
        st = pkg.Stepper(B, **over)
        st.set_state(S)
        obs, rew, done, sub = st.step(act.copy(), vec_mode=False)
        assert np.all(sub == 1) and not done.any()
        outs.append((st.get_state(), st.get_manifold()))
        st.close()
    monkeypatch.setenv("SNK_QUANTUM", "1")
    st = pkg.Stepper(B, **over)
    st.set_state(S)
    st.substep(act * np.float32(st.params.scaling_factor), 1)
    api = (st.get_state(), st.get_manifold())
    st.close()
    (s0, x0), m0 = outs[0]
    for (s1, x1), m1 in (outs[1], api):
        assert np.array_equal(s0, s1)
        assert np.array_equal(x0[:, :n], x1[:, :n])               # the motor torques (prev_x is the env-step's business)
        assert np.array_equal(m0, m1)
    # ... and every joint moved towards its target (no lane of delta-v lost anywhere)
    yaw = slice(13 + 1, 13 + n, 2)
    moved = (s0[:, yaw] - S[:, yaw]) * np.sign(act[:, 1::2] * st.params.scaling_factor - S[:, yaw])
    # (per joint position in the chain: the tail's joints as much as the head's)
    assert (np.median(moved, axis=0) > 1e-3).all() and (moved > 0).mean() > 0.9, np.median(moved, axis=0)


@pytest.mark.parametrize("n,streamed,box", [(16, False, False), (16, True, False), (32, False, False), (16, True, True)])
def test_every_velocity_component_against_the_oracle(pkg, oracle_mod, monkeypatch, n, streamed, box):
    """the lane-loss class of round 4 -- a value the substep keeps in vector registers across the sensor
    pass coming back with some LANES stale -- shows as whole velocity COMPONENTS that stand still, which a maximum over
    all components of random states can hide behind a loose bound.  Here every component of the generalized velocity is
    compared on its own: base twist (6), every joint (n), and the free box's twist (6; the streamed-row kernels' lanes
    n + 6 .. n + 11), after one substep and after two in one launch (the first of which runs without the sensor pass when
    it cannot be an env-step's last), from 192 ground states with every joint commanded.  Per component: the GPU's median
    distance from the float64 oracle within 2 x the float32 oracle's (+ a floor), and the component MOVED."""
    if streamed:
        monkeypatch.setenv("SNK_FORCE_STREAMED", "1")
    B = 192
    rng = np.random.default_rng(1234 + n + 7 * streamed + 13 * box)
    over = dict(n_modules=n, residual_threshold=0.0)
    if box:
        over.update(obstacle=2, obstacle_pos=[0.16, 0.0, 0.1])
    # snakes on the ground with a few degrees of pitch and roll (a cylinder lying exactly flat has equally deep rim
    # vertices: which one becomes the contact point is then a last-bit decision, tests/test_gpu_contact_models.py)
    from test_gpu_contact_models import _ground_states
    S = _ground_states(B, n, seed=77 + n)
    S[:, 0:2] = 0
    T = rng.uniform(-0.6, 0.6, (B, n)).astype(np.float32)
    for K in (1, 2):
        st = pkg.Stepper(B, **over)
        st.set_state(S)
        if box:
            b0, _ = st.get_box()
            b0[:, 7:13] = rng.uniform(-0.05, 0.05, (B, 6)).astype(np.float32)      # a box already in motion: its six lanes carry numbers
            st.set_box(b0)
        st.substep(T, K)
        G, _ = st.get_state()
        GB = st.get_box()[0] if box else None
        st.close()
        nv = 6 + n + (6 if box else 0)
        eg, ec, mv = np.zeros((B, nv)), np.zeros((B, nv)), np.zeros((B, nv))
        for i in range(B):
            outs = []
            for f32 in (False, True):
                e = oracle_mod.OracleEnv(f32=f32, **over)
                e.set_state(S[i].astype(np.float64))
                if box:
                    e.set_box(b0[i].astype(np.float64))
                for _ in range(K):
                    e.substep(T[i].astype(np.float64))
                v = np.concatenate([e.get_state()[7:13], e.get_state()[13 + n:]] + ([e.get_box()[0][7:13]] if box else []))
                outs.append(v)
            g = np.concatenate([G[i, 7:13], G[i, 13 + n:]] + ([GB[i, 7:13]] if box else []))
            v0 = np.concatenate([S[i, 7:13], S[i, 13 + n:]] + ([b0[i, 7:13]] if box else []))
            sc = 1.0 + np.abs(outs[0])
            eg[i], ec[i], mv[i] = np.abs(g - outs[0]) / sc, np.abs(outs[1] - outs[0]) / sc, np.abs(outs[0] - v0)
        mg, mc, mm = np.median(eg, axis=0), np.median(ec, axis=0), np.median(mv, axis=0)
        worst = int(np.argmax(mg / np.maximum(mc, 1e-6)))
        print("n %d streamed %s box %s K %d: per-component medians, worst ratio at component %d: GPU %.2e float32 oracle %.2e; "
              "smallest median motion %.2e" % (n, streamed, box, K, worst, mg[worst], mc[worst], mm.min()))
        # every component moved in the oracle (so a component that stood still on the GPU cannot hide) ...
        assert (mm[6:6 + n] > 1e-2).all() and (mm > 1e-5).all(), mm
        # ... and every component on its own is where float32 arithmetic puts it
        assert (mg <= 2.0 * mc + 2e-5).all(), (np.nonzero(mg > 2.0 * mc + 2e-5)[0], mg, mc)
