"""CPU tests that pin the ORACLE: known answers from the URDF
constants, an independent numpy formulation, and physics invariants.  PyBullet itself is
not available, so parity with PyBullet stays unpinned."""
import json
import os

import numpy as np
import pytest

import np_model
from conftest import random_state

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "appendix_b.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_fk_known_answers(oracle_mod, gold):
    e = oracle_mod.OracleEnv()
    assert e.L == gold["num_links_with_root"]
    com = e.link_com_world()
    # my link index = Bullet link index + 1
    assert np.allclose(com[0 + 1], gold["base_link_com_rest"], atol=1e-12)
    for k in range(1, 17):
        assert np.allclose(com[3 * k + 1], gold["output_body_com_rest"][k - 1], atol=1e-9)
        # INPUT_IF k COM sits on the revolute pivot (Appendix B)
        assert np.allclose(com[3 * k - 2 + 1], gold["output_body_com_rest"][k - 1], atol=1e-9)
    ax, org = e.joint_axes_world()
    assert np.allclose(ax, gold["joint_axes_rest"], atol=1e-9)
    assert np.allclose(org, gold["output_body_com_rest"], atol=1e-9)
    assert abs(e.mean_height() - gold["rest_mean_height"]) < 1e-9
    inert = e.link_inertials()
    assert abs(inert[:, 0].sum() - gold["total_mass_bullet_rule"]) < 1e-12
    assert (inert[:, 0] == 1.0).sum() == gold["num_links_without_inertial"]
    # DFS order: revolute child links are Bullet links 3,6,...,48
    par = e.link_parents()
    rev_links = [3 * k + 1 for k in range(1, 17)]
    for k, i in enumerate(rev_links):
        assert par[i] == i - 2          # OUTPUT_BODY's parent is INPUT_IF (collar sits between in DFS order)
    assert [i - 1 for i in rev_links] == gold["motor_joint_indices"]


@pytest.mark.parametrize("n", [1, 2, 16, 32])
def test_minv_against_numpy_mass_matrix(oracle_mod, n):
    rng = np.random.default_rng(n)
    e = oracle_mod.OracleEnv(n_modules=n)
    links = np_model.build_tree(n)
    for _ in range(3):
        s = random_state(rng, n, z=2.0)
        e.set_state(s)
        M = np_model.mass_matrix(links, s[0:3], s[3:7], s[13:13 + n])
        assert np.allclose(M, M.T, atol=1e-12)
        for _ in range(3):
            x = rng.normal(size=6 + n)
            y = e.minv_mul(M @ x)
            assert np.allclose(y, x, rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("n", [2, 16, 32])
def test_forward_dynamics_zero_velocity(oracle_mod, n):
    """a = M^-1 (tau + gravity) at rest."""
    rng = np.random.default_rng(10 + n)
    e = oracle_mod.OracleEnv(n_modules=n)
    links = np_model.build_tree(n)
    s = random_state(rng, n, z=2.0, vamp=0.0)
    e.set_state(s)
    tau = rng.normal(size=n) * 0.1
    acc = e.forward_dynamics(tau, gravity=True, damping=False)
    M = np_model.mass_matrix(links, s[0:3], s[3:7], s[13:13 + n])
    f = np_model.gravity_force(links, s[0:3], s[3:7], s[13:13 + n])
    f[6:] += tau
    ref = np.linalg.solve(M, f)
    assert np.allclose(acc, ref, rtol=1e-7, atol=1e-7)


def test_free_fall_com_acceleration(oracle_mod):
    """With gravity only, d(momentum)/dt = M_total g whatever the internal motion."""
    n = 16
    rng = np.random.default_rng(5)
    e = oracle_mod.OracleEnv()
    links = np_model.build_tree(n)
    s = random_state(rng, n, z=3.0)
    e.set_state(s)
    acc = e.forward_dynamics(np.zeros(n), gravity=True, damping=False)
    h = 1e-6
    g0 = np.concatenate([s[7:13], s[13 + n:]])
    P0, L0, _ = np_model.momentum(links, s[0:3], s[3:7], s[13:13 + n], g0)
    # advance the configuration and velocity by h along the flow
    s1 = s.copy()
    s1[0:3] += h * s[10:13]
    w = s[7:10]
    dq = np.concatenate([0.5 * h * w, [1.0]])
    q0 = s[3:7]
    # quaternion product dq * q0 (xyzw)
    x1, y1, z1, w1 = dq
    x2, y2, z2, w2 = q0
    s1[3:7] = [w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
               w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2]
    s1[13:13 + n] += h * s[13 + n:]
    g1 = g0 + h * acc
    P1, L1, _ = np_model.momentum(links, s1[0:3], s1[3:7], s1[13:13 + n], g1)
    mtot = sum(k["m"] for k in links)
    assert np.allclose((P1 - P0) / h, [0, 0, -9.8 * mtot], atol=5e-3)
    # angular momentum about the origin changes by the gravity torque  r_com x M g
    com = P0 * 0
    Rw, ow = np_model.fk(links, s[0:3], s[3:7], s[13:13 + n])
    for i, k in enumerate(links):
        com += k["m"] * (ow[i] + Rw[i] @ k["c"])
    com /= mtot
    assert np.allclose((L1 - L0) / h, np.cross(com, [0, 0, -9.8 * mtot]), atol=5e-3)


def test_momentum_matches_numpy(oracle_mod):
    n = 16
    rng = np.random.default_rng(6)
    e = oracle_mod.OracleEnv()
    links = np_model.build_tree(n)
    s = random_state(rng, n, z=1.0)
    e.set_state(s)
    lin, ang, K = e.momentum()
    g0 = np.concatenate([s[7:13], s[13 + n:]])
    P, L, K2 = np_model.momentum(links, s[0:3], s[3:7], s[13:13 + n], g0)
    assert np.allclose(lin, P, atol=1e-10) and np.allclose(ang, L, atol=1e-10) and abs(K - K2) < 1e-10


def test_zero_gravity_conservation(oracle_mod):
    """No gravity, no damping, motors inert, far from the plane: momentum is conserved and
    the semi-implicit Euler energy drift shrinks with dt."""
    n = 16
    rng = np.random.default_rng(7)
    s0 = random_state(rng, n, z=50.0, vamp=0.5)
    drift = []
    for dt in (1e-3, 5e-4):
        e = oracle_mod.OracleEnv(dt=dt, gravity_z=0.0, lin_damping=0.0, ang_damping=0.0, joint_damping=0.0,
                                 kp=0.0, kd=0.0)
        e.set_state(s0)
        P0, L0, K0 = e.momentum()
        for _ in range(int(round(0.05 / dt))):
            e.substep(np.zeros(n))
            assert e.last_num_contacts == 0
        P1, L1, K1 = e.momentum()
        drift.append((np.abs(P1 - P0).max(), np.abs(L1 - L0).max(), abs(K1 - K0)))
    scale = np.abs(L0).max()
    assert drift[1][0] < 1e-3 * np.abs(P0).max() + 1e-6
    assert drift[1][1] < 1e-2 * scale
    # first-order integrator: halving dt roughly halves the drift
    assert drift[1][2] < 0.7 * drift[0][2] + 1e-9


def test_motor_row_free_space(oracle_mod):
    """After one substep the motor row holds  qd+ = kp (q* - q)/dt + (1-kd) qd  (kd = 1)."""
    n = 16
    rng = np.random.default_rng(8)
    e = oracle_mod.OracleEnv(residual_threshold=0.0, n_iterations=5000)   # PGS over the chain converges slowly
    s = random_state(rng, n, z=5.0, vamp=0.2)
    e.set_state(s)
    targets = rng.uniform(-0.5, 0.5, n)
    e.substep(targets)
    s1 = e.get_state()
    want = 0.1 * (targets - s[13:13 + n]) * 240.0
    assert np.allclose(s1[13 + n:], want, atol=1e-9)
    # applied motor torque = impulse/dt is what getJointState()[3] reports
    tau, _, _ = e.get_aux()
    assert np.all(np.isfinite(tau)) and np.abs(tau).max() > 0


@pytest.mark.parametrize("model", ["default", "round1"])
def test_rest_on_plane_supports_weight(oracle_mod, model):
    """default: hulls + persistent manifolds at the relative threshold -- a resting snake ends with all four cached
    points of every cylinder (vertices 5 mm apart no longer merge at a 1.2-mm threshold) and the 50 unconverged sweeps
    over 128 contacts leave a jitter of a few tenths of a newton: the MEAN over the last 60 substeps carries the
    weight.  round1: two end-cap points per cylinder."""
    over = dict(hull_sides=0, contact_model=0, relative_breaking_threshold=0) if model == "round1" else {}
    e = oracle_mod.OracleEnv(**over)
    totals = []
    for k in range(480):
        e.substep(np.zeros(16))
        if k >= 420:
            totals.append(e.last_normal_impulses().sum() * 240.0)
    assert abs(np.mean(totals) - 21.296 * 9.8) < 0.5
    assert np.abs(np.array(totals) - 21.296 * 9.8).max() < (0.5 if model == "round1" else 2.0)
    s = e.get_state()
    assert np.abs(s[7:13]).max() < 1e-2 and np.abs(s[13:29]).max() < 1e-3
    assert abs(s[2]) < 2e-3                      # sinks ~1 mm (margin) and stays
    assert e.last_num_contacts == (64 if model == "round1" else 128)      # 32 cylinders x 2 end points / x 4 cached points


def test_anisotropic_friction_ratio(oracle_mod):
    """A straight snake sliding along its axis (link-local z, scale 0.01) decelerates far
    less than one sliding sideways (local x or y, scale 1 / 0.1)."""
    dec = {}
    for name, vel in (("axial", [0.5, 0, 0]), ("lateral", [0, 0.5, 0])):
        e = oracle_mod.OracleEnv()
        for _ in range(60):
            e.substep(np.zeros(16))          # settle
        s = e.get_state()
        s[10:13] = vel
        e.set_state(s)
        v0 = np.array(vel, float)
        for _ in range(5):
            e.substep(np.zeros(16))
        v1 = e.get_state()[10:13]
        dec[name] = np.linalg.norm(v0) - np.linalg.norm(v1)
    assert dec["axial"] > 0 and dec["lateral"] > 0
    assert dec["lateral"] > 5 * dec["axial"]


def test_joint_limit_row(oracle_mod):
    """A joint beyond +1.57 is pushed back by the limit row."""
    n = 16
    e = oracle_mod.OracleEnv(max_motor_impulse=0.0)     # motors off: an inf-force motor row outvotes the limit
    s = e.get_state()
    s[2] = 5.0
    s[13 + 4] = 1.6
    s[13 + n + 4] = 1.0
    e.set_state(s)
    e.substep(np.zeros(n))
    s1 = e.get_state()
    assert s1[13 + n + 4] < 0.0


def test_float_oracle_tracks_double(oracle_mod):
    rng = np.random.default_rng(9)
    a = oracle_mod.OracleEnv(residual_threshold=0.0)
    b = oracle_mod.OracleEnv(f32=True, residual_threshold=0.0)
    s = random_state(rng, 16, z=0.03, qamp=0.2, vamp=0.2, flat=True)
    a.set_state(s)
    b.set_state(s)
    t = rng.uniform(-0.5, 0.5, 16)
    a.substep(t)
    b.substep(t)
    sa, sb = a.get_state(), b.get_state()
    assert np.allclose(sa, sb, atol=2e-3)


# ------------------------------------------------------------------------------------------------------------------
# Row order (round 6): two [U] rules about the ORDER in which Bullet's solver meets its rows.
# ------------------------------------------------------------------------------------------------------------------
def test_quicksort_on_equal_keys_is_the_published_algorithm(oracle_mod):
    """orc_quicksort_equal_keys against a line-by-line Python restatement of btAlignedObjectArray::quickSortInternal
    (Hoare partition, pivot = the middle element, `i <= j` swap, recursion into [lo, j] and [i, hi]) run with a
    comparator that is always false -- what sorting by island id is when every constraint sits in the one island."""
    def qs(d, lo, hi, less):
        i, j = lo, hi
        x = d[(lo + hi) // 2]
        while True:
            while less(d[i], x):
                i += 1
            while less(x, d[j]):
                j -= 1
            if i <= j:
                d[i], d[j] = d[j], d[i]
                i += 1
                j -= 1
            if not i <= j:
                break
        if lo < j:
            qs(d, lo, j, less)
        if i < hi:
            qs(d, i, hi, less)
    for n in (1, 2, 3, 7, 32, 33, 64):
        d = list(range(n))
        if n > 1:
            qs(d, 0, n - 1, lambda a, b: False)
        assert oracle_mod.quicksort_equal_keys(n) == d, n
    # the 16-link world's list [limit_1..16, motor_1..16]: the motors come out first, in this order, then the limits
    p = oracle_mod.quicksort_equal_keys(32)
    assert sorted(p) == list(range(32)) and p != list(range(32))
    assert [x - 16 for x in p[:16]] == [5, 4, 7, 6, 1, 0, 3, 2, 13, 12, 15, 14, 9, 8, 11, 10]
    assert p[16:] == [5, 4, 7, 6, 1, 0, 3, 2, 13, 12, 15, 14, 9, 8, 11, 10]


def test_row_order_switches_keep_the_physics_and_change_the_numbers(oracle_mod):
    """noncontact_order 1 / contact_order 1, 3: the same constraints in another order.  A snake at rest still carries its
    weight (sum of normal impulses / dt = 21.296 x 9.8 N) whatever the order; the motor row still reaches its target in
    free space; and a moving snake ends a substep elsewhere than under the default order -- the order is part of what 50
    unconverged sweeps compute."""
    import bench
    n = 16
    for over in (dict(noncontact_order=1), dict(contact_order=1), dict(contact_order=2), dict(contact_order=3), dict(noncontact_order=1, contact_order=4)):
        e = oracle_mod.OracleEnv(residual_threshold=0.0, **over)
        e.reset()
        for _ in range(240):
            e.substep(np.zeros(n))
        lam = e.last_normal_impulses()
        assert abs(lam.sum() / e.params.dt - 21.296 * 9.8) < 0.02 * 21.296 * 9.8, (over, lam.sum() / e.params.dt)
        # the contacts come in the switch's order: runs of equal link, each link once
        links = e.last_contacts_full()[:, 4].astype(int)
        runs = [links[0]] + [b for a, b in zip(links[:-1], links[1:]) if a != b]
        assert len(set(runs)) == len(runs)
        if over.get("contact_order") == 1:
            assert runs == sorted(runs, reverse=True)
        elif over.get("contact_order") == 2:
            # link order after Bullet's quickSort on 32 equal keys: cylinder c sits on link 3 c / 2 + 2 (c even) or
            # 3 (c + 1) / 2 + 1 (c odd); a snake at rest has every cylinder on the ground
            want = [(3 * c // 2 + 2 if c % 2 == 0 else 3 * (c + 1) // 2 + 1) for c in oracle_mod.quicksort_equal_keys(2 * n)]
            assert runs == want, (runs, want)
        elif over.get("contact_order"):
            assert runs != sorted(runs) and runs != sorted(runs, reverse=True)
        else:
            assert runs == sorted(runs)
        # a moving snake: two env-steps of the gait, then one substep under this order and under the default from the same state
        a = oracle_mod.OracleEnv(residual_threshold=0.0, **over)
        a.reset()
        for j in range(2):
            a.env_step(bench.gait_actions([0], j)[0], vec_mode=False)
        S, M = a.get_state(), a.get_manifold()
        T = np.zeros(n)
        T[1::2] = bench.gait_actions([0], 2)[0] * (np.pi / 6)
        b = oracle_mod.OracleEnv(residual_threshold=0.0)
        b.set_state(S)
        b.set_manifold(M)
        a.substep(T)
        b.substep(T)
        diff = np.abs(a.get_state()[13 + n:] - b.get_state()[13 + n:]).max()
        assert 1e-6 < diff < 1.0, (over, diff)
