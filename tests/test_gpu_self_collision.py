"""GPU tests of link-link (self) collision, SURVEY 8(f)-2: what URDF_USE_SELF_COLLISION (/root/reference/snake.py:93)
adds in PyBullet [U].  It matters for the 32-link snake, whose coil can reach itself (twelve yaw joints at 27.5 deg
bring modules 1-3 against modules 24-26); for the 16-link snake the rows exist but can never carry an impulse inside the
joint limits (tools/self_collision_clearance.py) -- checked here on the oracle, which can evaluate them for any chain.

Tolerances: contact geometry of the coil pose (float32 GJK against the oracle's float64 GJK) 2e-5 m on points and
distances, 1e-3 on normals (two hull facets can tie); states after K substeps of a tightening coil as in
tests/test_gpu_parity.py's ground case (5e-4 on positions and angles, 5e-2 relative on velocities)."""
import numpy as np
import pytest

from conftest import f32_gate      # noqa: E402

gpu = pytest.mark.gpu

N = 32


def coil(ang_deg=27.5, m=12):
    """Yaw joints (odd indices) of the first m yaw modules, alternating sign: the snake curls in the plane."""
    q = np.zeros(N)
    yaw = np.arange(1, N, 2)[:m]
    q[yaw] = np.radians(ang_deg) * np.where(np.arange(m) % 2 == 0, 1.0, -1.0)
    return q


def _coiled_states(B, seed=0):
    rng = np.random.default_rng(seed)
    S = np.zeros((B, 13 + 2 * N), np.float32)
    for i in range(B):
        # a little pitch and roll: a cylinder lying exactly flat has its end caps equally deep, and which vertex is the
        # support point is then decided by the last bit (in Bullet too; float32 and float64 decide differently)
        pr = rng.uniform(0.004, 0.012, 2)
        S[i, 3:7] = [0.5 * pr[0], 0.5 * pr[1], 0.0, 1.0]
        S[i, 3:7] /= np.linalg.norm(S[i, 3:7])
        S[i, 2] = 0.002
        S[i, 13:13 + N] = coil(27.5 + 0.2 * rng.uniform(-1, 1)) + rng.uniform(-0.01, 0.01, N) * (np.arange(N) % 2 == 1)
    return S


@gpu
@pytest.mark.parametrize("hull", [0, 32])
def test_coil_self_contacts_match_oracle(pkg, oracle_mod, hull):
    # (implicit cylinders: three substeps -- in the fourth the motors have pushed some pairs past both collision margins,
    #  where the narrow phase changes tier (shrunk cores, Bullet: EPA), and the float32 GJK of the kernels and the
    #  oracle's, two independent implementations, take that step a substep apart: 2e-3 against 4e-5)
    B, K = 6, (4 if hull else 3)
    over = dict(n_modules=N, hull_sides=hull, residual_threshold=0.0)
    st = pkg.Stepper(B, **over)
    S = _coiled_states(B, seed=hull)
    st.set_state(S, np.zeros((B, N + 2), np.float32))
    T = np.tile(coil(30.0).astype(np.float32), (B, 1))          # the motors tighten the coil: the head pushes the tail
    refs, plain, refs32 = [], [], []
    for i in range(B):
        e = oracle_mod.OracleEnv(self_collision=1, max_self_contacts=32, **over)
        e.set_state(S[i].astype(np.float64))
        refs.append(e)
        e32 = oracle_mod.OracleEnv(self_collision=1, max_self_contacts=32, f32=True, **over)
        e32.set_state(S[i].astype(np.float64))
        refs32.append(e32)
        p = oracle_mod.OracleEnv(self_collision=0, **over)
        p.set_state(S[i].astype(np.float64))
        plain.append(p)
    worst_p = worst_v = cal_p = cal_v = 0.0
    flips = 0
    acted = 0
    alive = np.ones(B, bool)
    for k in range(K):
        info = st.substep(T, 1)
        G, _ = st.get_state()
        for i in range(B):
            refs[i].substep(T[i].astype(np.float64))
            refs32[i].substep(T[i].astype(np.float64))
            plain[i].substep(T[i].astype(np.float64))
            if not alive[i]:
                continue
            if refs[i].last_num_contacts != info[i, 1]:
                alive[i] = False          # a pair at the breaking threshold counted on one side only
                flips += 1
                continue
            lc = refs[i].last_contacts_full()
            pair = lc[:, 5] >= 0                                      # link-link contacts follow the ground's
            assert pair.any() and not pair[0] and info[i, 1] == len(lc)
            imp = refs[i].last_normal_impulses(512)
            if imp[:len(lc)][pair].max() > 1e-4:
                acted += 1
            ref = refs[i].get_state()
            worst_p = max(worst_p, np.abs(G[i, :7] - ref[:7]).max(), np.abs(G[i, 13:13 + N] - ref[13:13 + N]).max())
            worst_v = max(worst_v, (np.abs(G[i, 13 + N:] - ref[13 + N:]) / (1 + np.abs(ref[13 + N:]))).max())
            if refs32[i].last_num_contacts == refs[i].last_num_contacts:       # calibration: the float32 oracle
                r32 = refs32[i].get_state()
                cal_p = max(cal_p, np.abs(r32[:7] - ref[:7]).max(), np.abs(r32[13:13 + N] - ref[13:13 + N]).max())
                cal_v = max(cal_v, (np.abs(r32[13 + N:] - ref[13 + N:]) / (1 + np.abs(ref[13 + N:]))).max())
    print("hull", hull, "coil parity: worst pos", worst_p, "rel qd", worst_v, "| oracle-f32", cal_p, cal_v,
          "| threshold flips", flips, "acted", acted)
    assert acted >= B            # the link-link rows carried impulses
    assert flips <= 1
    # stiff pushing contact between unlimited-force motors: float32 round-off in the GJK witness points is
    # amplified; the GPU must be no worse than three times the float32 build of the oracle on the same steps
    f32_gate("hull %d coil parity: worst pos" % hull, worst_p, cal_p, 2.0, 1e-4, 5e-3)
    f32_gate("hull %d coil parity: worst rel qd" % hull, worst_v, cal_v, 2.0, 1e-2, 1.0)
    # and they matter: without them the oracle's coil closes further
    with_sc = np.array([r.get_state()[13:13 + N] for r in refs])
    without = np.array([p.get_state()[13:13 + N] for p in plain])
    assert np.abs(with_sc - without).max() > 1e-3
    st.close()


@gpu
def test_self_collision_switch_and_gait_invariance(pkg):
    """self_collision = 0 removes the rows; on the bench gait (alternating bends, never closer than the speculative
    margin lets act) both settings give the same observations bit for bit, at 32 links and trivially at 16."""
    import bench
    B = 64
    ids = np.arange(B)
    for n in (32, 16):
        A = n // 2
        on = pkg.Stepper(B, n_modules=n, self_collision=1)
        off = pkg.Stepper(B, n_modules=n, self_collision=0)
        on.reset(); off.reset()
        for j in range(3):
            a = bench.gait_actions(ids, j, A).astype(np.float32)
            o1, r1, d1, s1 = on.step(a.copy())
            o0, r0, d0, s0 = off.step(a.copy())
            assert np.array_equal(o1, o0) and np.array_equal(r1, r0) and np.array_equal(s1, s0)
        on.close(); off.close()
    # the coil, by contrast, differs
    S = _coiled_states(4)
    T = np.tile(coil(30.0).astype(np.float32), (4, 1))
    out = []
    for sc in (1, 0):
        st = pkg.Stepper(4, n_modules=N, self_collision=sc)
        st.set_state(S, np.zeros((4, N + 2), np.float32))
        info = st.substep(T, 6)
        out.append((st.get_state()[0], info))
        st.close()
    assert np.all(out[0][1][:, 1] > out[1][1][:, 1])                  # more contacts with the flag on
    assert np.abs(out[0][0] - out[1][0]).max() > 1e-3


@pytest.mark.parametrize("relative", [1, 0])
def test_sixteen_link_self_contacts_are_inert_in_the_oracle(oracle_mod, relative):
    """The product builds no link-link rows for 16 links.  The oracle can: over a gait rollout with the flag on, under
    the dispatcher's relative breaking threshold (1.2 mm, the default) no pair ever comes close enough for a row; with
    the absolute 0.02 m the rows exist (bends above ~18 deg bring neighbouring cylinders within the threshold) and
    every one of them keeps a zero impulse.  Either way the results equal those without the flag."""
    import bench
    a_env = oracle_mod.OracleEnv(self_collision=1, relative_breaking_threshold=relative)
    b_env = oracle_mod.OracleEnv(self_collision=0, relative_breaking_threshold=relative)
    a_env.reset(); b_env.reset()
    seen = 0
    for j in range(12):
        act = bench.gait_actions([3], j)[0]
        oa, ra, da, ka, _ = a_env.env_step(act.copy(), vec_mode=True)
        ob, rb, db, kb, _ = b_env.env_step(act.copy(), vec_mode=True)
        assert ka == kb and da == db and np.allclose(oa, ob, rtol=0, atol=1e-12) and abs(ra - rb) < 1e-12
        if a_env.last_num_contacts > b_env.last_num_contacts:
            seen += 1
            extra = a_env.last_normal_impulses(512)[b_env.last_num_contacts:]
            assert np.all(extra == 0.0)
    assert (seen == 0) if relative else (seen > 0)


@pytest.mark.gpu
def test_sixteen_links_folded_onto_themselves(pkg, oracle_mod):
    """URDF_USE_SELF_COLLISION (snake.py:93) holds for the 16-link snake too.  Inside the reference's command range its
    links never come within reach of each other (tools/self_collision_clearance.py), but a snake bent further -- a larger
    scaling_factor, a state set from outside -- folds onto itself, and the link-link contacts then carry real impulses.
    The register-resident solve has no two-body rows: a substep in which a pair may touch goes through the streamed-row
    solve, which builds them.  Random states with joint angles up to 1.7 rad (a third of them with link-link contacts in
    the oracle): the distribution of one-substep errors against the float64 oracle is the float32 oracle's.  (Until the
    end of round 3 the 16-link kernels built no such rows: the 90th percentile here was 0.59 against 3e-4.)"""
    from conftest import random_state
    B, n = 384, 16
    rng = np.random.default_rng(4321)
    S = np.zeros((B, 13 + 2 * n), np.float32)
    for i in range(B):
        s = random_state(rng, n, z=0.026, qamp=1.7, vamp=0.3, flat=True)
        s[9] *= 0.1; s[7:9] *= 0.1
        S[i] = s
    T = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32)
    st = pkg.Stepper(B, residual_threshold=0.0)
    st.set_state(S)
    st.substep(T, 1)
    G, _ = st.get_state()
    ov = st.contact_overflow()
    st.close()
    o = oracle_mod.OracleEnv(residual_threshold=0.0, max_contacts=0)
    o32 = oracle_mod.OracleEnv(residual_threshold=0.0, max_contacts=0, f32=True)
    eg, e32, touching = [], [], 0
    for i in range(B):
        for e in (o, o32):
            e.hard_reset()
            e.set_state(S[i].astype(np.float64))
            e.substep(T[i].astype(np.float64))
        lc = o.last_contacts_full()
        touching += int(len(lc) and (lc[:, 5] >= 0).any())
        r, r32 = o.get_state(), o32.get_state()
        f = lambda x: (np.abs(x[13 + n:] - r[13 + n:]) / (1 + np.abs(r[13 + n:]))).max()
        eg.append(f(G[i])); e32.append(f(r32))
    eg, e32 = np.array(eg), np.array(e32)
    print("folded 16-link snakes: states with link-link contacts", touching, "of", B, "| counters", ov,
          "| GPU median / p90", np.median(eg), np.percentile(eg, 90), "| oracle-f32", np.median(e32), np.percentile(e32, 90))
    assert touching >= B // 8 and ov[0] >= touching and ov[1] == 0 and ov[2] == 0
    assert np.median(eg) < 2 * np.median(e32) and np.percentile(eg, 90) < 3 * np.percentile(e32, 90)
