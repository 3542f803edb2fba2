"""The obstacle block of the reference's gait-test script: `snake/block.urdf`, a 0.2 x 0.8 x 0.2 m box
that `Snake.add_obstacle` (/root/reference/snake.py:83-84; commented out on the training path, :94) and
`snake_gait_test.py:51` put in front of the snake, and the script's "hit the wall" read-out: the z component of the
first motor joint's reaction force (`getJointState(robot, 3)[2][2] > 20`, snake_gait_test.py:33-40,126).

Two forms (snk_params / orc_params `obstacle`): 1 = STATIC (an immovable box; a 16-link handle stays on the
register-resident solve, up to 8 box contacts out of its 64 slots) and 2 = as the reference loads it (useFixedBase=0): a
FREE 200-kg body with its own six velocity components, gravity, damping and persistent manifold with the plane, rows
with the snake's links over both bodies (16 links, streamed-row kernels).  Contacts: GJK between each cylinder and the
box, one point per pair per step.  SNK_FORCE_STREAMED=1 puts a 16-link handle on the streamed-row kernels of the 32-link
chain, and the two solves must agree.

Tolerances: one env-step from a synchronised state, float32 GPU vs float64 oracle, judged against the float32 build of
the oracle on the same step (factor 3, floors 1e-3 on angles / pose, 5e-2 relative on joint velocities, 2 N on the
reaction force, which is impulse / dt: a 240x amplification of the solver's round-off)."""
import numpy as np
import pytest

from conftest import f32_gate      # noqa: E402

from conftest import ROUND1, random_state

gpu = pytest.mark.gpu
BOX = dict(obstacle=1, obstacle_pos=[0.100, 0.0, 0.1])        # its face 2 mm in front of the resting snake's head


def test_oracle_box_stops_the_snake_and_trips_the_wall_signal(oracle_mod):
    import bench
    e = oracle_mod.OracleEnv(**BOX)
    p = oracle_mod.OracleEnv()
    e.reset(); p.reset()
    mx = mx_free = 0.0
    for j in range(12):
        a = bench.gait_actions([5], j)[0]
        o, _, _, _, _ = e.env_step(a.copy(), vec_mode=False)
        o2, _, _, _, _ = p.env_step(a.copy(), vec_mode=False)
        mx = max(mx, abs(e.joint3_reaction_fz()))
        mx_free = max(mx_free, abs(p.joint3_reaction_fz()))
    assert o2[48] > 0.01 and o[48] < 0.003          # the free snake advances, the box holds the other back
    assert mx > 20.0 and mx_free < 10.0             # snake_gait_test.py:126's threshold separates the two


@gpu
@pytest.mark.parametrize("n,model", [(16, "default"), (32, "default"), (16, "round1"), (32, "round1")])
def test_obstacle_env_step_parity(pkg, oracle_mod, n, model):
    """model "default": everything at once -- 32-gon hulls, persistent ground manifolds (the cache is handed to the
    oracle with the state), link-link contacts (32 links) and the box; "round1": the stateless two-point manifold on
    implicit cylinders."""
    import bench
    B, J = 8, 6
    A = n // 2
    ids = np.arange(B)
    over = dict(BOX, n_modules=n)
    if model == "round1":
        over.update(ROUND1)
    st = pkg.Stepper(B, **over)
    st.reset()
    sc = 1 if n == 32 else 0                       # what the kernels evaluate for this chain length
    refs = [oracle_mod.OracleEnv(self_collision=sc, max_self_contacts=32, max_contacts=0, **over) for _ in range(B)]
    refs32 = [oracle_mod.OracleEnv(self_collision=sc, max_self_contacts=32, max_contacts=0, f32=True, **over)
              for _ in range(B)]
    w = dict(q=0.0, qd=0.0, r=0.0, f3=0.0)
    c = dict(q=0.0, qd=0.0, r=0.0, f3=0.0)
    mism = touched = 0
    for j in range(J):
        S, X = st.get_state()
        Mf = st.get_manifold()
        a = bench.gait_actions(ids + 5, j, A).astype(np.float32)
        obs, rew, done, sub = st.step(a.copy(), vec_mode=False)
        f3 = st.joint3_reaction_fz()
        for i in range(B):
            out = []
            for e in (refs[i], refs32[i]):
                e.set_state(S[i].astype(np.float64))
                e.set_aux(X[i, :n].astype(np.float64), float(X[i, n]), float(X[i, n + 1]))
                if Mf is not None:
                    e.set_manifold(Mf[i].astype(np.float64))
                out.append(e.env_step(a[i].astype(np.float64), vec_mode=False) + (e.joint3_reaction_fz(),))
            (o, r, d, k, _, g3), (o32, r32, d32, k32, _, g32) = out
            lc = refs[i].last_contacts_full()
            if len(lc) and (lc[:, 5] != -1).any():       # a link-link (>= 0) or box (-2) contact among the solved ones
                touched += 1

            def smooth(rr, oo):
                # the reward without its -10 step at |joint-0 force| > 10 (SnakeGymEnv.py:94): the force is impulse / dt,
                # good to a few newtons in float32, so which side of 10 it falls on is a boundary decision like the
                # substep count
                return rr + (10.0 if abs(oo[3 * n + 7]) > 10.0 else 0.0)

            def errs(oo, rr, ff):
                q = max(np.abs(oo[:n] - o[:n]).max(), np.abs(oo[3 * n:3 * n + 7] - o[3 * n:3 * n + 7]).max())
                qd = (np.abs(oo[n:2 * n] - o[n:2 * n]) / (1 + np.abs(o[n:2 * n]))).max()
                return dict(q=q, qd=qd, r=abs(smooth(rr, oo) - smooth(r, o)), f3=abs(ff - g3))
            if k32 == k and d32 == d:
                for key, v in errs(o32, r32, g32).items():
                    c[key] = max(c[key], v)
            if k != sub[i] or d != bool(done[i]):
                mism += 1
                assert abs(k - sub[i]) <= 1
                continue
            if k == 0:
                continue                          # no substep ran: the read-out keeps its previous value
            for key, v in errs(obs[i].astype(np.float64), float(rew[i]), float(f3[i])).items():
                w[key] = max(w[key], v)
    print("obstacle parity n =", n, model, "GPU-f32", w, "| oracle-f32", c, "| boundary mismatches", mism, "| steps touching the box", touched)
    assert touched >= B - 2          # the case does exercise box / link-link contacts (a count along a chaotic trajectory)
    assert mism <= max(2, B * J // 10)
    # (maxima of ~40 chaotic samples; 32 links: 2 x anyway)
    for key, fl in (("q", 5e-4), ("qd", 5e-2), ("r", 2e-3), ("f3", 2.0)):
        f32_gate("obstacle parity n = %d %s: worst %s" % (n, model, key), w[key], c[key], 2.0, fl)
    st.close()


@gpu
def test_wall_signal_on_device(pkg):
    """Free-running: with the box in its way the 16-link snake stays put and the first motor joint's reaction exceeds the
    script's threshold; without it neither happens.  The read-out is there on every handle."""
    import bench
    B = 16
    ids = np.arange(B) + 5
    res = {}
    for name, over in (("box", BOX), ("free", dict(obstacle=1, obstacle_pos=[5.0, 0.0, 0.1]))):
        st = pkg.Stepper(B, **over)
        st.reset()
        mx = np.zeros(B)
        for j in range(12):
            o, r, d, s = st.step(bench.gait_actions(ids, j).astype(np.float32), vec_mode=False)
            mx = np.maximum(mx, np.abs(st.joint3_reaction_fz()))
        res[name] = (o[:, 48].copy(), mx)
        st.close()
    assert np.median(res["free"][0]) > 0.01 and np.median(res["box"][0]) < 0.004
    assert np.median(res["box"][1]) > 20.0 and np.median(res["free"][1]) < 10.0
    plain = pkg.Stepper(2)
    plain.reset()
    plain.step(bench.gait_actions(np.arange(2), 0).astype(np.float32))
    assert np.all(np.isfinite(plain.joint3_reaction_fz())) and np.abs(plain.joint3_reaction_fz()).max() > 0
    plain.close()


@gpu
def test_add_obstacle_mirror(pkg):
    """Snake.add_obstacle (snake.py:83-84) on the single-env mirror: the world is rebuilt with the box in it."""
    robot = pkg.Snake(None, "snake/snake.urdf", None)
    env = pkg.SnakeGymEnv(robot, None)
    robot.add_obstacle("block.urdf", [0.100, 0.0, 0.1])
    assert env.params.obstacle == 2                       # loadURDF's default: useFixedBase = 0, a free body
    import bench
    mx = 0.0
    for j in range(12):
        o, r, d, info = env.step(bench.gait_actions([5], j)[0].astype(np.float64))
        mx = max(mx, abs(float(env._stepper.joint3_reaction_fz()[0])))
    assert o[48] < 0.004 and mx > 20.0
    robot.add_obstacle("block.urdf", [0.100, 0.0, 0.1], static=True)
    assert env.params.obstacle == 1
    env.close()


@gpu
def test_test_mode_survives_world_rebuilds(pkg):
    """mode='test' replays every env-step substep by substep on a scratch handle (SnakeGymEnv._record_telemetry) and
    insists on ending on the step kernel's bits.  The scratch handle must follow the world: add_obstacle AFTER a first
    step rebuilds it with the box, a hard reset empties its contact cache like the main handle's."""
    import argparse
    import bench
    args = argparse.Namespace(alpha=1.0, beta=0.01, gamma=0.1, mode="test", gaitSelection=1, scaling_factor=6,
                              motorVelocityLimit=np.inf, motorTorqueLimit=np.inf)
    robot = pkg.Snake(None, None, args=args)
    env = pkg.SnakeGymEnv(robot, args=args)
    env.reset()
    o, r, d, info = env.step(bench.gait_actions([5], 0)[0].astype(np.float64))
    assert len(info["internal_observations"]) == robot.counter > 0
    robot.add_obstacle("block.urdf", [0.100, 0.0, 0.1])              # a hard reset of the world, now with the box
    mx = 0.0
    for j in range(6):
        o, r, d, info = env.step(bench.gait_actions([5], j)[0].astype(np.float64))     # raises if the replay diverges
        assert len(info["link_positions"]) == robot.counter
        mx = max(mx, abs(float(env._stepper.joint3_reaction_fz()[0])))
    assert o[48] < 0.004 and mx > 10.0                                # the box is there, in both handles
    env.reset(hardReset=True)                                          # a populated contact cache must not survive
    for j in range(3):
        o, r, d, info = env.step(bench.gait_actions([5], j)[0].astype(np.float64))
    env.close()


@gpu
@pytest.mark.parametrize("box", [False, True])
def test_two_solves_one_physics(pkg, monkeypatch, box):
    """The 16-link chain through both solves: rows resident in registers (lane = row builder, two rows per register) and
    rows streamed from memory (SNK_FORCE_STREAMED=1: rows built lane = velocity component from the columns of M^-1,
    40-lane solve) -- without and WITH the obstacle box in the snake's way (its contacts take slots of the register-
    resident solve; both paths run their own float32 GJK, row builder and constraint pass).  Two independent float32 implementations of the same substep: one env-step
    (13-31 substeps of 50 Gauss-Seidel iterations on a contact-rich state) from a common state, for gait and for random
    actions.  They part the way float32 and float64 part (test_env_step_parity_*: 1e-3 in angle at worst), not the way
    two models would: median 1e-4 in angle, nine in ten near 1e-3."""
    import bench
    B = 64
    rng = np.random.default_rng(3)
    over = dict(BOX) if box else {}
    a_ = pkg.Stepper(B, **over)
    monkeypatch.setenv("SNK_FORCE_STREAMED", "1")
    b_ = pkg.Stepper(B, **over)
    monkeypatch.delenv("SNK_FORCE_STREAMED")
    a_.reset(); b_.reset()
    f3a = []
    eq, eqd, er = [], [], []
    mism = 0
    for j in range(8):
        S, X = a_.get_state()
        b_.set_state(S, X)
        b_.set_manifold(a_.get_manifold())            # the contact cache is simulator state too
        act = bench.gait_actions(np.arange(B), j).astype(np.float32) if j < 5 else rng.uniform(-1, 1, (B, 8)).astype(np.float32)
        oa, ra, da, sa = a_.step(act.copy(), vec_mode=False)
        ob, rb, db, sb = b_.step(act.copy(), vec_mode=False)
        same = (sa == sb) & (da == db)
        mism += int((~same).sum())
        if box:
            fa, fb = a_.joint3_reaction_fz(), b_.joint3_reaction_fz()
            f3a.append(np.abs(fa - fb)[same & (sa > 0)])
        oa, ob = oa[same].astype(np.float64), ob[same].astype(np.float64)
        eq.append(np.maximum(np.abs(oa[:, :16] - ob[:, :16]).max(axis=1), np.abs(oa[:, 48:55] - ob[:, 48:55]).max(axis=1)))
        eqd.append((np.abs(oa[:, 16:32] - ob[:, 16:32]) / (1 + np.abs(oa[:, 16:32]))).max(axis=1))
        d = np.abs(ra[same] - rb[same])
        er.append(d[d < 5.0])                      # (the -10 step of the reward at |joint-0 force| > 10 is a boundary decision)
    eq, eqd, er = np.concatenate(eq), np.concatenate(eqd), np.concatenate(er)
    st = {k: (float(np.median(v)), float(np.percentile(v, 90)), float(v.max())) for k, v in (("q", eq), ("qd", eqd), ("r", er))}
    print("register-resident vs streamed-row solve, 16 links (median, 90th percentile, worst):", st,
          "| substep-count mismatches", mism, "of", 8 * B)
    if box:
        f3d = np.concatenate(f3a)
        print("   joint-3 reaction, the two solves (median, 90th percentile):", float(np.median(f3d)), float(np.percentile(f3d, 90)))
        assert np.median(f3d) < 0.5 and np.percentile(f3d, 90) < 5.0          # impulse / dt: newtons of float32 noise
        assert a_.contact_overflow() == (0, 0, 0)
    assert mism <= 8 * B // 20
    # (the float32 oracle against the float64 one on the same kind of steps: medians 3e-4 / 1e-2, 90th percentiles
    #  1e-3 / 5e-2 -- test_env_step_parity_random_actions[16])
    assert st["q"][0] < 3e-4 and st["q"][1] < 3e-3 and st["q"][2] < 2e-2
    assert st["qd"][0] < 1e-2 and st["qd"][1] < 0.15
    assert st["r"][0] < 2e-4 and st["r"][1] < 2e-3
    a_.close(); b_.close()


# ---------------------------------------------------------------------------------------------------------------------
# obstacle = 2: the box as the reference loads it (useFixedBase=0): a free 200-kg body (snake/block.urdf:6)
# ---------------------------------------------------------------------------------------------------------------------
GAIT_TEST_WORLD = dict(dt=0.01, gravity_z=-9.81, max_motor_impulse=4.0 * 0.01)      # snake_gait_test.py:52-53,76: setTimeStep(0.01),
                                                                                    # g = -9.81, forces = 4 N m -> impulse 4 dt


def test_oracle_free_box_rests_on_its_four_corners(oracle_mod):
    """The box is a btMultiBody without links: gravity, damping, ONE new support corner per step into a persistent
    manifold.  Dropped where loadURDF puts it, it rocks onto its four bottom corners within a quarter of a second and
    the ground then carries exactly its weight."""
    e = oracle_mod.OracleEnv(obstacle=2, obstacle_pos=[2.0, 0.0, 0.1])
    e.reset()
    counts = []
    for k in range(240):
        e.substep(np.zeros(16))
        counts.append(int(e.get_box()[1][0]))
    s, m = e.get_box()
    assert counts[0] == 1 and counts[-1] == 4
    assert abs(s[2] - 0.1) < 2e-4 and np.abs(s[7:]).max() < 1e-4 and np.abs(s[3:6]).max() < 1e-4
    lc = e.last_contacts_full()
    imp = e.last_normal_impulses(512)[:len(lc)]
    assert abs(imp[lc[:, 4] < 0].sum() * 240.0 - 200.0 * 9.8) < 0.5
    # and the snake does not feel it from two metres away: same state as without the box
    p = oracle_mod.OracleEnv()
    p.reset()
    for k in range(240):
        p.substep(np.zeros(16))
    assert np.abs(p.get_state() - e.get_state()).max() < 1e-12


@gpu
@pytest.mark.parametrize("world", ["default", "gait_test"])
def test_free_box_env_step_parity(pkg, oracle_mod, world):
    """obstacle = 2 on the device (16 links, streamed-row kernels with six more velocity components) against the
    oracle, one env-step at a time from synchronised states -- snake, contact cache, box and the box's own manifold --
    in the reference's training world and in the gait-test script's (dt 0.01, 4-N-m motors, g -9.81,
    snake_gait_test.py:51-53,71-76), whose read-out (joint-3 reaction > 20: "the snake has hit the wall", :126) is
    compared as well.  Tolerances: calibrated against the float32 build of the oracle on the same steps, factor 4."""
    import bench
    B, J, n = 6, 8, 16
    over = dict(obstacle=2, obstacle_pos=[0.100, 0.0, 0.1])
    if world == "gait_test":
        over.update(GAIT_TEST_WORLD)
    st = pkg.Stepper(B, **over)
    st.reset()
    refs = [oracle_mod.OracleEnv(max_self_contacts=32, max_contacts=0, **over) for _ in range(B)]
    refs32 = [oracle_mod.OracleEnv(max_self_contacts=32, max_contacts=0, f32=True, **over) for _ in range(B)]
    ids = np.arange(B)
    w = dict(q=0.0, qd=0.0, r=0.0, f3=0.0, box=0.0)
    c = dict(q=0.0, qd=0.0, r=0.0, f3=0.0, box=0.0)
    mism = mism32 = touched = compared = 0
    peak = 0.0
    for j in range(J):
        S, X = st.get_state()
        Mf = st.get_manifold()
        BS, BM = st.get_box()
        a = bench.gait_actions(ids + 5, j).astype(np.float32)
        obs, rew, done, sub = st.step(a.copy(), vec_mode=False)
        f3 = st.joint3_reaction_fz()
        BS2, _ = st.get_box()
        peak = max(peak, float(np.abs(f3).max()))
        for i in range(B):
            out = []
            for e in (refs[i], refs32[i]):
                e.sync(S[i], X[i], Mf[i])
                e.set_box(BS[i], BM[i])
                out.append(e.env_step(a[i].astype(np.float64), vec_mode=False) + (e.joint3_reaction_fz(), e.get_box()[0]))
            (o, r, d, k, _, g3, b64), (o32, r32, d32, k32, _, g32, b32) = out
            lc = refs[i].last_contacts_full()
            if len(lc) and (lc[:, 5] == -2).any():
                touched += 1

            def smooth(rr, oo):
                return rr + (10.0 if abs(oo[3 * n + 7]) > 10.0 else 0.0)

            def errs(oo, rr, ff, bb):
                q = max(np.abs(oo[:n] - o[:n]).max(), np.abs(oo[3 * n:3 * n + 7] - o[3 * n:3 * n + 7]).max())
                qd = (np.abs(oo[n:2 * n] - o[n:2 * n]) / (1 + np.abs(o[n:2 * n]))).max()
                return dict(q=q, qd=qd, r=abs(smooth(rr, oo) - smooth(r, o)), f3=abs(ff - g3),
                            box=max(np.abs(bb[:7] - b64[:7]).max(), 0.1 * np.abs(bb[7:] - b64[7:]).max()))
            if k32 == k and d32 == d:
                for key, v in errs(o32, r32, g32, b32).items():
                    c[key] = max(c[key], v)
            else:
                mism32 += 1                      # (the float32 oracle leaves the servo loop a substep apart as well)
            if k != sub[i] or d != bool(done[i]):
                mism += 1
                assert abs(k - sub[i]) <= 1
                continue
            if k == 0:
                continue
            compared += 1
            for key, v in errs(obs[i].astype(np.float64), float(rew[i]), float(f3[i]), BS2[i].astype(np.float64)).items():
                w[key] = max(w[key], v)
    print("free box parity,", world, "GPU-f32", w, "| oracle-f32", c, "| boundary mismatches", mism, "(oracle-f32:", mism32, ") | steps touching the box",
          touched, "| peak joint-3 reaction", peak, "| overflow", st.contact_overflow())
    assert touched >= B and compared >= B * J // 2
    assert mism <= max(2, B * J // 8, 2 * mism32)          # threshold decisions (servo tolerance, 41-substep cap)
    for key, fl in (("q", 5e-4), ("qd", 5e-2), ("r", 2e-3), ("f3", 2.0), ("box", 1e-5)):
        f32_gate("free box parity, %s: worst %s" % (world, key), w[key], c[key], 2.0, fl)
    assert peak > 20.0                                   # snake_gait_test.py:126: "The snake has hit the wall"
    st.close()


@gpu
def test_free_box_settles_moves_and_survives_a_checkpoint(pkg, tmp_path):
    """Free-running on the device: far from the snake the box settles on four corners where it was put; in the
    snake's way it is pushed (a little: 200 kg on friction 0.5 against 4-N-m motors) while a static one is not; results
    do not depend on the schedule, and a checkpoint carries the box."""
    import bench
    B = 64
    ids = np.arange(B) + 5
    far = pkg.Stepper(B, obstacle=2, obstacle_pos=[2.0, 0.0, 0.1])
    far.reset()
    for j in range(3):
        far.step(bench.gait_actions(ids, j).astype(np.float32))
    s, m = far.get_box()
    assert np.all(m[:, 0] >= 3) and np.abs(s[:, 2] - 0.1).max() < 3e-4 and np.abs(s[:, :2] - [2.0, 0.0]).max() < 1e-3
    far.close()
    near = pkg.Stepper(B, obstacle=2, obstacle_pos=[0.100, 0.0, 0.1])
    near.reset()
    outs = []
    for j in range(6):
        if j == 3:
            pkg.save_state(near, str(tmp_path / "box.npz"))
        outs.append(near.step(bench.gait_actions(ids, j).astype(np.float32)))
    s, m = near.get_box()
    assert np.median(s[:, 0]) > 0.1 + 1e-5 and np.abs(s[:, 0] - 0.1).max() < 5e-3       # pushed forward, a little
    again = pkg.Stepper(B, obstacle=2, obstacle_pos=[0.100, 0.0, 0.1])
    pkg.load_state(again, str(tmp_path / "box.npz"))
    for j in range(3, 6):
        o, r, d, k = again.step(bench.gait_actions(ids, j).astype(np.float32))
        assert np.array_equal(o, outs[j][0]) and np.array_equal(r, outs[j][1]) and np.array_equal(k, outs[j][3])
    s2, m2 = again.get_box()
    assert np.array_equal(s, s2) and np.array_equal(m, m2)
    near.close(); again.close()
    with pytest.raises(RuntimeError):
        pkg.Stepper(2, n_modules=32, obstacle=2)


@gpu
def test_more_than_eight_box_contacts_take_the_other_solve(pkg, oracle_mod):
    """The register-resident 16-link solve has eight slots for contacts with the box; a snake lying across it touches
    it with more cylinders than that.  Such a substep goes through the streamed-row solve (room for every cylinder), like
    one with more ground points than slots: nothing is dropped (snk_contact_overflow[2] stays 0), and the distribution
    of one-substep errors against the float64 oracle is the float32 oracle's -- medians and 90th percentiles over random
    ground states around the box (found with tools/acc_distribution.py at the end of round 3: with the eight-contact
    cap the 90th percentile was 70 x the float32 oracle's)."""
    B, n = 384, 16
    over = dict(obstacle=1, obstacle_pos=[0.35, 0.0, 0.1])
    rng = np.random.default_rng(4321)
    S = np.zeros((B, 13 + 2 * n), np.float32)
    for i in range(B):
        s = random_state(rng, n, z=0.026, qamp=0.3, vamp=0.3, flat=True)
        s[9] *= 0.1; s[7:9] *= 0.1
        S[i] = s
    T = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32)
    st = pkg.Stepper(B, residual_threshold=0.0, **over)
    st.set_state(S)
    st.substep(T, 1)
    G, _ = st.get_state()
    ov = st.contact_overflow()
    st.close()
    o = oracle_mod.OracleEnv(residual_threshold=0.0, max_contacts=0, **over)
    o32 = oracle_mod.OracleEnv(residual_threshold=0.0, max_contacts=0, f32=True, **over)
    eg, e32, many = [], [], 0
    for i in range(B):
        for e in (o, o32):
            e.hard_reset()
            e.set_state(S[i].astype(np.float64))
            e.substep(T[i].astype(np.float64))
        lc = o.last_contacts_full()
        many += int(len(lc) and (lc[:, 5] == -2).sum() > 8)
        r, r32 = o.get_state(), o32.get_state()
        f = lambda x: (np.abs(x[13 + n:] - r[13 + n:]) / (1 + np.abs(r[13 + n:]))).max()
        eg.append(f(G[i])); e32.append(f(r32))
    eg, e32 = np.array(eg), np.array(e32)
    print("box across the snake: states with more than 8 box contacts", many, "| counters", ov,
          "| GPU median / p90", np.median(eg), np.percentile(eg, 90), "| oracle-f32", np.median(e32), np.percentile(e32, 90))
    assert many >= 10 and ov[0] >= many and ov[1] == 0 and ov[2] == 0
    assert np.median(eg) < 2 * np.median(e32) and np.percentile(eg, 90) < 3 * np.percentile(e32, 90)
