"""GPU tests of the env-step path (SnakeGymEnv.step / SubprocVecEnv.step equivalents):
per-step parity against the oracle from synchronised states, API semantics, the 32-link and
per-env-friction configs, and size-independent properties at BASELINE's full 4096 envs.

Tolerances for one env-step (<= 41 substeps with ~64 contacts, 50 unconverged Gauss-Seidel
sweeps each; float32 GPU vs float64 oracle, re-synchronised before every step):
joint angles / base pose 5e-3 (1e-2 for the 32-link chain), reward 5e-3, joint velocities:
90th percentile of |dqd|/(1+|qd|) below 5e-2 (0.4: only 24 samples) and maximum below 0.75 (1.0).  These are the sensitivity of the system
to float32 round-off, not kernel error: the oracle itself built in float32 differs from the
float64 oracle by 2.5e-3 / 0.24 (max) on the same steps, and the test requires the GPU to
be no worse than twice (32-link: three times) that calibration, measured in the same run.  Substep counts and done
flags must be identical unless the deciding quantity is within 1e-3 of its threshold."""
import importlib
import os

import numpy as np
import pytest

from conftest import f32_gate, mismatch_gate      # noqa: E402

pytestmark = pytest.mark.gpu


def gait(ids, j, A=8):
    k = np.arange(A)
    return (-np.sin((2 * k[None, :] + 1) * 4.0 + 0.2 * j + 0.37 * np.asarray(ids)[:, None])).astype(np.float32)


def _near_threshold(o_ref, act, n, oracle_env):
    """True when the oracle's decision quantities sit within 1e-3 of a threshold."""
    if abs(abs(o_ref[9]) - 0.5) < 1e-3:
        return True
    if abs(oracle_env.mean_height() - 0.1) < 1e-3:
        return True
    return False


@pytest.mark.parametrize("n", [16, 32])
def test_env_step_parity_resynced(pkg, oracle_mod, n):
    B, J = (24, 6) if n == 16 else (8, 3)
    A = n // 2
    st = pkg.Stepper(B, n_modules=n)
    st.reset()
    refs = [oracle_mod.OracleEnv(n_modules=n) for _ in range(B)]
    ref32 = oracle_mod.OracleEnv(n_modules=n, f32=True)
    worst = dict(q=0.0, qd=0.0, r=0.0)
    cal = dict(q=0.0, qd=0.0, r=0.0)
    qd_errs, cal_qd_errs = [], []
    mism = cal_mism = 0
    for j in range(J):
        S, X = st.get_state()
        Mf = st.get_manifold()
        a = gait(range(B), j, A)
        obs, rew, done, sub = st.step(a.copy(), vec_mode=False)
        for i in range(B):
            e = refs[i]
            e.sync(S[i], X[i], None if Mf is None else Mf[i])
            o, r, d, k, _ = e.env_step(a[i].astype(np.float64), vec_mode=False)
            # calibration: the float32 build of the oracle on the same step
            ref32.sync(S[i], X[i], None if Mf is None else Mf[i])
            o32, r32, d32, k32, _ = ref32.env_step(a[i].astype(np.float64), vec_mode=False)
            if k32 == k and d32 == d:
                cal["q"] = max(cal["q"], np.abs(o32[:n] - o[:n]).max(), np.abs(o32[3 * n:3 * n + 7] - o[3 * n:3 * n + 7]).max())
                cqd = (np.abs(o32[n:2 * n] - o[n:2 * n]) / (1 + np.abs(o[n:2 * n]))).max()
                cal_qd_errs.append(cqd)
                cal["qd"] = max(cal["qd"], cqd)
                cal["r"] = max(cal["r"], abs(r32 - r))
            else:
                cal_mism += 1
            if k != sub[i] or d != bool(done[i]):
                mism += 1
                # allowed only at a decision boundary
                assert abs(k - sub[i]) <= 1 or _near_threshold(o, a[i], n, e), (i, j, k, sub[i], d, done[i])
                continue
            worst["q"] = max(worst["q"], np.abs(obs[i, :n] - o[:n]).max(), np.abs(obs[i, 3 * n:3 * n + 7] - o[3 * n:3 * n + 7]).max())
            eqd = (np.abs(obs[i, n:2 * n] - o[n:2 * n]) / (1 + np.abs(o[n:2 * n]))).max()
            qd_errs.append(eqd)
            worst["qd"] = max(worst["qd"], eqd)
            worst["r"] = max(worst["r"], abs(rew[i] - r))
            # joint-0 force of the LAST substep (the fused kernel evaluates the sensor pass only on
            # substeps that can be the last one): impulse / dt amplifies the solver's round-off 240x
            worst["fz"] = max(worst.get("fz", 0.0), abs(obs[i, 3 * n + 7] - o[3 * n + 7]) / (1.0 + abs(o[3 * n + 7])))
    p90 = float(np.percentile(qd_errs, 90))
    p90c = float(np.percentile(cal_qd_errs, 90))
    print("n", n, "GPU-f32 vs oracle-f64 worst", worst, "qd p90", p90, "| oracle-f32 vs oracle-f64", cal, "qd p90", p90c,
          "| boundary mismatches", mism, "(oracle-f32:", cal_mism, ") of", B * J)
    # the 32-link chain is twice as long and correspondingly more sensitive to round-off
    # kcal: how far beyond the float32 oracle's own distance from float64 the GPU may be.  Observed (round 5,
    # profiles/r05_accuracy_calibration.txt): 16 links q 1.00 x, reward 1.12 x, qd 0.61 x; 32 links 1.04 x, 1.16 x, 0.61 x
    # -- so 1.5 (2.0 for the longer chain's 24 samples), where rounds 3-4 allowed 2.0 (3.0)
    tq, tp90, tmax, kcal = (5e-3, 5e-2, 0.75, 1.5) if n == 16 else (1e-2, 0.4, 1.0, 2.0)
    # (hulls on a persistent manifold of one to four points roll more easily than round 1's two end-cap points per
    #  cylinder: the float32 ORACLE itself is 1e-2 off the float64 one on the worst of these steps, so the absolute caps
    #  give way to the calibration below)
    # (calibrated bounds, each with a hard outer cap beside it: a defect shared by the float32 and the float64 build of
    #  the oracle must not widen the gate with it -- ADVICE r3; the calibration values go into the printed line above)
    assert worst["q"] < min(max(tq, kcal * cal["q"]), 5 * tq) and worst["r"] < min(max(5e-3, kcal * cal["r"]), 2.5e-2)
    assert worst["fz"] < 1.0      # sanity only: bit-exactness of the sensor is test_sensor_pass_only_when_observable
    # joint velocities: the 90th percentile within twice the float32 oracle's (every rebuild
    # re-associates FMAs, so an absolute cap on a heavy-tailed error is a coin toss), hard cap tmax
    assert p90 < max(tp90, kcal * p90c) and worst["qd"] < tmax
    f32_gate("resynced env-steps n = %d: worst q / pose of %d" % (n, B * J), worst["q"], cal["q"], kcal, 1e-4)
    f32_gate("resynced env-steps n = %d: worst rel qd" % n, worst["qd"], cal["qd"], kcal, 1e-3)
    f32_gate("resynced env-steps n = %d: worst reward" % n, worst["r"], cal["r"], kcal, 2e-3)     # the energy term (qd x motor torque) is noisy
    f32_gate("resynced env-steps n = %d: rel qd p90" % n, p90, p90c, 2.0, 1e-3)
    mismatch_gate("resynced env-steps n = %d" % n, mism, cal_mism)
    assert mism <= B * J // 6


def test_vec_env_semantics(pkg, oracle_mod):
    """SubprocVecEnv contract: ndarrays out, done rows carry the post-reset obs and the -5."""
    B = 16
    env = pkg.SnakeVecEnv(B)
    assert env.num_envs == B and len(env) == B
    assert env.observation_space.shape == (56,) and env.action_space.shape == (8,)
    obs = env.reset()
    assert obs.shape == (B, 56) and obs.dtype == np.float32
    assert np.all(obs[:, :51] == 0) and np.all(obs[:, 51:55] == [0, 0, 0, 1])
    seen_done = False
    for j in range(8):
        a = gait(range(B), j) * (np.float32(1.5) if j == 3 else np.float32(1.0))
        a_before = a.copy()
        obs, rews, dones, infos = env.step(a[:, :, None] if j % 2 else a)   # ARS shape (N,8,1) / PPO (N,8)
        # SubprocVecEnv pickles the actions to its workers (ppo/multiprocessing_env.py:119-122): checkBound clips THEIR
        # copies, the trainer's array stays as it was (only the single-env seam mutates the caller's array)
        assert np.array_equal(a, a_before)
        assert isinstance(obs, np.ndarray) and rews.shape == (B,) and dones.dtype == bool
        assert isinstance(infos, tuple) and len(infos) == B and infos[0] == {}
        # a fresh, writable, picklable dict per env per step, as the reference's workers send: a wrapper
        # may annotate one env's info without touching another's or a later step's
        assert type(infos[0]) is dict and infos[0] is not infos[1]
        infos[0]["episode"] = j
        assert infos[1] == {}
        assert (1 - dones).sum() + dones.sum() == B                        # ppo/train.py:134
        if dones.any():
            seen_done = True
            i = int(np.argmax(dones))
            assert rews[i] < -4.0
            assert np.all(obs[i, :32] == 0) and np.all(obs[i, 48:51] == 0) and np.all(obs[i, 51:55] == [0, 0, 0, 1])
        assert np.all((env.last_substeps >= 0) & (env.last_substeps <= 41))
    assert seen_done
    env.close()
    env.close()   # idempotent like SubprocVecEnv.close
    # the opt-in form: ONE read-only dict for every env and step; still a dict, still picklable, writing fails loudly
    import pickle
    env = pkg.SnakeVecEnv(4, shared_infos=True)
    env.reset()
    infos = env.step(gait(range(4), 0))[3]
    assert isinstance(infos[0], dict) and infos[0] == {} and pickle.loads(pickle.dumps(infos)) == ({},) * 4
    with pytest.raises(TypeError):
        infos[0]["episode"] = 1
    env.close()


def test_single_env_api(pkg):
    """SnakeGymEnv: terminal obs on done, in-place clip, info {}, robot getters."""
    robot = pkg.Snake(None, "snake/snake.urdf", None)
    env = pkg.SnakeGymEnv(robot, None)
    o = env.reset()
    assert o.shape == (56,) and o.dtype == np.float64
    assert env.render().size == 0 and env.mode == 'train'
    a = np.array([2.0, -3.0, 0.5, 0.2, -0.2, 0.1, 0.0, 0.3])
    o, r, d, info = env.step(a)
    assert info == {} and isinstance(r, float) and isinstance(d, bool)
    assert a[0] == 1.0 and a[1] == -1.0                                     # checkBound mutates the caller's array
    assert 1 <= robot.counter <= 41
    assert len(robot.getBasePosition()) == 3 and not robot.checkSnakeHeight()
    done = False
    for j in range(6):
        o, r, done, _ = env.step(gait([0], j)[0].astype(np.float64))
        if done:
            break
    assert done and abs(o[9]) > 0.5 and r < -4                              # terminal obs, not the reset one
    assert np.all(env.robot.getObservation()[:32] == 0)                     # ... but the env has soft-reset
    o2 = env.reset(hardReset=True)
    assert np.all(o2[:48] == 0)
    with pytest.raises(SystemError):
        env.step(np.zeros(5))
    env.close()


def test_subproc_vec_env_dropin(pkg):
    """ppo/train.py:69-70 style construction from thunks."""
    def make_env():
        def _thunk():
            robot = pkg.Snake(None, "snake/snake.urdf", None)
            return pkg.SnakeGymEnv(robot, None)
        return _thunk
    envs = pkg.SubprocVecEnv([make_env() for _ in range(4)])
    assert envs.num_envs == 4 and envs.observation_space.shape[0] == 56 and envs.action_space.shape[0] == 8
    state = envs.reset()
    assert state.shape == (4, 56)
    ns, rw, dn, _ = envs.step(np.zeros((4, 8)))
    assert ns.shape == (4, 56) and sum(rw) == 0.0
    envs.close()
    # thunks wrapped the way the reference's own SubprocVecEnv wraps them for its workers (multiprocessing_env.py:107)
    envs = pkg.SubprocVecEnv([pkg.CloudpickleWrapper(make_env()) for _ in range(3)])
    assert envs.num_envs == 3 and envs.reset().shape == (3, 56)
    envs.close()
    # thunks that build test-mode envs (ppo/params.py --mode test): every env's info carries its per-substep telemetry,
    # as each of the reference's workers would send it through its Pipe (SnakeGymEnv.py:43-44 via multiprocessing_env.py:
    # 11-16) -- the same lists, bit for bit, that the single-env seam gives for the same env-step
    class Args:
        alpha, beta, gamma = 1.0, 0.01, 0.1
        gaitSelection, scaling_factor, mode = 1, 6.0, 'test'
        motorVelocityLimit, motorTorqueLimit = np.inf, np.inf

    def make_test_env():
        return lambda: pkg.SnakeGymEnv(pkg.Snake(None, "snake/snake.urdf", Args()), Args())
    import bench
    NE = 3
    envs = pkg.SubprocVecEnv([make_test_env() for _ in range(NE)])
    assert envs.mode == 'test'
    singles = [make_test_env()() for _ in range(NE)]
    envs.reset()
    for e in singles:
        e.reset()
    ends = 0
    for j in range(12):
        a = bench.gait_actions(np.arange(NE), j).astype(np.float32)
        a[1] *= 1.5                                     # env 1: beyond the bounds (clipped), so the three differ
        S, X = envs._stepper.get_state()
        M = envs._stepper.get_manifold()
        obs, rews, dones, infos = envs.step(a)
        assert isinstance(infos, tuple) and len(infos) == NE
        for i, e in enumerate(singles):
            e._stepper.set_state(S[i:i + 1], X[i:i + 1])
            e._stepper.set_manifold(M[i:i + 1])
            o1, r1, d1, info1 = e.step(a[i].copy())
            k = envs.last_substeps[i]
            assert sorted(infos[i]) == sorted(info1) == ['frames', 'internal_observations', 'link_positions']
            assert infos[i]['frames'] == [] and len(infos[i]['internal_observations']) == len(infos[i]['link_positions']) == k == e.robot.counter
            for u, v in zip(infos[i]['internal_observations'] + infos[i]['link_positions'], info1['internal_observations'] + info1['link_positions']):
                assert u.dtype == np.float64 and np.array_equal(u, v)
            assert bool(dones[i]) == d1 and rews[i] == np.float32(r1)
            if k and not d1:
                assert np.array_equal(infos[i]['internal_observations'][-1].astype(np.float32), obs[i])
            ends += int(d1)
    envs.close()
    for e in singles:
        e.close()


def test_ground_friction_config(pkg, oracle_mod):
    """BASELINE config 5: per-env plane friction; each env matches an oracle with that mu."""
    B = 8
    rng = np.random.default_rng(1)
    # plane friction above ~1.25 (mu > 2.5) puts the resting snake in a stick-slip regime where
    # the float32 build of the oracle itself differs from float64 by 6e-4 (1.5) .. 1e-2 (2.0)
    # after 6 substeps; the parity range stays below it and the tolerance is calibrated per env
    mu = rng.uniform(0.3, 1.2, B).astype(np.float32)
    st = pkg.Stepper(B, residual_threshold=0.0)
    st.set_ground_friction(mu)
    st.reset()
    T = gait(range(B), 0, 16) * 0.3
    st.substep(T, 6)
    S, _ = st.get_state()
    xs = []
    for i in range(B):
        e = oracle_mod.OracleEnv(residual_threshold=0.0)
        e.set_plane_friction(float(mu[i]))
        e.reset()
        for _ in range(6):
            e.substep(T[i].astype(np.float64))
        ref = e.get_state()
        e32 = oracle_mod.OracleEnv(residual_threshold=0.0, f32=True)
        e32.set_plane_friction(float(mu[i]))
        e32.reset()
        for _ in range(6):
            e32.substep(T[i].astype(np.float64))
        r32 = e32.get_state()
        cal_p, cal_v = np.abs(r32[:7] - ref[:7]).max(), np.abs(r32[13:29] - ref[13:29]).max()
        # 6 substeps from rest: float32 round-off accumulates to ~5e-4 in the joint angles
        # (per environment: one sample each, hence 2 x)
        f32_gate("ground friction env %d: pose after 6 substeps" % i, np.abs(S[i, :7] - ref[:7]).max(), cal_p, 2.0, 2e-4)
        f32_gate("ground friction env %d: joint angles" % i, np.abs(S[i, 13:29] - ref[13:29]).max(), cal_v, 2.0, 1e-3)
        xs.append(ref[0:2])
    assert np.ptp(np.array(xs), axis=0).max() > 1e-6      # friction actually changes the motion


def test_full_size_properties(pkg):
    """4096 envs (BASELINE configs[1]): determinism, env independence, invariants."""
    B = 4096
    a_envs = pkg.Stepper(B)
    b_envs = pkg.Stepper(B)
    a_envs.reset(); b_envs.reset()
    ids = np.arange(B)
    tot_sub = 0
    for j in range(3):
        act = gait(ids, j)
        oa, ra, da, sa = a_envs.step(act.copy())
        ob, rb, db, sb = b_envs.step(act.copy())
        # bitwise reproducible
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db) and np.array_equal(sa, sb)
        assert np.all(np.isfinite(oa)) and np.all(np.isfinite(ra))
        assert sa.min() >= 0 and sa.max() <= 41
        assert np.all(ra[da] < -4.0) and np.all(np.abs(ra[~da]) < 4.0)
        q = oa[:, 51:55]
        assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-5)
        assert np.all(oa[da][:, :32] == 0)
        tot_sub += sa.sum()
    assert tot_sub > 0
    # env i does not depend on who shares the batch: rerun 5 of them alone
    pick = [0, 1, 777, 2048, 4095]
    small = pkg.Stepper(len(pick))
    small.reset()
    ref_all = pkg.Stepper(B)
    ref_all.reset()
    for j in range(2):
        act = gait(ids, j)
        o_all, r_all, d_all, s_all = ref_all.step(act.copy())
        o_s, r_s, d_s, s_s = small.step(act[pick].copy())
        assert np.array_equal(o_all[pick], o_s) and np.array_equal(r_all[pick], r_s) and np.array_equal(s_all[pick], s_s)


def test_masked_reset_and_state_roundtrip(pkg):
    B = 6
    st = pkg.Stepper(B)
    st.reset()
    st.step(gait(range(B), 0))
    S, X = st.get_state()
    mask = np.array([1, 0, 0, 1, 0, 0], dtype=np.uint8)
    obs = st.reset(mask)
    S2, X2 = st.get_state()
    assert np.all(S2[[0, 3], :3] == 0) and np.all(S2[[0, 3], 13:] == 0)
    assert np.array_equal(S2[[1, 2, 4, 5]], S[[1, 2, 4, 5]])
    assert np.array_equal(X2[:, :17], X[:, :17])             # motor-torque / sensor caches survive a soft reset
    assert np.all(obs[[1, 2, 4, 5]] == 0)                      # unmasked rows untouched (buffer was zeros)
    st.set_state(S, X)
    S3, X3 = st.get_state()
    assert np.array_equal(S3, S) and np.array_equal(X3, X)


def test_device_vec_env_torch(pkg):
    """Device-pointer form: torch tensors in/out on the current stream, in-place clipping."""
    import torch
    B = 32
    env = pkg.DeviceVecEnv(B, device_index=0)
    host = pkg.Stepper(B)
    env.reset(); host.reset()
    a = gait(range(B), 0) * 1.5
    ta = torch.from_numpy(a.copy()).cuda()
    obs, rew, done = env.step(ta)
    torch.cuda.synchronize()
    ho, hr, hd, _ = host.step(a.copy())
    assert np.array_equal(obs.cpu().numpy(), ho) and np.array_equal(rew.cpu().numpy(), hr)
    assert np.array_equal(done.cpu().numpy().astype(bool), hd)
    assert float(ta.abs().max()) <= 1.0
    env.close()


def test_sharded_env_over_rccl_world1(pkg):
    """The RCCL (backend 'nccl') scatter/gather path of ShardedVecEnv with a single rank:
    same calls the N-GPU bench makes, results equal the unsharded device env."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B = 64
        local = pkg.DeviceVecEnv(B, device_index=0)
        ref = pkg.DeviceVecEnv(B, device_index=0)
        env = pkg.ShardedVecEnv(local, root=0, device=torch.device("cuda", 0))
        o0 = env.reset(); r0 = ref.reset()
        torch.cuda.synchronize()
        assert torch.equal(o0, r0)
        for j in range(3):
            a = torch.from_numpy(gait(range(B), j)).cuda()
            obs, rew, done, infos = env.step(a.clone())
            o2, r2, d2 = ref.step(a.clone())
            torch.cuda.synchronize()
            assert len(infos) == B
            assert torch.equal(obs, o2) and torch.equal(rew, r2) and torch.equal(done, d2.bool())
    finally:
        dist.destroy_process_group()


def test_link_positions_rest_pose_known_answers(pkg):
    """snk_link_positions = getLinkPositions (snake.py:138-146) against the URDF-derived rest-pose
    COM table (tests/golden/appendix_b.json)."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_b.json")))
    st = pkg.Stepper(3)
    st.reset()
    lp = st.link_positions()
    assert lp.shape == (3, 51)
    ref = np.array([g["base_link_com_rest"]] + g["output_body_com_rest"])      # [17, 3]
    for e in range(3):
        got = lp[e].reshape(3, 17).T
        assert np.abs(got - ref).max() < 1e-6


def test_test_mode_telemetry(pkg, oracle_mod):
    """mode='test': info carries the observation and the link positions after every substep
    (SnakeGymEnv.py:43-44, snake.py:291-293); checked substep by substep against the oracle."""
    import argparse
    args = argparse.Namespace(alpha=1.0, beta=0.01, gamma=0.1, mode="test", gaitSelection=1, scaling_factor=6,
                              motorVelocityLimit=np.inf, motorTorqueLimit=np.inf)
    env = pkg.SnakeGymEnv(pkg.Snake(None, None, args=args), args=args)
    ref = oracle_mod.OracleEnv()
    ref32 = oracle_mod.OracleEnv(f32=True)      # calibration: the oracle built in float32, same substeps
    env.reset()
    ref.reset()
    idx = 1 + np.arange(0, 49, 3)          # oracle rows: root + Bullet links 0..48
    for j in range(3):
        a = gait(range(1), j, 8)[0].astype(np.float64) * 1.3       # some components get clipped
        S, X = env._stepper.get_state()
        Mf = env._stepper.get_manifold()
        ref.sync(S[0], X[0], None if Mf is None else Mf[0])
        ref32.sync(S[0], X[0], None if Mf is None else Mf[0])
        obs, rew, done, info = env.step(a.copy())
        k = env.robot.counter
        assert set(info) == {"frames", "internal_observations", "link_positions"} and info["frames"] == []
        assert len(info["internal_observations"]) == k == len(info["link_positions"]) and k > 0
        assert np.array_equal(info["internal_observations"][-1], obs)
        targets = np.zeros(16)
        targets[1::2] = np.clip(a, -1, 1) * (np.pi / 6)
        for i in range(k):
            ref.substep(targets)
            ref32.substep(targets)
            o = ref.get_obs()
            o32 = ref32.get_obs()
            cal = max(np.abs(o32[:16] - o[:16]).max(), np.abs(o32[48:55] - o[48:55]).max())
            tol = min(max(2e-4 * (i + 1), 3 * cal), 2e-3 * (i + 1))           # float32 drift over the substeps of one env-step
            assert np.abs(info["internal_observations"][i][:16] - o[:16]).max() < tol
            assert np.abs(info["internal_observations"][i][48:55] - o[48:55]).max() < tol
            lp = info["link_positions"][i].reshape(3, 17).T
            assert np.abs(lp - ref.link_com_world()[idx]).max() < tol
        if done:
            break
    env.close()


@pytest.mark.parametrize("variant", ["gait0_weights", "gait2_identity"])
def test_env_step_variants(pkg, oracle_mod, variant):
    """createAction's other branches (snake.py:247-269: gait 0 -> even slots, else 16 actions
    1:1), a different action scale and other reward weights: same parity as the default config."""
    if variant == "gait0_weights":
        over = dict(gait=0, scaling_factor=np.pi / 4, alpha=2.0, beta=0.05, gamma=0.2)
        A = 8
    else:
        over = dict(gait=2, scaling_factor=np.pi / 8)
        A = 16
    B = 8
    st = pkg.Stepper(B, **over)
    assert st.act_dim == A
    st.reset()
    refs = [oracle_mod.OracleEnv(**over) for _ in range(B)]
    ref32 = oracle_mod.OracleEnv(f32=True, **over)
    rng = np.random.default_rng(12)
    compared = 0
    wq_all, wr_all, cq_all, cr_all = [], [], [], []
    for j in range(3):
        S, X = st.get_state()
        Mf = st.get_manifold()
        a = rng.uniform(-1.2, 1.2, (B, A)).astype(np.float32)
        a_in = a.copy()
        obs, rew, done, sub = st.step(a, vec_mode=False)
        assert np.array_equal(a, np.clip(a_in, -1, 1))                  # clipped in place
        for i in range(B):
            e = refs[i]
            e.sync(S[i], X[i], None if Mf is None else Mf[i])
            o, r, d, k, _ = e.env_step(a_in[i].astype(np.float64), vec_mode=False)
            ref32.sync(S[i], X[i], None if Mf is None else Mf[i])
            o32, r32, d32, k32, _ = ref32.env_step(a_in[i].astype(np.float64), vec_mode=False)
            if k != sub[i] or d != bool(done[i]):
                assert abs(k - sub[i]) <= 1 or _near_threshold(o, a[i], 16, e)
                continue
            compared += 1
            # one env-step of float32 round-off on a stiff system: 2.5e-3 is typical for the worst env; these variants
            # drive the pitch joints (the snake lifts itself off the ground), where the float32 ORACLE is off by
            # several 1e-2 on some steps: calibrated against it -- the worst and the median of the <= 24 env-steps, 2 x
            cq = cr = 0.0
            if k32 == k and d32 == d:
                cq = max(np.abs(o32[:16] - o[:16]).max(), np.abs(o32[48:55] - o[48:55]).max())
                cr = abs(r32 - r)
            wq_all.append(max(np.abs(obs[i, :16] - o[:16]).max(), np.abs(obs[i, 48:55] - o[48:55]).max()))
            wr_all.append(abs(rew[i] - r))
            if k32 == k and d32 == d:
                cq_all.append(cq); cr_all.append(cr)
            assert wq_all[-1] < 0.25 and wr_all[-1] < 0.5         # hard caps per env-step (the pitch joints lift the snake: the float32 ORACLE is 0.1 off on some); the calibrated gates below
    assert compared >= 2 * B
    f32_gate("env-step variant %s: worst q / pose of %d" % (variant, compared), max(wq_all), max(cq_all), 2.0, 5e-3)
    f32_gate("env-step variant %s: median q / pose" % variant, np.median(wq_all), np.median(cq_all), 2.0, 5e-4)
    f32_gate("env-step variant %s: worst reward" % variant, max(wr_all), max(cr_all), 2.0, 1e-2)


def test_ragged_sizes_and_argument_errors(pkg):
    """Batch sizes that are not multiples of anything (1, 3, 65, 1000): env i's trajectory does not
    depend on the batch it is stepped in.  Bad arguments fail with a message, never a crash."""
    def run(n):
        st = pkg.Stepper(n)
        st.reset()
        outs = []
        for j in range(2):
            o, r, d, s = st.step(gait(range(n), j, 8))
            outs.append((o, r, d, s))
        st.close()
        return outs
    big = run(1000)
    for n in (1, 3, 65):
        small = run(n)
        for (o, r, d, s), (O, R, D, S) in zip(small, big):
            assert np.array_equal(o, O[:n]) and np.array_equal(r, R[:n]) and np.array_equal(d, D[:n]) and np.array_equal(s, S[:n])
    with pytest.raises(RuntimeError, match="n_envs"):
        pkg.Stepper(0)
    with pytest.raises(RuntimeError):
        pkg.Stepper(4, n_modules=12)             # kernels exist for the 16- and 32-link chains
    st = pkg.Stepper(4)
    lib = pkg.load()
    assert lib.snk_step_host(st.h, None, None, None, None, None, 1) != 0 and b"null" in lib.snk_last_error()
    assert lib.snk_get_obs(None, None) != 0
    with pytest.raises(AssertionError):
        st.step(np.zeros((4, 7), np.float32))    # wrong action width is caught before the C call
    with pytest.raises(AssertionError):
        st.set_ground_friction(np.ones(3, np.float32))
    st.close()


def test_env_logic_branches(pkg, oracle_mod):
    """The servo loop's and the reward's rare branches, constructed on purpose, GPU vs oracle:
    0-substep step (snake.py:283 never enters the loop), the 41-substep cap (:303), height
    termination after one substep (:299-301), and the -10 collision term (SnakeGymEnv.py:94)."""
    n = 16

    def both(over, state=None, aux=None, action=None, want32=False):
        st = pkg.Stepper(1, **over)
        st.reset()
        e = oracle_mod.OracleEnv(**over)
        e.reset()
        if state is not None:
            S, X = st.get_state()
            S[0, :len(state)] = state
            if aux is not None:
                X[0] = aux
            st.set_state(S, X)
            e.set_state(S[0].astype(np.float64))
            e.set_aux(X[0, :n].astype(np.float64), float(X[0, n]), float(X[0, n + 1]))
        a = np.zeros((1, 8), np.float32) if action is None else np.asarray(action, np.float32).reshape(1, 8)
        obs, rew, done, sub = st.step(a.copy(), vec_mode=False)
        o, r, d, k, _ = e.env_step(a[0].astype(np.float64), vec_mode=False)
        st.close()
        if want32:          # calibration: the float32 build of the oracle on the same step (from a reset state only)
            e32 = oracle_mod.OracleEnv(f32=True, **over)
            e32.reset()
            o32 = e32.env_step(a[0].astype(np.float64), vec_mode=False)[0]
            return (obs[0], float(rew[0]), bool(done[0]), int(sub[0])), (o, r, d, k), o32
        return (obs[0], float(rew[0]), bool(done[0]), int(sub[0])), (o, r, d, k)

    # (1) targets already reached: no physics at all, observation is the old state
    g, o = both({})
    assert g[3] == 0 == o[3] and not g[2] and g[1] == 0.0 == o[1]
    assert np.all(g[0][:51] == 0)
    # (2) motors too weak to reach the target: the loop stops at the 41-substep cap
    g, o, o32 = both(dict(max_motor_impulse=2e-5), action=[1.0] * 8, want32=True)
    assert g[3] == 41 == o[3] and g[2] == o[2]
    # 41 substeps of saturated motors against sticking contacts: float32 round-off grows to a few 1e-2 (more on hulls
    # that can roll over their vertices): no worse than 3x the float32 oracle on the same step
    assert np.abs(g[0][:16] - o[0][:16]).max() < max(0.1, 3 * np.abs(o32[:16] - o[0][:16]).max())
    # (3) snake in the air: mean height > 0.1 ends the step after ONE substep, done, -5
    g, o = both({}, state=[0.0, 0.0, 0.5], action=[0.5] * 8)
    assert g[3] == 1 == o[3] and g[2] and o[2]
    assert abs(g[1] - o[1]) < 1e-3 and g[1] < -4.0
    # (4) stale joint-0 force above 10 with a 0-substep step: reward = -10 exactly (dx = dy = energy = 0)
    aux = np.zeros(n + 2, np.float32); aux[n] = 15.0
    g, o = both({}, state=[0.0, 0.0, 0.0], aux=aux)
    assert g[3] == 0 == o[3] and g[1] == -10.0 == o[1] and g[0][55] == 15.0


@pytest.mark.parametrize("n", [16, 32])
def test_sensor_pass_only_when_observable(pkg, n):
    """The fused env-step kernel skips the joint-0 force sensor pass on substeps that cannot be
    the last of their env-step.  obs[55] must still be exactly what the always-evaluating
    single-substep API gives when the same substeps are replayed one by one."""
    from bench import gait_actions
    B, A = (512, 8) if n == 16 else (128, 16)
    st = pkg.Stepper(B, n_modules=n)
    rp = pkg.Stepper(B, n_modules=n)
    st.reset()
    rng = np.random.default_rng(3)
    for j in range(6 if n == 16 else 3):
        a = gait_actions(np.arange(B), j, A).astype(np.float32) if j % 2 == 0 else rng.uniform(-1, 1, (B, A)).astype(np.float32)
        S, X = st.get_state()
        Mf = st.get_manifold()                                           # the contact cache is simulator state too
        obs, rew, done, sub = st.step(a.copy(), vec_mode=False)          # terminal obs even when done
        # replay with the substep API: group envs by their substep count
        T = np.zeros((B, n), np.float32)
        T[:, 1::2] = np.clip(a, -1, 1) * np.float32(np.pi / 6)
        # Replay one substep per call (which solve a substep takes is decided from the state alone, substep by substep,
        # so the fused kernel and the one-substep calls go the same way)
        Xf = X.copy()
        rp.set_state(S, X)
        if Mf is not None:
            rp.set_manifold(Mf)
        for c in range(1, int(sub.max()) + 1):
            rp.substep(T, 1)
            if np.any(sub == c):
                Xc = rp.get_state()[1]
                Xf[sub == c] = Xc[sub == c]
        moved = sub > 0
        assert moved.sum() > B // 2
        assert np.array_equal(obs[moved, 3 * n + 7], Xf[moved, n]), np.abs(obs[moved, 3 * n + 7] - Xf[moved, n]).max()
        assert np.array_equal(obs[~moved, 3 * n + 7], X[~moved, n])
    st.close(); rp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,order", [(16, 0), (32, 0), (16, 3), (32, 3)])
def test_schedule_does_not_change_results(pkg, monkeypatch, n, order):
    """The step kernel cuts env-steps into slices that move between waves (snk_device.hpp:
    env_step_sched_kernel).  Whatever the slice length -- 1 substep, 3, whole env-steps, or the
    unscheduled kernel (SNK_QUANTUM=0) -- every env must end every step on the same bits: a slice
    boundary stores and reloads exactly the state a continuing wave keeps.  5000 envs over 4 steps is
    ~300 000 hand-offs between waves on all 8 XCDs, so a stale record would show.  order: snk_params::contact_order (round 6)
    -- the compact contact list laid out in another sweep order must be as indifferent to where a slice ends."""
    B = (5000 if n == 16 else 1200) if order == 0 else (2000 if n == 16 else 600)
    A = n // 2

    def run(quantum):
        monkeypatch.setenv("SNK_QUANTUM", str(quantum))
        st = pkg.Stepper(B, n_modules=n, contact_order=order)
        st.reset()
        st.set_ground_friction((0.5 + np.arange(B) % 11 / 10.0).astype(np.float32))
        outs = []
        for j in range(4):
            a = (gait(range(B), j, A) * 1.2).astype(np.float32)     # some components get clipped in place
            o, r, d, s = st.step(a)
            outs.append((o.copy(), r.copy(), d.copy(), s.copy(), a.copy()))
        S, X = st.get_state()
        st.close()
        return outs, S, X

    ref, S0, X0 = run(0)
    assert max(s.max() for _, _, _, s, _ in ref) > 25 and min(s.min() for _, _, _, s, _ in ref) < 20
    for quantum in (1, 3, 64):
        got, S, X = run(quantum)
        for (o, r, d, s, a), (O, R, D, Sx, Ax) in zip(got, ref):
            assert np.array_equal(s, Sx) and np.array_equal(d, D)
            assert np.array_equal(o, O) and np.array_equal(r, R) and np.array_equal(a, Ax)
        assert np.array_equal(S, S0) and np.array_equal(X, X0)


@pytest.mark.gpu
def test_many_live_handles(pkg):
    """The scheduled kernel reads its model from one of 32 constant-memory slots; handles beyond that
    fall back to the unscheduled kernel and give the same results; slots come back on close()."""
    a = gait(range(4), 0, 8)
    hs = [pkg.Stepper(4) for _ in range(36)]
    outs = []
    for h in hs:
        h.reset()
        outs.append(h.step(a.copy()))
    for o, r, d, s in outs[1:]:
        assert np.array_equal(o, outs[0][0]) and np.array_equal(r, outs[0][1]) and np.array_equal(s, outs[0][3])
    for h in hs:
        h.close()
    h = pkg.Stepper(4, kp=0.2)           # a freed slot, a different model
    h.reset()
    o2, _, _, s2 = h.step(a.copy())
    assert s2.max() < outs[0][3].max()   # stiffer servo converges in fewer substeps
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("over", [dict(kp=1.0), dict(kp=0.02), dict(max_counter=100, servo_tol=1e-4),
                                  dict(gait=2, kp=0.3), dict(max_counter=0)])
def test_schedule_with_unusual_servo_parameters(pkg, monkeypatch, over):
    """The scheduler predicts the remaining substeps from kp and the servo error; a wrong prediction may cost time,
    never results: one-substep servos (kp = 1), servos that hit the counter cap (kp = 0.02), caps beyond the
    queue's 64 priority classes, the identity gait and a cap of zero all match the unscheduled kernel bit for bit."""
    B = 700
    A = 16 if over.get("gait") == 2 else 8

    def run(quantum):
        monkeypatch.setenv("SNK_QUANTUM", str(quantum))
        st = pkg.Stepper(B, **over)
        st.reset()
        outs = []
        for j in range(3):
            o, r, d, s = st.step(gait(range(B), j, A).astype(np.float32))
            outs.append((o.copy(), r.copy(), d.copy(), s.copy()))
        st.close()
        return outs

    ref = run(0)
    for quantum in (1, 2):
        for got, want in zip(run(quantum), ref):
            for g, w in zip(got, want):
                assert np.array_equal(g, w)
    if "kp" in over and over["kp"] == 1.0:
        assert ref[0][3].max() <= 4
    if over.get("kp") == 0.02:
        assert ref[0][3].max() == 41
    if over.get("max_counter") == 100:
        assert ref[0][3].max() > 41


@pytest.mark.gpu
def test_free_running_gait_aggregates(pkg, oracle_mod):
    """SURVEY Appendix C-2, last bullet: what a trainer sees over a free-running rollout (no resynchronisation,
    auto-reset on, 32 envs x 40 env-steps of the bench's gait) -- mean substeps per env-step, episode-end rate and
    mean reward -- GPU float32 against the float64 oracle.  Trajectories are chaotic step by step; these
    aggregates agree to 3e-4 relative when measured (17.362 vs 17.366, 0.1000 vs 0.1000, -0.49390 vs -0.49392)."""
    import bench
    B, T = 32, 40
    ids = np.arange(B)
    st = pkg.Stepper(B)
    st.reset()
    refs = [oracle_mod.OracleEnv() for _ in range(B)]
    for r in refs:
        r.reset()
    g = np.zeros(3)
    o = np.zeros(3)
    for j in range(T):
        a = bench.gait_actions(ids, j)
        _, r, d, s = st.step(a.astype(np.float32))
        g += [s.sum(), d.sum(), r.sum()]
        for i in range(B):
            _, rr, rd, rk, _ = refs[i].env_step(a[i].copy(), vec_mode=True)
            o += [rk, int(rd), rr]
    g /= B * T
    o /= B * T
    assert abs(g[0] - o[0]) < 0.01 * o[0], (g, o)          # substeps per env-step
    assert abs(g[1] - o[1]) < 0.01, (g, o)                 # episode ends per env-step
    # mean reward: the -5 of an episode end dwarfs everything else in it (a handful of marginal |q9| > 0.5 crossings
    # falling the other way moves it by more than all the physics), so the penalty part is judged by the episode-end
    # rate above and the rest separately
    assert abs((g[2] + 5.0 * g[1]) - (o[2] + 5.0 * o[1])) < 0.05 * abs(o[2] + 5.0 * o[1]) + 2e-4, (g, o)
    assert abs(g[2] - o[2]) < 0.02 * abs(o[2]) + 5.0 * abs(g[1] - o[1]) + 1e-4, (g, o)
    st.close()


@pytest.mark.gpu
def test_bench_multi_rank_rehearsal():
    """`python bench.py --gpus 2` from a PLAIN shell (no torchrun around it, no WORLD_SIZE): bench.py launches its own
    ranks as a child process (the reference's SubprocVecEnv fans out by itself,
    ppo/multiprocessing_env.py:97-128), ShardedVecEnv scatter / gather, max-over-ranks timing, and rank 0's one JSON
    line comes back as the parent's only stdout line.  Rehearsed with two ranks that share this box's GPU over the
    gloo backend (SNK_BENCH_BACKEND=gloo: host-staged collectives; the RCCL form of the same calls is covered by
    test_sharded_env_over_rccl_world1 and runs on the driver's 8-GPU node)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SNK_BENCH_BACKEND"] = "gloo"
    # two ranks: with the launcher, the parent bench.py (which never opens the GPU) and pytest that is at most four
    # processes with the GPU open, inside the box's limit of six (more ranks: tests/test_dist_gloo.py runs 2 / 4 / 8)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--envs-per-gpu", "512"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]      # the ONE line, nothing else on stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["envs_per_gpu"] == 512 and r.get("cpu_baseline") is None      # the CPU baseline is an N = 1 leg
    assert r["config"]["world_size"] == 2 and r["config"]["backend"] == "gloo" and r["config"]["self_launched"] is True
    pids = {k["pid"] for k in r["config"]["ranks"]}
    assert len(pids) == 2 and os.getpid() not in pids, r["config"]["ranks"]
    assert sorted(k["rank"] for k in r["config"]["ranks"]) == [0, 1]
    # round 6: every rank's own time in the record, the entry point named, both timed regions
    assert all(k["ms_per_step"] > 0 and k["kernel_ms"] > 0 for k in r["config"]["ranks"])
    assert r["config"]["api"] == "ShardedVecEnv.step_block" and "H2D of the step's action block" in r["config"]["timed_region"]
    assert r["actions_resident"]["value"] > 0 and r["actions_resident"]["ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_line_contract_and_profile_form():
    """The bench line's round-6 fields on one GPU, small and quick: the action upload inside the timed region (named in
    config.timed_region), the actions-resident rate beside it, the entry point, rank 0's own time; and `--profile`, the form
    rocprofv3 is put around: W + K launches and nothing else -- no second timed region, no contact histogram, no CPU
    baseline, no variants (tools/summarize_prof.py relies on exactly W + K launches of the step kernel)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SNK_BENCH_BACKEND")}
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--envs-per-gpu", "512"]
    out = subprocess.run(base + ["--no-cpu-baseline", "--no-variants"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 1 and r["unit"] == "env-steps/s" and r["dtype"] == "f32" and r["vs_baseline"] is None
    assert r["config"]["api"] == "DeviceVecEnv.step_packed" and "H2D of the step's action block" in r["config"]["timed_region"]
    assert r["actions_resident"]["value"] > 0 and r["config"]["contacts_per_substep"]["substeps"] > 0
    assert r["config"]["ranks"][0]["ms_per_step"] > 0 and r["config"]["contact_order"] == 0
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1 and r["roofline"]["launches"] == 3
    out = subprocess.run(base + ["--profile"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    p = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert p["actions_resident"] is None and p["config"]["contacts_per_substep"] is None
    assert p["cpu_baseline"] is None and p["variants"] is None and p["roofline"]["launches"] == 3


@pytest.mark.gpu
def test_test_mode_telemetry_follows_overridden_parameters(pkg):
    """gait / scaling_factor passed through **over reach the device but not the robot facade; the test-mode
    replay must take them from the parameters the device uses (it raised "replay diverged" before)."""
    class Args:
        alpha, beta, gamma = 1.0, 0.01, 0.1
        gaitSelection, scaling_factor, mode = 1, 6.0, 'test'
        motorVelocityLimit, motorTorqueLimit = np.inf, np.inf
    robot = pkg.Snake(None, "snake/snake.urdf", Args())
    env = pkg.SnakeGymEnv(robot, Args(), gait=0, scaling_factor=np.pi / 5)      # overrides the args' gait 1, pi/6
    env.reset()
    o, r, d, info = env.step(np.linspace(-0.8, 0.8, 8))
    assert len(info['internal_observations']) == robot.counter >= 1
    assert np.array_equal(info['internal_observations'][-1].astype(np.float32), o.astype(np.float32))
    q = info['internal_observations'][-1][:16]
    assert np.abs(q[1::2]).max() < 1e-3 and np.abs(q[0::2]).max() > 0.05          # gait 0 drives the even slots
    env.close()


@pytest.mark.parametrize("quantum", ["1", "0"])
def test_step_packed_matches_step(pkg, monkeypatch, quantum):
    """snk_step_packed (rows [obs | reward | done u32 | padding] written by the step kernel, what ShardedVecEnv gathers)
    against snk_step from the same state: bit for bit, padding untouched; scheduled and unscheduled kernels."""
    import torch
    monkeypatch.setenv("SNK_QUANTUM", quantum)
    B, O = 96, 56
    env = pkg.DeviceVecEnv(B)
    env.reset()
    for j in range(3):
        env.step(torch.tensor(gait(range(B), j)).cuda())
    torch.cuda.synchronize()
    S, X = env.stepper.get_state()
    M = env.stepper.get_manifold()
    a = torch.tensor(gait(range(B), 3) * np.float32(1.3)).cuda()       # (some components leave [-1, 1]: clipped in place)
    a1 = a.clone()
    o, r, d = env.step(a1)
    torch.cuda.synchronize()
    o, r, d, sub = o.clone(), r.clone(), d.clone(), env.substeps.clone()
    env.stepper.set_state(S, X)
    env.stepper.set_manifold(M)
    packed = torch.full((B, O + 5), float("nan"), device="cuda")
    a2 = a.clone()
    env.step_packed(a2, packed)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and float(a2.abs().max()) <= 1.0
    assert torch.equal(packed[:, :O], o) and torch.equal(packed[:, O], r)
    assert torch.equal(packed.view(torch.int32)[:, O + 1], d.to(torch.int32))
    assert bool(torch.isnan(packed[:, O + 2:]).all()) and torch.equal(env.substeps, sub)
    assert int(d.sum()) > 0 or int(sub.max()) > 0
    env.close()


def test_robot_level_step_is_the_fused_kernels_servo_loop(pkg):
    """Snake.step called by itself (test_script.py:25) -- createAction, checkFeedback, one substep per pass, height exit,
    counter cap -- ends where the fused env-step kernel ends from the same state: same counter, same observation (the
    fused kernel adds reward / termination / reset on top, which Snake.step does not have)."""
    robot = pkg.Snake(None, "snake/snake.urdf", None)
    env = pkg.SnakeGymEnv(robot, None)
    env.reset()
    twin = pkg.Stepper(1)
    for j in range(5):
        a = gait([2], j)[0].astype(np.float64) * 0.9            # (|q9| stays below 0.5: no termination in the twin)
        S, X = env._stepper.get_state()
        twin.set_state(S, X)
        twin.set_manifold(env._stepper.get_manifold())
        obs, rew, done, sub = twin.step(a.astype(np.float32).reshape(1, 8).copy(), vec_mode=False)
        assert robot.step(list(a)) is True
        assert robot.counter == int(sub[0]) and not robot.endDue2Height and not done[0]
        assert np.array_equal(robot.getObservation().astype(np.float32), obs[0])
        assert np.array_equal(robot.getPosition(), robot.getObservation()[:16]) and robot.getLinkPositions().shape == (51,)
        assert robot.createAction(list(range(1, 9))) == [0, 1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 0, 7, 0, 8]
    env.close()
    twin.close()


@pytest.mark.gpu
def test_overflow_substep_replicas_agree(pkg, monkeypatch):
    """A substep whose contacts outgrow the register-resident solve's slots runs through the streamed-row substep in place,
    its contact cache handed over through global memory.  385 replicas of such an environment, scattered among 4615
    ordinary ones (their partner waves busy with the register-resident solve), must all end the env-step on the same bits,
    pass after pass.  Round 4 found them in two camps (64 against 65 contacts in that substep, a quarter of the replicas
    in the minority): the hand-over's write-through stores had not refreshed the storing CU's own L1, and the loads behind
    them sometimes hit lines cached when the environment was loaded (snk_device.hpp: substep())."""
    monkeypatch.setenv("SNK_QUANTUM", "0")
    B, n, A = 5000, 16, 8
    fr = (0.5 + np.arange(B) % 11 / 10.0).astype(np.float32)
    st = pkg.Stepper(B, n_modules=n)
    st.reset()
    st.set_ground_friction(fr)
    a = np.clip(gait(range(B), 0, A) * 1.2, -1, 1).astype(np.float32)
    S, X = st.get_state()
    Mf = st.get_manifold()
    tg = np.zeros((B, n), np.float32)
    tg[:, 1::2] = a * np.float32(np.pi / 6)
    e = None
    for k in range(28):                                  # the gait's first env-step: an env that overflows on the way
        info = st.substep(tg, 1)
        hit = np.nonzero(info[:, 1] > 64)[0]
        if len(hit):
            e = int(hit[0])
            break
    assert e is not None, "no substep beyond 64 contacts in the gait's first env-step"
    idx = np.arange(7, B, 13)
    S2, X2, M2, a2, f2 = S.copy(), X.copy(), Mf.copy(), a.copy(), fr.copy()
    S2[idx], X2[idx], M2[idx], a2[idx], f2[idx] = S[e], X[e], Mf[e], a[e], fr[e]
    first = None
    for rep in range(3):
        st.set_ground_friction(f2)
        st.set_state(S2, X2)
        st.set_manifold(M2)
        c0 = st.contact_overflow()[0]
        o, r, d, s = st.step(a2.copy())
        assert st.contact_overflow()[0] - c0 >= len(idx)             # every replica took the in-place streamed substep
        camps = np.unique(o[idx], axis=0)
        assert len(camps) == 1, "pass %d: %d different outcomes among %d replicas of env %d" % (rep, len(camps), len(idx), e)
        assert np.unique(s[idx]).size == 1 and np.unique(r[idx]).size == 1
        first = o.copy() if first is None else first
        assert np.array_equal(o, first)                              # ... and the whole handle repeats bit for bit
    st.close()


REPLICA_CASES = [(16, 4000, {}), (16, 3000, dict(warm_start=1)), (32, 2600, {}),
                 (16, 3000, dict(obstacle=1, obstacle_pos=[0.12, 0.0, 0.1])), (16, 3000, dict(obstacle=2, obstacle_pos=[0.12, 0.0, 0.1])),
                 (16, 4000, dict(hull_sides=0, contact_model=0, relative_breaking_threshold=0)),
                 (32, 2600, dict(obstacle=1, obstacle_pos=[0.12, 0.0, 0.1]))]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(REPLICA_CASES)))
def test_replicas_agree_whatever_runs_beside_them(pkg, monkeypatch, case):
    """The stronger form of the schedule test (round 4, after the stale contact cache of the in-place streamed substep):
    replicas of one environment -- state, contact cache, free box, friction, action -- every 13th slot of a handle full of
    ordinary environments.  Whatever wave runs a replica, whatever runs beside it on the SIMD and the CU, it must end the
    env-step on the bits the source environment itself ended on; scheduled and unscheduled kernels."""
    n, B, over = REPLICA_CASES[case]
    A = n // 2
    for quantum in (1, 0):
        monkeypatch.setenv("SNK_QUANTUM", str(quantum))
        st = pkg.Stepper(B, n_modules=n, **over)
        st.reset()
        fr = (0.5 + np.arange(B) % 11 / 10.0).astype(np.float32)
        st.set_ground_friction(fr)
        for j in range(2):
            st.step((gait(range(B), j, A) * 1.2).astype(np.float32))
        S, X = st.get_state()
        Mf = st.get_manifold()
        BX = st.get_box() if over.get("obstacle") == 2 else None
        a = (gait(range(B), 2, A) * 1.2).astype(np.float32)
        o0, r0, d0, s0 = st.step(a.copy())
        idx = np.arange(7, B, 13)
        for e in (int(np.argmax(s0)), int(np.argmin(s0 + 100 * (s0 == 0))), 1234 % B):
            S2, X2, a2, f2 = S.copy(), X.copy(), a.copy(), fr.copy()
            S2[idx], X2[idx], a2[idx], f2[idx] = S[e], X[e], a[e], fr[e]
            st.set_ground_friction(f2)
            st.set_state(S2, X2)
            if Mf is not None:
                M2 = Mf.copy()
                M2[idx] = Mf[e]
                st.set_manifold(M2)
            if BX is not None:
                b0, b1 = BX[0].copy(), BX[1].copy()
                b0[idx], b1[idx] = BX[0][e], BX[1][e]
                st.set_box(b0, b1)
            o, r, d, s = st.step(a2.copy())
            assert len(np.unique(o[idx], axis=0)) == 1, (n, over, quantum, e)
            assert np.array_equal(o[idx[0]], o0[e]) and r[idx[0]] == r0[e] and s[idx[0]] == s0[e], (n, over, quantum, e)
        st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,B,over", [(16, 4096, {}), (32, 2048, {}), (16, 2048, dict(obstacle=2, obstacle_pos=[0.12, 0.0, 0.1]))])
def test_twin_environments_stay_identical(pkg, n, B, over):
    """env i and env i + B/2: same friction, same actions (gait, every seventh step random and out of range), same start --
    other slots, other waves, other neighbours.  80 env-steps, auto-resets and in-place streamed substeps included; the
    twins must agree bit for bit throughout (tools/dbg/twins_soak.py runs six configurations for 300 steps)."""
    A, H = n // 2, B // 2
    st = pkg.Stepper(B, n_modules=n, **over)
    st.reset()
    fr = (0.5 + np.arange(H) % 11 / 10.0).astype(np.float32)
    st.set_ground_friction(np.concatenate([fr, fr]))
    rng = np.random.default_rng(3)
    for j in range(80):
        a = (gait(range(H), j, A) * 1.2).astype(np.float32)
        if j % 7 == 3:
            a = rng.uniform(-2, 2, (H, A)).astype(np.float32)
        o, r, d, s = st.step(np.concatenate([a, a]))
        assert np.array_equal(o[:H], o[H:]) and np.array_equal(r[:H], r[H:]) and np.array_equal(s[:H], s[H:]), j
    S, X = st.get_state()
    assert np.array_equal(S[:H], S[H:]) and np.array_equal(X[:H], X[H:])
    if n == 16 and not over:
        assert st.contact_overflow()[0] > 0            # the run went through the in-place streamed substep as well
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,B,over", [(16, 4096, {}), (32, 2048, {}), (16, 2048, dict(obstacle=2, obstacle_pos=[0.12, 0.0, 0.1])),
                                      (16, 3000, dict(warm_start=1, obstacle=1, obstacle_pos=[0.12, 0.0, 0.1]))])
def test_slot_permutation_invariance(pkg, n, B, over):
    """An environment's outcome does not depend on the slot it sits in: the same handle stepped from the same states in
    natural order and in a random permutation (state, contact cache, free box, friction, actions permuted alike) gives
    the permuted outputs, bit for bit, over three env-steps in a row -- every environment of the handle, not a few
    replicated ones."""
    A = n // 2
    st = pkg.Stepper(B, n_modules=n, **over)
    st.reset()
    fr = (0.5 + np.arange(B) % 11 / 10.0).astype(np.float32)
    st.set_ground_friction(fr)
    for j in range(2):
        st.step((gait(range(B), j, A) * 1.2).astype(np.float32))
    S, X = st.get_state()
    Mf = st.get_manifold()
    BX = st.get_box() if over.get("obstacle") == 2 else None
    acts = [(gait(range(B), 2 + j, A) * 1.2).astype(np.float32) for j in range(3)]

    def run(p):
        st.set_ground_friction(fr[p])
        st.set_state(S[p], X[p])
        if Mf is not None:
            st.set_manifold(Mf[p])
        if BX is not None:
            st.set_box(BX[0][p], BX[1][p])
        outs = []
        for a in acts:
            o, r, d, s = st.step(a[p].copy())
            outs.append((o.copy(), r.copy(), d.copy(), s.copy()))
        return outs, st.get_state()[0]

    ident = np.arange(B)
    perm = np.random.default_rng(11).permutation(B)
    ref, Sr = run(ident)
    got, Sg = run(perm)
    for (o, r, d, s), (O, R, D, Sx) in zip(got, ref):
        assert np.array_equal(s, Sx[perm]) and np.array_equal(d, D[perm])
        assert np.array_equal(o, O[perm]) and np.array_equal(r, R[perm])
    assert np.array_equal(Sg, Sr[perm])
    st.close()


def test_alarm_poisons_the_handle(pkg):
    """The scheduler's failure path: every wait inside the step kernel is bounded, and a wave whose
    wait runs out sets a host-mapped word.  Here the HOST sets that word (snk_debug_raise_alarm: nothing waits, nothing
    hangs).  From then on the handle refuses: the step calls (device-pointer, packed and host-buffer forms), reset,
    substep and the state accessors all fail with the scheduler's message in snk_last_error; the counters and
    snk_destroy still work; a fresh handle is unaffected.  The reference's whole failure story is close() draining and
    joining the workers (ppo/multiprocessing_env.py:140-150): destroy and create again is this build's."""
    import torch
    B = 64
    st = pkg.Stepper(B)
    st.reset()
    a = gait(range(B), 0)
    obs, rew, done, sub = st.step(a.copy())
    assert np.isfinite(obs).all()
    S0, X0 = st.get_state()
    st.debug_raise_alarm()
    msg = "env-step scheduler: a bounded wait ran out"
    calls = {
        "snk_step_host": lambda: st.step(a.copy()),
        "snk_reset_host": lambda: st.reset(),
        "snk_substep_host": lambda: st.substep(np.zeros((B, 16), dtype=np.float32), 1),
        "snk_get_state": lambda: st.get_state(),
        "snk_set_state": lambda: st.set_state(S0, X0),
        "snk_get_manifold": lambda: st.get_manifold(),
        "snk_get_obs": lambda: st.get_obs(),
        "snk_mean_height": lambda: st.mean_height(),
    }
    for name, fn in calls.items():
        with pytest.raises(RuntimeError) as ei:
            fn()
        assert msg in str(ei.value), (name, str(ei.value))
    # the device-pointer forms refuse BEFORE launching anything
    t_a = torch.tensor(a).cuda()
    t_o = torch.zeros((B, 56), device="cuda")
    t_r = torch.zeros((B,), device="cuda")
    t_d = torch.zeros((B,), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError) as ei:
        st.step_device(t_a.data_ptr(), t_o.data_ptr(), t_r.data_ptr(), t_d.data_ptr())
    assert msg in str(ei.value)
    t_p = torch.zeros((B, 58), device="cuda")
    with pytest.raises(RuntimeError) as ei:
        st.step_packed_device(t_a.data_ptr(), t_p.data_ptr(), 58)
    assert msg in str(ei.value)
    torch.cuda.synchronize()
    assert float(t_o.abs().sum()) == 0.0 and float(t_p.abs().sum()) == 0.0          # nothing ran
    assert msg in pkg.load().snk_last_error().decode()
    # what still works: the counters, the dimensions, and destroying the handle
    assert len(st.contact_overflow()) == 3 and st.obs_dim == 56
    st.close()
    # a new handle is healthy, and the poisoned one's alarm did not leak into it
    st2 = pkg.Stepper(B)
    st2.reset()
    obs2, _, _, _ = st2.step(a.copy())
    assert np.array_equal(obs2, obs)
    st2.close()
