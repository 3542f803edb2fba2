"""The reference's own end-to-end script, test_script_env.py: Snake + SnakeGymEnv on the module-level client, reset, 60 x
env.step([0.5] * 8) + env.render(), rewards summed.  tests/golden/make_test_script_vectors.py EXECUTED that script
(runpy) behind the oracle-backed client and stored what SnakeGymEnv.step saw and returned: 26 substeps for the first
step, none after it (the targets are reached), no episode end, total reward -0.042538.

CPU: the oracle's env-step reproduces the 60 steps from a reset, without re-synchronisation.
GPU: the script restated on the product's classes (same calls, same order) gives the same substep counts, no done,
observations and the total reward as close to the record as the float32 build of the oracle gets."""
import os

import numpy as np
import pytest

from conftest import f32_gate      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vec():
    d = np.load(os.path.join(HERE, "golden", "test_script_env_vectors.npz"))
    return {k: d[k] for k in d.files}


def test_vectors_hold_the_scripts_run(vec):
    assert vec["obs"].shape == (60, 56) and np.all(vec["action_in"] == 0.5) and np.all(vec["action_out"] == 0.5)
    assert vec["substeps"][0] == 26 and not vec["substeps"][1:].any() and not vec["done"].any()
    assert abs(float(vec["total_reward"]) + 0.042538) < 1e-6
    # the loop was left when checkFeedback's norm fell to 0.05 (snake.py:228-235)
    assert vec["servo_err"][0, 24] > 0.05 >= vec["servo_err"][0, 25]


def test_oracle_reproduces_the_script(vec, oracle_mod):
    e = oracle_mod.OracleEnv()
    e.reset()
    total = 0.0
    for i in range(60):
        assert np.array_equal(e.get_state(), vec["state"][i]), i
        o, r, d, k, _ = e.env_step(np.full(8, 0.5), vec_mode=False)
        assert (k, d) == (int(vec["substeps"][i]), bool(vec["done"][i])), i
        assert np.abs(o - vec["obs"][i]).max() < 1e-12 and abs(r - vec["reward"][i]) < 1e-12, i
        total += r
    assert abs(total - float(vec["total_reward"])) < 1e-12


def run_script_on(pkg_like):
    """test_script_env.py:8-22 in this file's words."""
    robot = pkg_like.Snake(None, "snake/snake.urdf")
    env = pkg_like.SnakeGymEnv(robot)
    obs = [env.reset()]
    R, out = 0.0, []
    for i in range(60):
        o, r, d, info = env.step([0.5] * 8)
        R += r
        assert env.render().size == 0 and not info
        out.append((np.array(o, dtype=np.float64), float(r), bool(d), int(robot.counter)))
    return obs[0], out, R, env


@pytest.mark.gpu
def test_product_classes_run_the_script(vec, pkg, oracle_mod):
    obs0, out, R, env = run_script_on(pkg)
    # yardstick: the float32 build of the oracle on the same 60 steps (the start pose is degenerate -- every cylinder flat
    # on the plane -- so float32 and float64 take different first contact points; tests/test_gait_test_golden.py)
    o32 = oracle_mod.OracleEnv(f32=True)
    o32.reset()
    c_obs = c_rew = 0.0
    for i in range(60):
        o, r, d, k, _ = o32.env_step(np.full(8, 0.5), vec_mode=False)
        c_obs = max(c_obs, np.abs(o - vec["obs"][i])[np.r_[0:32, 48:55]].max())
        c_rew = max(c_rew, abs(r - vec["reward"][i]))
    ks = [k for (_, _, _, k) in out]
    assert abs(ks[0] - 26) <= 1 and not any(ks[1:]) and not any(d for (_, _, d, _) in out)
    g_obs = max(np.abs(o - vec["obs"][i])[np.r_[0:32, 48:55]].max() for i, (o, _, _, _) in enumerate(out))
    g_rew = max(abs(r - vec["reward"][i]) for i, (_, r, _, _) in enumerate(out))
    print("test_script_env.py on the kernels: substeps %s..., total reward %.6f (recorded %.6f); |d obs| %.2e (float32 oracle %.2e), "
          "|d reward| %.2e (%.2e)" % (ks[:3], R, float(vec["total_reward"]), g_obs, c_obs, g_rew, c_rew))
    f32_gate("test_script_env.py: worst |d obs| of 60 steps", g_obs, c_obs, 1.5, 1e-4)
    f32_gate("test_script_env.py: worst |d reward|", g_rew, c_rew, 1.5, 1e-5)
    assert abs(R - float(vec["total_reward"])) < 60 * (1.5 * c_rew + 1e-5)
    env.close() if hasattr(env, "close") else None
