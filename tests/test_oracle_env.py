"""Env-logic quirks of the reference, checked on the oracle."""
import numpy as np
import pytest

SC = np.pi / 6


def gait_action(j, A=8, phase=0.0):
    """BASELINE 'serpenoid gait actions' (snake_gait_test.py:65-67,86): a_k = -sin((2k+1) s + w t + phi)."""
    k = np.arange(A)
    return -np.sin((2 * k + 1) * 4.0 + 2.0 * (0.1 * j) + phase)


def test_obs_layout_and_reset(oracle_mod):
    e = oracle_mod.OracleEnv()
    o = e.reset()
    assert o.shape == (56,)
    assert np.all(o[:48] == 0) and np.all(o[48:51] == 0)
    assert np.all(o[51:55] == [0, 0, 0, 1]) and o[55] == 0


def test_zero_substep_step(oracle_mod):
    """F4: a step whose targets are already within 0.05 does no physics at all."""
    e = oracle_mod.OracleEnv()
    e.reset()
    s0 = e.get_state()
    o, r, d, k, _ = e.env_step(np.full(8, 0.01))       # ||0.01*pi/6 * 8 joints|| = 0.0148 < 0.05
    assert k == 0 and not d
    assert np.array_equal(e.get_state(), s0)
    assert r == 0.0


def test_counter_cap_41(oracle_mod):
    """F4: the loop breaks at counter > 40, i.e. after 41 substeps."""
    e = oracle_mod.OracleEnv(kp=0.001)                  # servo too weak to converge
    e.reset()
    _, _, _, k, _ = e.env_step(np.ones(8))
    assert k == 41


def test_action_scatter_and_clip(oracle_mod):
    """F6: 8 actions -> odd motor slots, scaled by pi/6; checkBound clips in place."""
    e = oracle_mod.OracleEnv()
    e.reset()
    a = np.array([2.0, -3.0, 0.5, 0.2, -0.2, 0.1, 0.0, 0.3])
    o, r, d, k, a_clipped = e.env_step(a)
    assert np.array_equal(a_clipped, np.clip(a, -1, 1))
    q = o[:16] if not d else None
    if q is not None:
        # even slots are commanded to 0 and stay near 0; odd slots head towards a*pi/6
        assert np.abs(q[0::2]).max() < 0.1
        assert np.sign(q[1]) == 1 and np.sign(q[3]) == -1
    e2 = oracle_mod.OracleEnv(gait=0)
    e2.reset()
    o2, _, d2, _, _ = e2.env_step(np.array([0.5] * 8))
    if not d2:
        assert np.abs(o2[1:16:2]).max() < 0.1 and o2[0] > 0.05


def test_termination_on_obs9(oracle_mod):
    """F8: done when |obs[9]| > 0.5; slot 9 is a driven joint whose range is +-pi/6 = 0.5236."""
    e = oracle_mod.OracleEnv()
    e.reset()
    done = False
    for j in range(5):                                  # canonical gait: slot 9 target is -sin(36+.2j)*pi/6
        o, r, done, k, _ = e.env_step(gait_action(j), vec_mode=False)
        if done:
            break
    assert done
    assert abs(o[9]) > 0.5                              # single-env mode returns the TERMINAL obs
    assert r < -4.0                                     # includes the -5
    s = e.get_state()
    assert np.all(s[13:] == 0) and np.all(s[:3] == 0)   # ... and has already soft-reset inside


def test_vec_mode_returns_post_reset_obs_and_refreshes_prev(oracle_mod):
    e = oracle_mod.OracleEnv()
    f = oracle_mod.OracleEnv()
    e.reset()
    f.reset()
    for j in range(5):
        oe, re, de, _, _ = e.env_step(gait_action(j), vec_mode=True)
        of, rf, df, _, _ = f.env_step(gait_action(j), vec_mode=False)
        assert re == rf and de == df
        if de:
            break
    assert de
    assert np.all(oe[:32] == 0) and np.all(oe[48:51] == 0)      # post-reset obs
    # stale caches survive the soft reset: torque and joint-0 force are the last substep's
    assert np.array_equal(oe[32:48], of[32:48]) and oe[55] == of[55]
    # prev-x: vec mode refreshed it to the reset obs (0), single mode keeps the terminal x
    assert e.get_aux()[2] == 0.0
    assert f.get_aux()[2] == of[48]


def test_reward_terms(oracle_mod):
    e = oracle_mod.OracleEnv()
    e.reset()
    prev_x = e.get_aux()[2]
    o, r, d, k, _ = e.env_step(gait_action(0))
    assert not d and k > 0
    energy = np.sum(o[16:32] * o[32:48] * 0.01)
    col = -10.0 if abs(o[55]) > 10 else 0.0
    want = 1.0 * (o[48] - prev_x) + col - 0.01 * abs(o[49]) - 0.1 * energy
    assert abs(r - want) < 1e-12


def test_gait_moves_forward(oracle_mod):
    """The serpenoid gait propels the snake along +x (reward's forward direction)."""
    e = oracle_mod.OracleEnv()
    e.reset()
    xs = []
    for j in range(3, 15):
        o, r, d, k, _ = e.env_step(gait_action(j), vec_mode=False)
        if d:
            break
        xs.append(o[48])
    assert len(xs) >= 8 and xs[-1] - xs[2] > 0.02


def test_height_termination(oracle_mod):
    """checkSnakeHeight: mean COM height of links {0,3,...,48} above 0.1 ends the step."""
    e = oracle_mod.OracleEnv()
    e.reset()
    s = e.get_state()
    s[2] = 0.5
    e.set_state(s)
    assert e.mean_height() > 0.1
    o, r, d, k, _ = e.env_step(np.ones(8))
    assert d and k == 1


@pytest.mark.parametrize("n", [32])
def test_long_chain(oracle_mod, n):
    e = oracle_mod.OracleEnv(n_modules=n)
    o = e.reset()
    assert o.shape == (3 * n + 8,)
    o, r, d, k, _ = e.env_step(gait_action(0, A=n // 2))
    assert np.all(np.isfinite(o)) and 1 <= k <= 41
