import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, so import through importlib)."""
    return importlib.import_module("bullet-envs_amd")


# The contact model of rounds 1 and 2 (still a switch): stateless two-point manifold on implicit cylinders, absolute
# 0.02-m breaking threshold.  The defaults since round 3 are Bullet's own: 32-gon hulls, persistent manifold, the
# dispatcher's relative threshold.
ROUND1 = dict(hull_sides=0, contact_model=0, relative_breaking_threshold=0)


def random_state(rng, n, z=0.2, qamp=0.5, vamp=1.0, flat=False):
    """A reachable random state [pos3, quat4, omega3, vel3, q n, qd n]."""
    s = np.zeros(13 + 2 * n)
    s[0:3] = rng.uniform(-0.1, 0.1, 3)
    s[2] = z
    if flat:
        yaw = rng.uniform(-np.pi, np.pi)
        s[3:7] = [0, 0, np.sin(yaw / 2), np.cos(yaw / 2)]
    else:
        qv = rng.normal(size=4)
        s[3:7] = qv / np.linalg.norm(qv)
    s[7:10] = rng.uniform(-vamp, vamp, 3)
    s[10:13] = rng.uniform(-vamp, vamp, 3)
    s[13:13 + n] = rng.uniform(-qamp, qamp, n)
    s[13 + n:] = rng.uniform(-vamp, vamp, n)
    return s


# ------------------------------------------------------------------------------------------------------------------
# Count / done mismatches against the reference's recorded env-steps: one rule for every test that meets them
# (ADVICE r5 medium).
# ------------------------------------------------------------------------------------------------------------------
def SERVO_WINDOW(k):
    """How far from the 0.05 tolerance the reference's servo error may sit where a float32 computation stops one substep
    earlier or later than the reference: float32 round-off in the error norm after k stiff substeps.  Calibrated on the
    float32 build of the ORACLE over the reference's 2240 recorded vector env-steps (test_float32_oracle_yardstick_...,
    CPU): its one-substep mismatches sit within 1.6e-3 of the tolerance at k <= 20 and 3.75e-3 at k = 31 (PPO step 24 env
    3: 30 substeps for the reference's 31, the float64 oracle unmoved by perturbations).  The window is ~1.4 x that
    (round 5 allowed 1.5e-3 + 2e-4 k: 7.7e-3 at k = 31)."""
    return 5e-4 + 1.5e-4 * k


def count_spread(oracle_mod, S, X, M, a, vec_mode, seed, **over):
    """Substep counts the FLOAT64 oracle gives for one env-step when its inputs are moved by float32-sized amounts: the
    state / contact cache / action rounded to float32 (what the GPU is handed), and four random relative perturbations
    of 6e-8.  More than one value = the step sits at a bifurcation (a contact about to stick or slip decides how fast the
    servo error decays): no float32 computation can be expected to land on the reference's count there.  Observed, e.g.,
    ARS step 26 env 4: 18 from the exact state, 20 from the rounded one, 21 from perturbed ones."""
    rng = np.random.default_rng(seed)
    ks = []
    e = oracle_mod.OracleEnv(**over)
    for trial in range(5):
        if trial == 0:
            S2, M2, a2 = (x.astype(np.float32).astype(np.float64) for x in (S, M, a))
        else:
            S2 = S * (1 + rng.uniform(-1, 1, S.shape) * 6e-8)
            M2 = M * (1 + rng.uniform(-1, 1, M.shape) * 6e-8)
            M2[:, 0] = M[:, 0]
            a2 = a * (1 + rng.uniform(-1, 1, a.shape) * 6e-8)
        e.hard_reset()
        e.sync(S2, X, M2)
        ks.append(e.env_step(a2.copy(), vec_mode=vec_mode)[3])
    return ks


def mismatch_gate(what, gpu, f32, factor=1.5, slack=4):
    """The GPU's count / done mismatches against the float32 oracle's, both counted over the same env-steps by the same
    rule: GPU <= factor x float32 oracle + slack."""
    print("  mismatch gate [%s]: GPU %d, float32 oracle %d, bound %.1f" % (what, gpu, f32, factor * f32 + slack))
    assert gpu <= factor * f32 + slack, (what, gpu, f32)


# ------------------------------------------------------------------------------------------------------------------
# The float32 tolerance, stated once and applied through ONE function: a GPU figure
# against the same figure of the oracle built in float32, both measured against the float64 oracle (or the reference's
# recorded run) on the same inputs in the same run.
#   factor   1.5  medians, and worst values over >= 30 samples of the 16-link chain
#            2.0  90th percentiles, anything over < 30 samples, the 32-link chain
#            a test that needs more says why where it calls this, and DESIGN.md 3's table lists it with the observed ratio
#   floor    the value below which the figure is round-off of the comparison itself (max(floor, factor x float32))
#   cap      the hard outer bound, whatever the float32 oracle does
# Every call prints one "GATE" line (pytest -s): profiles/r06_accuracy_calibration.txt is those lines.
# SNK_GATE_SOFT=1: print, do not assert (a calibration run).
# ------------------------------------------------------------------------------------------------------------------
def f32_gate(what, gpu, f32, factor=1.5, floor=0.0, cap=float("inf")):
    gpu, f32 = float(gpu), float(f32)
    limit = min(max(floor, factor * f32), cap)
    ratio = gpu / f32 if f32 > 0 else float("inf") if gpu > 0 else 0.0
    print("  GATE %-58s GPU %.3e | float32 oracle %.3e | ratio %5.2f | limit %.3e = min(max(%.1e, %.1f x), %.1e) %s"
          % (what, gpu, f32, ratio, limit, floor, factor, cap, "" if gpu < limit else "<-- OVER"))
    if not os.environ.get("SNK_GATE_SOFT"):
        assert gpu < limit, (what, gpu, f32, limit)
    return gpu < limit
