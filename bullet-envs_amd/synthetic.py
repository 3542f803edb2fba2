"""The synthetic inputs the bench and the tests drive the path with: counter-based per-env gait phases,
the serpenoid action stream of snake_gait_test.py:65-67,86, and BASELINE configs[4]'s per-env plane friction.  Pure numpy;
bench.py loads this file by path, tests through the package."""
import numpy as np


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def env_phases(global_ids, seed=0):
    """phi_e = 2 pi u_e, u_e from a counter-based generator keyed by (seed, global env index); phi_0 = 0."""
    with np.errstate(over="ignore"):
        h = splitmix64(np.asarray(global_ids, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3))
    u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    phi = 2.0 * np.pi * u
    phi[np.asarray(global_ids) == 0] = 0.0
    return phi


def gait_actions(global_ids, j, A=8):
    """a[e,k] = -sin((2k+1) s + w t_j + phi_e), s=4, w=2, t_j = 0.1 j (snake_gait_test.py:65-67,86)."""
    k = np.arange(A)
    phi = env_phases(global_ids)
    return -np.sin((2 * k[None, :] + 1) * 4.0 + 2.0 * (0.1 * j) + phi[:, None])


def env_friction(global_ids, seed):
    """BASELINE configs[4]: per-env plane friction mu_e ~ U[0.5, 1.5), counter-based, keyed by (seed, global env)."""
    with np.errstate(over="ignore"):
        h = splitmix64(np.asarray(global_ids, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
                       + np.uint64(seed) * np.uint64(0x100000001B3))
    return 0.5 + (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
