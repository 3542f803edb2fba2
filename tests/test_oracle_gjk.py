"""Pins the oracle's narrow phase (GJK distance between two cylinders, oracle/snake_oracle.cpp: gjk_distance) against a
brute-force answer that shares nothing with it: dense samples of both surfaces and the minimum pairwise distance.
Known answers first (side by side, stacked, rim to rim, crossed), then 24 random poses per shape with the second
cylinder 48-80 mm away (touching to 20 mm apart), implicit cylinders and the 32-gon hulls PyBullet imports them as.
The sampled distance is an upper bound of the true one, exact to the sampling step (0.45 mm around, 0.55 mm along):
tolerance 0.1 mm below, never above; intersecting surfaces must be reported as overlap."""
import numpy as np
import pytest

R_CYL, HL = 0.026, 0.0165


def _frame(c, R=np.eye(3)):
    return np.r_[np.asarray(c, float), np.asarray(R, float).ravel()]


def _rot(rng):
    q = rng.normal(size=4)
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _surface(c, R, hull, n_ang=360, n_len=61):
    if hull:
        th = 2 * np.pi * np.arange(hull) / hull
        vx, vy = R_CYL * np.sin(th), R_CYL * np.cos(th)                 # the importer's vertices (snake_oracle.cpp)
        t = np.linspace(0, 1, n_ang // hull + 1)[:-1]
        px = (vx[:, None] * (1 - t) + np.roll(vx, -1)[:, None] * t).ravel()
        py = (vy[:, None] * (1 - t) + np.roll(vy, -1)[:, None] * t).ravel()
    else:
        th = 2 * np.pi * np.arange(n_ang) / n_ang
        px, py = R_CYL * np.cos(th), R_CYL * np.sin(th)
    z = np.linspace(-HL, HL, n_len)
    side = np.stack([np.repeat(px, n_len), np.repeat(py, n_len), np.tile(z, px.size)], axis=1)
    rad = np.linspace(0, 1, 30)
    cap = np.stack([np.outer(rad, px).ravel(), np.outer(rad, py).ravel()], axis=1)
    loc = np.concatenate([side, np.c_[cap, np.full(len(cap), HL)], np.c_[cap, np.full(len(cap), -HL)]])
    return np.asarray(c) + loc @ np.asarray(R).T


def _brute(ca, Ra, cb, Rb, hull):
    A, B = _surface(ca, Ra, hull), _surface(cb, Rb, hull)
    d2 = ((A[::9, None, :] - B[None, ::9, :]) ** 2).sum(-1)
    ia, ib = np.unravel_index(np.argmin(d2), d2.shape)
    sa = A[np.linalg.norm(A - A[9 * ia], axis=1) < 0.008]
    sb = B[np.linalg.norm(B - B[9 * ib], axis=1) < 0.008]
    return float(np.sqrt(((sa[:, None, :] - sb[None, :, :]) ** 2).sum(-1).min()))


def test_gjk_known_answers(oracle_mod):
    e = oracle_mod.OracleEnv(n_modules=32)
    Rx = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]])
    cases = [(_frame([0.06, 0, 0]), 0.06 - 2 * R_CYL),                                   # side by side
             (_frame([0, 0, 0.05]), 0.05 - 2 * HL),                                       # stacked, cap to cap
             (_frame([0.06, 0, 0.05]), np.hypot(0.06 - 2 * R_CYL, 0.05 - 2 * HL)),        # rim to rim
             (_frame([0.06, 0, 0], Rx), 0.06 - 2 * R_CYL)]                                # crossed axes
    for fb, want in cases:
        d, pa, pb = e.debug_gjk(_frame([0, 0, 0]), fb)
        assert abs(d - want) < 1e-9 and abs(np.linalg.norm(pa - pb) - d) < 1e-9
    assert e.debug_gjk(_frame([0, 0, 0]), _frame([0.03, 0, 0.01]))[0] < 0                 # overlapping


@pytest.mark.parametrize("hull", [0, 32])
def test_gjk_distance_matches_brute_force(oracle_mod, hull):
    e = oracle_mod.OracleEnv(n_modules=32, hull_sides=hull)
    rng = np.random.default_rng(5 + hull)
    apart = touching = 0
    for _ in range(24):
        RA, RB = _rot(rng), _rot(rng)
        cb = rng.normal(size=3)
        cb *= (0.048 + 0.032 * rng.uniform()) / np.linalg.norm(cb)
        d, pa, pb = e.debug_gjk(_frame([0, 0, 0], RA), _frame(cb, RB))
        br = _brute(np.zeros(3), RA, cb, RB, hull)
        if d < 0:
            assert br < 4e-4, (d, br)            # surfaces that intersect: the samples come within the sampling step
            touching += 1
            continue
        assert -1e-4 < br - d < 6e-4, (d, br)    # samples can only be farther apart than the true closest points
        assert abs(np.linalg.norm(pa - pb) - d) < 1e-9
        apart += 1
    assert apart >= 14 and touching >= 1
