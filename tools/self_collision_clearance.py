"""How close do the snake's own collision cylinders get?  (the reference loads the URDF with
URDF_USE_SELF_COLLISION, snake.py:93; this build has ground contacts only.)

Brute force on the URDF's geometry (every module = INPUT_IF cylinder, revolute about y at
z = 0.0366, OUTPUT_BODY cylinder, fixed joint (0, 0, 0.0273) rpy (0, 0, -1.57075); cylinders r = 0.026,
L = 0.033 centred at z = 0.0183 of their link, collision margin 0.001): sample every cylinder's surface, place
the chain by forward kinematics for a joint-angle pattern, and take the minimum distance over all pairs Bullet
would test (every pair except direct parent-child links).  A contact row exists in Bullet when a pair is closer
than the 0.02 breaking threshold; it can only act when the gap closes within one time step.

    python tools/self_collision_clearance.py
"""
import itertools
import numpy as np
from scipy.spatial import cKDTree

R, L, ZC, MARGIN = 0.026, 0.033, 0.0183, 0.001
PIVOT, NEXT, YAW = 0.0366, 0.0273, -1.57075


def cyl_points(n_ring=96, n_len=12, n_rad=6):
    """Surface samples of one cylinder in its link frame (side + both caps), inflated by the margin."""
    th = np.linspace(0, 2 * np.pi, n_ring, endpoint=False)
    r = R + MARGIN
    z = np.linspace(ZC - L / 2 - MARGIN, ZC + L / 2 + MARGIN, n_len)
    side = np.array([[r * np.cos(t), r * np.sin(t), zz] for zz in z for t in th])
    caps = np.array([[rr * np.cos(t), rr * np.sin(t), zz] for zz in (z[0], z[-1])
                     for rr in np.linspace(0, r, n_rad) for t in th])
    return np.vstack([side, caps])


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def chain(q):
    """World (= first INPUT_IF frame) poses of the 2n cylinder links: I1, O1, I2, O2, ..."""
    Rw, p = np.eye(3), np.zeros(3)
    out = []
    for qi in q:
        out.append((Rw.copy(), p.copy()))                    # INPUT_IF
        p = p + Rw @ np.array([0, 0, PIVOT])
        Rw = Rw @ rot_y(qi)
        out.append((Rw.copy(), p.copy()))                    # OUTPUT_BODY
        p = p + Rw @ np.array([0, 0, NEXT])
        Rw = Rw @ rot_z(YAW)
    return out


def min_clearance(q, pts):
    poses = chain(q)
    clouds = [pts @ Rw.T + p for Rw, p in poses]
    trees = [cKDTree(c) for c in clouds]
    best = (np.inf, None)
    for i, j in itertools.combinations(range(len(poses)), 2):
        if j == i + 1:
            continue                                          # direct parent-child: excluded by Bullet
        if np.linalg.norm(poses[i][1] - poses[j][1]) > 0.2:
            continue
        d = trees[i].query(clouds[j])[0].min()
        if d < best[0]:
            best = (d, (i, j))
    return best


def name(k):
    return ("I%d" if k % 2 == 0 else "O%d") % (k // 2 + 1)


if __name__ == "__main__":
    pts = cyl_points()
    res = 2 * np.pi * (R + MARGIN) / 96
    print("surface sampling %.1f mm; distances below are upper bounds good to about that" % (res * 1e3))
    n = 16
    print("\none joint bent, the others straight (pairs across that joint):")
    for deg in (0, 10, 17.5, 30, 45, 60, 75, 90):
        q = np.zeros(n); q[7] = np.radians(deg)
        d, (i, j) = min_clearance(q, pts)
        print("  %5.1f deg: closest pair %s-%s  %.1f mm" % (deg, name(i), name(j), d * 1e3))
    print("\ncommand range (|target| <= 30 deg, SCALING_FACTOR = pi/6, snake.py:41,63):")
    rng = np.random.default_rng(0)
    worst = np.inf
    pats = [np.full(n, np.radians(30)), np.tile([np.radians(30), -np.radians(30)], n // 2),
            np.where(np.arange(n) % 2 == 1, np.radians(30), 0.0), np.where(np.arange(n) % 2 == 0, np.radians(30), 0.0)]
    pats += [rng.uniform(-1, 1, n) * np.radians(30) for _ in range(40)]
    pats += [rng.choice([-1.0, 1.0], n) * np.radians(30) for _ in range(40)]
    for q in pats:
        worst = min(worst, min_clearance(q, pts)[0])
    print("  16 links, %d patterns (all +30, alternating, yaw-only, pitch-only, random, random corners): min clearance %.1f mm"
          % (len(pats), worst * 1e3))
    for n in (16, 32):
        # the yaw axes alternate in sign along the chain (Appendix B), so a planar coil is +30, -30, +30, ... on the yaw joints
        q = np.zeros(n); q[1::4] = np.radians(30); q[3::4] = -np.radians(30)
        d, (i, j) = min_clearance(q, pts)
        print("  %d links, planar coil (every yaw joint bent 30 deg the same way): closest pair %s-%s  %.1f mm"
              % (n, name(i), name(j), d * 1e3))
    print("\njoint limits (|q| <= 1.57): every joint at the limit, same sign, 16 links:")
    d, (i, j) = min_clearance(np.full(16, 1.57), pts)
    print("  closest pair %s-%s  %.1f mm" % (name(i), name(j), d * 1e3))
    q = np.where(np.arange(16) % 2 == 1, 1.57, 0.0)
    d, (i, j) = min_clearance(q, pts)
    print("  yaw joints at the limit, pitch straight: closest pair %s-%s  %.1f mm" % (name(i), name(j), d * 1e3))
