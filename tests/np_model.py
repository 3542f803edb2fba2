"""Independent numpy restatement of the snake model (test helper, not product code).

A third derivation of the same constants (snake/snake.urdf), written
in world-frame classical mechanics: link Jacobians -> mass matrix / momentum.  It shares
no code with oracle/ (link-coordinate spatial algebra) or with the HIP kernels (merged
composite bodies), so agreement between the three is evidence, not tautology.
"""
import numpy as np

M_LINK = 0.103
I_FILE = np.array([5.4796e-05, 5.4796e-05, 3.4814e-05])


def rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def rot_y(q):
    c, s = np.cos(q), np.sin(q)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def quat_to_mat(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def inertia_rules(margin=0.001, default_mass=1.0, from_file=False):
    """[U] Bullet import rules as restated in DESIGN.md §3."""
    if from_file:
        return I_FILE.copy(), np.ones(3)
    hx = 0.026 + 3 * margin
    hz = 0.033 / 2 + 3 * margin
    lx, lz = 2 * hx, 2 * hz
    I_cyl = M_LINK / 12.0 * np.array([lx * lx + lz * lz, lx * lx + lz * lz, 2 * lx * lx])
    le = 2 * margin
    I_empty = default_mass / 12.0 * 2 * le * le * np.ones(3)
    return I_cyl, I_empty


def build_tree(n, from_file=False, default_mass=1.0):
    """List of link dicts in Bullet DFS order; index 0 is the root (Bullet link -1)."""
    I_cyl, I_empty = inertia_rules(from_file=from_file, default_mass=default_mass)
    links = [dict(name="kdl_dummy_root", parent=-1, R=np.eye(3), p=np.zeros(3), rev=False, dof=-1,
                  m=default_mass, c=np.zeros(3), I=I_empty)]
    links.append(dict(name="base", parent=0, R=rpy(0, -1.57079632679, 0), p=np.array([0, 0, 0.026]),
                      rev=False, dof=-1, m=default_mass, c=np.zeros(3), I=I_empty))
    for k in range(1, n + 1):
        if k == 1:
            par, R, p = 1, np.eye(3), np.zeros(3)
        else:
            par, R, p = len(links) - 1, rpy(0, 0, -1.57075), np.array([0, 0, 0.0273])
        idx_in = len(links)
        links.append(dict(name="IN%d" % k, parent=par, R=R, p=p, rev=False, dof=-1, m=M_LINK,
                          c=np.array([0, 0, 0.0366]), I=I_cyl, cyl=np.array([0, 0, 0.0183])))
        links.append(dict(name="COLLAR%d" % k, parent=idx_in, R=np.eye(3), p=np.zeros(3), rev=False, dof=-1,
                          m=default_mass, c=np.zeros(3), I=I_empty))
        links.append(dict(name="OUT%d" % k, parent=idx_in, R=np.eye(3), p=np.array([0, 0, 0.0366]), rev=True,
                          dof=k - 1, m=M_LINK, c=np.zeros(3), I=I_cyl, cyl=np.array([0, 0, 0.0183])))
    return links


def fk(links, pos, quat, q):
    Rw = [quat_to_mat(np.asarray(quat, float))]
    ow = [np.asarray(pos, float)]
    for i in range(1, len(links)):
        k = links[i]
        R = k["R"] @ rot_y(q[k["dof"]]) if k["rev"] else k["R"]
        Rw.append(Rw[k["parent"]] @ R)
        ow.append(ow[k["parent"]] + Rw[k["parent"]] @ k["p"])
    return Rw, ow


def link_jacobians(links, Rw, ow):
    """Per link (Jw, Jv_com): 3 x (6+n), generalized velocity = [omega_w, v_w(root origin), qd]."""
    n = sum(1 for k in links if k["rev"])
    nd = 6 + n
    out = []
    for i, k in enumerate(links):
        cw = ow[i] + Rw[i] @ k["c"]
        Jw = np.zeros((3, nd))
        Jv = np.zeros((3, nd))
        Jw[:, 0:3] = np.eye(3)
        Jv[:, 3:6] = np.eye(3)
        r = cw - ow[0]
        Jv[:, 0:3] = -np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        j = i
        while j > 0:
            kj = links[j]
            if kj["rev"]:
                a = Rw[j] @ np.array([0.0, 1.0, 0.0])
                Jw[:, 6 + kj["dof"]] = a
                Jv[:, 6 + kj["dof"]] = np.cross(a, cw - ow[j])
            j = kj["parent"]
        out.append((Jw, Jv, cw))
    return out


def mass_matrix(links, pos, quat, q):
    Rw, ow = fk(links, pos, quat, q)
    J = link_jacobians(links, Rw, ow)
    nd = J[0][0].shape[1]
    M = np.zeros((nd, nd))
    for i, k in enumerate(links):
        Jw, Jv, _ = J[i]
        Iw = Rw[i] @ np.diag(k["I"]) @ Rw[i].T
        M += k["m"] * Jv.T @ Jv + Jw.T @ Iw @ Jw
    return M


def momentum(links, pos, quat, q, gvel):
    """Total linear momentum, angular momentum about the world origin, kinetic energy."""
    Rw, ow = fk(links, pos, quat, q)
    J = link_jacobians(links, Rw, ow)
    P = np.zeros(3)
    Lw = np.zeros(3)
    K = 0.0
    for i, k in enumerate(links):
        Jw, Jv, cw = J[i]
        w = Jw @ gvel
        v = Jv @ gvel
        Iw = Rw[i] @ np.diag(k["I"]) @ Rw[i].T
        P += k["m"] * v
        Lw += Iw @ w + np.cross(cw, k["m"] * v)
        K += 0.5 * k["m"] * v @ v + 0.5 * w @ Iw @ w
    return P, Lw, K


def gravity_force(links, pos, quat, q, gz=-9.8):
    Rw, ow = fk(links, pos, quat, q)
    J = link_jacobians(links, Rw, ow)
    f = np.zeros(J[0][0].shape[1])
    for i, k in enumerate(links):
        f += J[i][1].T @ np.array([0, 0, gz * k["m"]])
    return f
