"""Opportunistic cross-check against a live PyBullet -- TEST INFRASTRUCTURE.

`import pybullet` fails in the image this was developed in, so nothing here has ever met a real PyBullet: that is
what "parity unpinned" means.  If a box ever has the wheel, `report()` drives it with this file's own
restatement of the reference's call sequence (snake.py:88-107 hard reset, :219-221 motor commands, :286 step,
:180-206 reads) on the URDF text oracle/urdf_gen.py generates, dumps the engine and per-link parameters the oracle's
[U] switches stand for, and compares single substeps and one gait env-step with the oracle.

    python oracle/pybullet_live.py            # prints "PyBullet not available" or the report (JSON)
"""
import json
import os
import sys
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)


def available():
    try:
        import pybullet  # noqa: F401
        return True
    except Exception:   # noqa: BLE001  (ImportError, or a broken wheel)
        return False


class LiveSnake(object):
    """The reference's world (snake.py:88-107) in a DIRECT-mode PyBullet, built from generated URDF text."""

    def __init__(self, n=16):
        import pybullet as p
        import urdf_gen
        self.p, self.n = p, n
        self.cid = p.connect(p.DIRECT)
        self.dir = tempfile.mkdtemp(prefix="snk_urdf_")
        for name, text in (("plane.urdf", urdf_gen.plane_urdf()), ("snake.urdf", urdf_gen.snake_urdf(n))):
            with open(os.path.join(self.dir, name), "w") as f:
                f.write(text)
        p.resetSimulation()
        p.setAdditionalSearchPath(self.dir)
        p.setGravity(0, 0, -9.8)                                            # snake.py:8,91
        self.plane = p.loadURDF("plane.urdf")
        self.snake = p.loadURDF("snake.urdf", [0, 0, 0], useFixedBase=0, flags=p.URDF_USE_SELF_COLLISION)   # snake.py:93
        fr = [1, 0.1, 0.01]                                                 # FRICTION_VALUES, snake.py:104-106
        p.changeDynamics(self.snake, -1, lateralFriction=2, anisotropicFriction=fr)
        for i in range(p.getNumJoints(self.snake)):
            p.changeDynamics(self.snake, i, lateralFriction=2, anisotropicFriction=fr)
            p.enableJointForceTorqueSensor(self.snake, i, 1)
        self.motors = list(range(3, p.getNumJoints(self.snake), 3))        # snake.py:80

    def close(self):
        self.p.disconnect(self.cid)

    def soft_reset(self):                                                   # snake.py:96-99,119-127
        self.p.resetBasePositionAndOrientation(self.snake, [0, 0, 0], [0, 0, 0, 1])
        for j in self.motors:
            self.p.resetJointState(self.snake, j, 0.0)

    def command(self, targets):                                            # snake.py:219-221
        self.p.setJointMotorControlArray(self.snake, self.motors, self.p.POSITION_CONTROL, list(targets),
                                         forces=[np.inf] * len(self.motors))

    def step(self):
        self.p.stepSimulation()                                            # snake.py:286

    def observation(self):                                                 # snake.py:180-217
        p = self.p
        js = [p.getJointState(self.snake, j) for j in self.motors]
        pos, orn = p.getBasePositionAndOrientation(self.snake)
        obs = np.zeros(3 * self.n + 8)
        obs[0:self.n] = [s[0] for s in js]
        obs[self.n:2 * self.n] = [s[1] for s in js]
        obs[2 * self.n:3 * self.n] = [s[3] for s in js]
        obs[3 * self.n:3 * self.n + 3] = pos
        obs[3 * self.n + 3:3 * self.n + 7] = orn
        obs[3 * self.n + 7] = p.getJointState(self.snake, 0)[2][2]
        return obs

    def parameters(self):
        """Appendix C-1: what the oracle's [U] switches stand for, as this PyBullet reports it."""
        p = self.p
        nj = p.getNumJoints(self.snake)
        out = {"version": getattr(p, "__version__", None), "api": p.getAPIVersion(),
               "engine": {k: (v if not isinstance(v, tuple) else list(v)) for k, v in p.getPhysicsEngineParameters().items()},
               "num_joints": nj, "links": [], "joints": []}
        for i in range(-1, nj):
            d = p.getDynamicsInfo(self.snake, i)
            out["links"].append({"index": i, "mass": d[0], "lateral_friction": d[1], "inertia_diag": list(d[2]),
                                 "inertial_pos": list(d[3]), "inertial_orn": list(d[4])})
        for i in range(nj):
            j = p.getJointInfo(self.snake, i)
            out["joints"].append({"index": j[0], "name": j[1].decode(), "type": j[2], "damping": j[6], "friction": j[7],
                                  "lower": j[8], "upper": j[9], "max_force": j[10], "max_velocity": j[11],
                                  "link": j[12].decode(), "axis": list(j[13]), "parent": j[16]})
        return out


def report(n=16, substeps=50):
    """Returns a dict; never raises for a missing PyBullet."""
    if not available():
        return {"pybullet": None, "note": "PyBullet not available on this box; parity with PyBullet stays unpinned "
                                          "(the CPU baseline and the tests use the build's own float64 oracle)"}
    import oracle as orc
    live = LiveSnake(n)
    ref = orc.OracleEnv(n_modules=n)
    out = {"pybullet": live.parameters()}
    # (ii)/(iii) of Appendix C-1: from the rest pose on the plane, one motor command, `substeps` substeps
    live.soft_reset()
    ref.reset()
    targets = np.zeros(n)
    targets[1::2] = 0.3
    live.command(targets)
    worst = np.zeros(3)
    for k in range(substeps):
        live.step()
        ref.substep(targets)
        a, b = live.observation(), ref.get_obs()
        worst = np.maximum(worst, [np.abs(a[:n] - b[:n]).max(), np.abs(a[n:2 * n] - b[n:2 * n]).max(),
                                   np.abs(a[3 * n:3 * n + 7] - b[3 * n:3 * n + 7]).max()])
    out["substeps"] = substeps
    out["max_abs_diff"] = {"q": worst[0], "qd": worst[1], "base_pose": worst[2]}
    live.close()
    return out


if __name__ == "__main__":
    print(json.dumps(report(), indent=1, default=float))
