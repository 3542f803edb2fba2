"""Generates tests/golden/gait_test_vectors.npz by RUNNING the reference's own /root/reference/snake_gait_test.py::test()
-- the script behind SURVEY 8(f)-3: the snake under its serpenoid gait in the script's OWN world (time step 0.01, gravity
-9.81, motor force 4 N m, no self-collision flag, the 200-kg block of snake/block.urdf loaded free at [2, 0, 0.1]) and
its read-out `getJointState(robot, 3)[2][2]` ("the snake has hit the wall" when > 20, snake_gait_test.py:126).

The script talks to the module-level `pybullet` it imports as `p`; here `p` is `GaitClient` below, whose calls are
answered by the CPU oracle (the world is built from what the script's own setup calls say, the way
bullet-envs_amd/pybullet_client.py does for the product).  `time` is replaced by a clock that advances 0.01 s per call
(the script feeds wall-clock time into its gait: not reproducible otherwise) and `matplotlib.pyplot` by an empty module.
What is EXECUTED is the script's call sequence, its gait signal and its recording; what stepSimulation computes is the
oracle's restatement of Bullet ([U], parity unpinned).

Stored (arrays only): the 16 motor targets of every step, the recorded joint-3 reaction Fz, and -- from the client's
side -- the snake's state after every step and the parameters the script's calls implied.
tests/test_gait_test_golden.py replays the same calls on the product's BulletClient (CPU: the oracle behind it; GPU: the
HIP kernels).

Run here:  python tests/golden/make_gait_test_vectors.py [steps=2000]
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402

orc.build()
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000      # the script's own `test(2000)`


class GaitClient(types.ModuleType):
    """`import pybullet as p` for snake_gait_test.py: the calls of its test() on one oracle world."""
    GUI, DIRECT, POSITION_CONTROL, JOINT_REVOLUTE = 1, 2, 2, 0

    def __init__(self):
        super().__init__("pybullet")
        self.world = {}
        self.e = None
        self.calls = []
        self.states = []
        self.targets = None

    def _env(self):
        if self.e is None:
            self.e = orc.OracleEnv(**self.world)
            self.e.hard_reset()
            self.targets = np.zeros(self.e.n)
        return self.e

    def connect(self, mode):
        self.calls.append("connect")
        return 0

    def resetSimulation(self):
        self.world, self.e = {}, None

    def setAdditionalSearchPath(self, path):
        pass

    def loadURDF(self, name, basePosition=None, useFixedBase=0, flags=0):
        base = os.path.basename(name)
        if base == "plane.urdf":
            return 0
        if base == "snake.urdf":
            assert list(basePosition) == [0, 0, 0] and useFixedBase == 0
            self.world["self_collision"] = 1 if flags & 8 else 0      # the script passes no URDF_USE_SELF_COLLISION
            return 1
        if base == "block.urdf":
            self.world["obstacle"] = 1 if useFixedBase else 2
            self.world["obstacle_pos"] = [float(v) for v in basePosition]
            return 2
        raise AssertionError(name)

    def setGravity(self, x, y, z):
        assert x == 0 and y == 0
        self.world["gravity_z"] = float(z)

    def setTimeStep(self, dt):
        self.world["dt"] = float(dt)

    def setRealTimeSimulation(self, on):
        assert on == 0

    def resetDebugVisualizerCamera(self, **kw):
        pass

    def getCameraImage(self, width=0, height=0, **kw):
        return (width, height, [], [], [])

    def changeDynamics(self, body, link, lateralFriction=None, anisotropicFriction=None):
        if body == 1:
            self.world["mu_link"] = float(lateralFriction)
            self.world["aniso"] = [float(v) for v in anisotropicFriction]

    def getNumJoints(self, body):
        return 49

    def getJointInfo(self, body, i):
        return (i,)

    def enableJointForceTorqueSensor(self, body, i, on):
        pass

    def setJointMotorControlArray(self, body, joints, mode, targetPositions=None, forces=None):
        assert mode == self.POSITION_CONTROL and list(joints) == list(range(3, 49, 3))
        f = np.asarray(forces, dtype=np.float64)
        assert np.all(f == f[0])
        want = float(f[0]) * self.world.get("dt", 1.0 / 240.0)
        if self.e is None:
            self.world["max_motor_impulse"] = want
        assert abs(self.world["max_motor_impulse"] - want) < 1e-15
        self._env()
        self.targets = np.asarray(targetPositions, dtype=np.float64).copy()

    def stepSimulation(self):
        e = self._env()
        e.substep(self.targets)
        self.states.append((self.targets.copy(), e.get_state().copy(), e.get_box()[0].copy()))

    def getJointState(self, body, joint):
        e = self._env()
        nan = float("nan")
        fz = e.joint3_reaction_fz() if joint == 3 else nan
        j = joint // 3 - 1
        s = e.get_state()
        return (float(s[13 + j]), float(s[13 + e.n + j]), (nan, nan, fz, nan, nan, nan), float(e.get_aux()[0][j]))


class Clock(types.ModuleType):
    """time.time() advances by the script's own pacing (it sleeps 0.01 s per step); time.sleep() does nothing."""

    def __init__(self):
        super().__init__("time")
        self.t = 0.0

    def time(self):
        self.t += 0.01
        return self.t

    def sleep(self, s):
        pass


def main():
    client = GaitClient()
    pbd = types.ModuleType("pybullet_data")
    pbd.getDataPath = lambda: "pybullet_data"
    mpl = types.ModuleType("matplotlib")
    plt = types.ModuleType("matplotlib.pyplot")
    mpl.pyplot = plt
    sys.modules.update({"pybullet": client, "pybullet_data": pbd, "matplotlib": mpl, "matplotlib.pyplot": plt})
    sys.path.insert(0, "/root/reference")
    import snake_gait_test as ref  # noqa: E402
    ref.time = Clock()
    ref.motorList = np.arange(3, 51, 3)          # set under __main__ in the script (snake_gait_test.py:121)
    devnull = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, devnull     # (the script prints two shapes per step)
    try:
        states, torque = ref.test(STEPS, create_video=False, record_torque=True)
    finally:
        sys.stdout = stdout
    assert len(client.states) == STEPS and torque.shape == (STEPS, 16)
    signals = np.array(states[1:])               # states[0] is the script's initial [0] * 16
    tg = np.stack([s[0] for s in client.states])
    assert np.array_equal(signals, tg)
    fz3 = torque[:, 0]
    assert np.isnan(torque[:, 1:]).all()
    w = client.world
    out = os.path.join(HERE, "gait_test_vectors.npz")
    # (the full state for the first 200 steps -- what the parity tests replay --, then the head's and the box's positions)
    np.savez_compressed(out, targets=tg, joint3_fz=fz3, state=np.stack([s[1] for s in client.states[:200]]),
                        head_xyz=np.stack([s[1][:3] for s in client.states]).astype(np.float32),
                        box_xyz=np.stack([s[2][:3] for s in client.states]).astype(np.float32),
                        world_keys=np.array(sorted(w)), world_json=np.array(__import__("json").dumps(w, sort_keys=True)))
    hit = np.nonzero(fz3 > 20)[0]
    print("wrote", out, os.path.getsize(out), "bytes:", STEPS, "steps; world", w)
    print("joint-3 reaction Fz: min %.2f max %.2f; first > 20 at %s; head x after the run %.3f m"
          % (fz3.min(), fz3.max(), hit[0] if len(hit) else None, client.states[-1][1][0]))


if __name__ == "__main__":
    main()
