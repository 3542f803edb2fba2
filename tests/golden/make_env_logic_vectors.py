"""Generates tests/golden/env_logic_vectors.npz by RUNNING the reference's own env-logic code:

    /root/reference/snake.py                      Snake.step / reset / createAction / checkFeedback / getObservation ...
    /root/reference/SnakeGymEnv.py                SnakeGymEnv.step / reset / checkBound / calculateReward / checkTermination
    /root/reference/ppo/multiprocessing_env.py    worker() -- the SubprocVecEnv auto-reset loop

The reference takes its physics engine as an INJECTED client object (snake.py:14-18; SURVEY 8(b) "inner seam") and
contains no dynamics of its own.  Here that client is `OracleClient` below: the ~16 PyBullet calls the path makes
(snake.py:79-286), answered by the CPU oracle (oracle/, one substep per stepSimulation).  So every line of the
reference's Python that orders the calls, clips, scatters, counts substeps, packs the observation, computes reward /
termination and resets is EXECUTED, not restated; what stepSimulation itself computes stays the oracle's restatement of
Bullet ([U], parity unpinned: DESIGN.md 3).  The module-level `import pybullet / pybullet_data / gym` of those files are
satisfied by empty stand-in modules: the path never calls into them (gym only for the base class and spaces.Box).

Stored per env-step (arrays only, no reference text): the simulator state the step started from (state, aux with
prev_x = the reference's `_observation[48]`, contact cache), the action as given and as the reference left it (clipped
in place), observation / reward / done / substep count as the reference returned them, and for test mode the
per-substep telemetry.  tests/test_env_logic_golden.py checks that oracle/'s own orc_env_step reproduces them (CPU) and
that the fused HIP kernel does (GPU).

Run here (the reference does not exist on the GPU box):  python tests/golden/make_env_logic_vectors.py
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

# ---- stand-ins for the three third-party imports (never called on this path) -------------------------------------


class _Box(object):
    def __init__(self, low, high):
        self.low, self.high = np.asarray(low), np.asarray(high)
        self.shape = self.low.shape


_pb = types.ModuleType("pybullet")
_pbd = types.ModuleType("pybullet_data")
_pbd.getDataPath = lambda: "pybullet_data"
_gym = types.ModuleType("gym")
_gym.Env = object
_gym.spaces = types.SimpleNamespace(Box=_Box)
sys.modules.setdefault("pybullet", _pb)
sys.modules.setdefault("pybullet_data", _pbd)
sys.modules.setdefault("gym", _gym)

sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/ppo")
import snake as ref_snake  # noqa: E402
import SnakeGymEnv as ref_env  # noqa: E402
import multiprocessing_env as ref_mp  # noqa: E402

ref_snake.time.sleep = lambda s: None       # snake.py:296 sleeps 10 ms per substep (pacing for the GUI)


# ---- the injected client ----------------------------------------------------------------------------------------
class OracleClient(object):
    """The PyBullet calls of snake.py:79-286 on one oracle environment.  Bullet link / joint index i is the oracle's
    link i + 1 (its link 0 is the floating base `kdl_dummy_root`)."""
    URDF_USE_SELF_COLLISION = 8
    POSITION_CONTROL = 2

    def __init__(self, **params):
        self.e = orc.OracleEnv(**params)
        self.n = self.e.n
        self.targets = np.zeros(self.n)
        self.calls = {}
        self.servo_err = []
        self.snake_id = None

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # world building (snake.py:88-107)
    def resetSimulation(self):
        self._count("resetSimulation")
        self.e.hard_reset()
        self.targets[:] = 0

    def setAdditionalSearchPath(self, path):
        self._count("setAdditionalSearchPath")

    def setGravity(self, x, y, z):
        assert (x, y) == (0, 0) and z == self.e.params.gravity_z, (x, y, z)

    def loadURDF(self, name, basePosition=None, useFixedBase=0, flags=0):
        self._count("loadURDF")
        if name == "plane.urdf":
            return 0
        assert basePosition is None or list(basePosition) == [0, 0, 0]
        assert useFixedBase == 0 and flags == self.URDF_USE_SELF_COLLISION
        self.snake_id = 1
        return 1

    def changeDynamics(self, body, link, lateralFriction=None, anisotropicFriction=None):
        assert body == self.snake_id and lateralFriction == self.e.params.mu_link
        assert list(anisotropicFriction) == list(self.e.params.aniso)

    def enableJointForceTorqueSensor(self, body, joint, on):
        assert on == 1

    def getNumJoints(self, body):
        return self.e.L - 1

    # reset (snake.py:119-127)
    def resetBasePositionAndOrientation(self, body, pos, orn):
        s = self.e.get_state()
        s[0:3] = pos
        s[3:7] = orn
        s[7:13] = 0.0                      # [U] zeroes the base twist
        self.e.set_state(s)

    def resetJointState(self, body, joint, value):
        assert joint % 3 == 0 and 3 <= joint <= 3 * self.n
        j = joint // 3 - 1
        s = self.e.get_state()
        s[13 + j] = value
        s[13 + self.n + j] = 0.0
        self.e.set_state(s)

    # the substep (snake.py:219-221, 286)
    def setJointMotorControlArray(self, body, joints, mode, targetPositions, forces=None):
        assert mode == self.POSITION_CONTROL and list(joints) == list(range(3, 3 * self.n + 1, 3))
        # forces=[MAX_TORQUE]*16 = inf on the reference's path (snake.py:26-27); scenario 9/10 lower Snake.forces to
        # reach the 41-substep cap, and the oracle behind this client was created with the matching impulse bound
        mi = self.e.params.max_motor_impulse
        assert forces is not None and all((f == np.inf and mi == np.inf) or abs(f * self.e.params.dt - mi) < 1e-15 for f in forces)
        self.targets = np.asarray(targetPositions, dtype=np.float64).copy()

    def stepSimulation(self):
        self._count("stepSimulation")
        self.e.substep(self.targets)
        # (for the tests' boundary checks: the servo error the reference's checkFeedback sees after this substep)
        self.servo_err.append(float(np.linalg.norm(self.targets - self.e.get_state()[13:13 + self.n])))

    # read-out (snake.py:130-146, 180-206, 237-245)
    def getJointState(self, body, joint):
        o = self.e.get_obs()
        n = self.n
        if joint == 0:
            return (0.0, 0.0, (0.0, 0.0, float(o[3 * n + 7]), 0.0, 0.0, 0.0), 0.0)
        j = joint // 3 - 1
        return (float(o[j]), float(o[n + j]), (0.0,) * 6, float(o[2 * n + j]))

    def getBasePositionAndOrientation(self, body):
        o = self.e.get_obs()
        n = self.n
        return tuple(o[3 * n:3 * n + 3]), tuple(o[3 * n + 3:3 * n + 7])

    def getLinkStates(self, body, indices):
        # [0] of a link state is the world position of the link's COM -- the only column this path reads
        # (snake.py:144, 241).  PyBullet's real tuples are ragged (3- and 4-vectors) and snake.py:143 turns them into
        # an array, which numpy >= 1.24 refuses; the orientation column is therefore filled with a 3-vector placeholder.
        com = self.e.link_com_world()
        return [(tuple(com[int(i) + 1]), (0.0, 0.0, 0.0)) for i in indices]


def make_args(gait, mode):
    """The fields Snake.setParams (snake.py:34-53) and SnakeGymEnv.__init__ (SnakeGymEnv.py:7-12) read; values =
    ppo/params.py's defaults."""
    return types.SimpleNamespace(
        selfCollisionEnabled=True, motorVelocityLimit=np.inf, motorTorqueLimit=np.inf, kp=10, kd=0.1,
        gaitSelection=gait, scaling_factor=6, cam_dist=5.0, cam_yaw=50, cam_pitch=-35, cam_roll=0, upAxisIndex=2,
        render_height=720, render_width=1280, fov=60, nearVal=0.1, farVal=100, mode=mode,
        alpha=1, beta=0.01, gamma=0.1)


class Recorder(object):
    def __init__(self):
        self.rows = []

    def pre(self, client, env):
        e = client.e
        tau, fz, _ = e.get_aux()
        prev_x = float(env._observation[3 * e.n]) if hasattr(env, "_observation") else 0.0
        return dict(state=e.get_state(), aux=np.concatenate([tau, [fz, prev_x]]), manifold=e.get_manifold())

    def add(self, scen, pre, a_in, a_out, obs, rew, done, k, vec, gait, mode, telem=None, mmi=np.inf, err=()):
        assert len(err) == k, (len(err), k)
        self.rows.append(dict(scen=scen, pre=pre, mmi=float(mmi), err=np.pad(np.array(err, dtype=np.float64), (0, 41 - len(err))), a_in=np.array(a_in, dtype=np.float64).reshape(-1),
                              a_out=np.array(a_out, dtype=np.float64).reshape(-1), obs=np.array(obs, dtype=np.float64),
                              rew=float(rew), done=bool(done), k=int(k), vec=int(vec), gait=int(gait),
                              mode=1 if mode == "test" else 0, telem=telem))


def new_env(gait=1, mode="train", use_args=False, params=None):
    client = OracleClient(gait=gait, **(params or {}))
    args = make_args(gait, mode) if (use_args or gait != 1 or mode != "train") else None
    robot = ref_snake.Snake(client, "snake/snake.urdf", args)
    env = ref_env.SnakeGymEnv(robot, args)
    return client, robot, env


def gait_action(j, phi, A=8):
    k = np.arange(A)
    return -np.sin((2 * k + 1) * 4.0 + 2.0 * (0.1 * j) + phi)


def run_single(rec, scen, actions, gait=1, mode="train", setup=None, use_args=False, params=None):
    """SnakeGymEnv.step called directly (the eval seam: terminal obs on done, stale _observation)."""
    client, robot, env = new_env(gait, mode, use_args, params)
    env.reset()
    if setup:
        setup(client, env)
    for a in actions:
        a = np.array(a, dtype=np.float64)
        pre = rec.pre(client, env)
        a_in = a.copy()
        client.servo_err = []
        obs, rew, done, info = env.step(a)
        telem = None
        if mode == "test":
            io = np.array(info["internal_observations"]).reshape(-1, 3 * client.n + 8)
            lp = np.array(info["link_positions"]).reshape(-1, 3 * (client.n + 1))
            assert len(io) == robot.counter == len(lp) and info["frames"] == []
            telem = (io, lp)
        else:
            assert info == {}
        rec.add(scen, pre, a_in, a, obs, rew, done, robot.counter, 0, gait, mode, telem, client.e.params.max_motor_impulse,
                err=list(client.servo_err))
    return client


class FakeRemote(object):
    """Stands in for the worker's end of the Pipe (multiprocessing_env.py:7-29)."""

    def __init__(self, cmds, on_step):
        self.cmds = list(cmds)
        self.sent = []
        self.on_step = on_step

    def recv(self):
        cmd = self.cmds.pop(0)
        if cmd[0] == "step":
            self.on_step(cmd[1])
        return cmd

    def send(self, x):
        self.sent.append(x)

    def close(self):
        pass


def run_worker(rec, scen, actions, gait=1, setup=None, params=None):
    """The SubprocVecEnv worker loop itself (auto-reset: post-reset obs, reward with the -5, done True)."""
    client, robot, env = new_env(gait, params=params)
    if setup:
        env.reset()
        setup(client, env)
    pres, a_ins = [], []

    def on_step(a):
        pres.append(rec.pre(client, env))
        a_ins.append(np.array(a, dtype=np.float64).copy())

    acts = [np.array(a, dtype=np.float64) for a in actions]
    cmds = ([] if setup else [("reset", None)]) + [("step", a) for a in acts] + [("get_spaces", None), ("close", None)]
    remote = FakeRemote(cmds, on_step)
    counters = []
    real_step = env.step

    errs = []

    def step_and_count(a):
        client.servo_err = []
        out = real_step(a)
        counters.append(robot.counter)
        errs.append(list(client.servo_err))
        return out
    env.step = step_and_count
    ref_mp.worker(remote, FakeRemote([], None), types.SimpleNamespace(x=lambda: env))
    outs = remote.sent[(0 if setup else 1):-1]
    ospace, aspace = remote.sent[-1]
    assert ospace.shape == (3 * client.n + 8,) and aspace.shape == (len(acts[0]),)
    assert len(outs) == len(acts) == len(counters)
    for a, a_in, pre, (obs, rew, done, info), k, er in zip(acts, a_ins, pres, outs, counters, errs):
        assert info == {}
        rec.add(scen, pre, a_in, a, obs, rew, done, k, 1, gait, "train", None, client.e.params.max_motor_impulse, err=er)
    return client


def main():
    rec = Recorder()
    rng = np.random.default_rng(2024)
    # 0-2: the bench's serpenoid gait through the worker loop, three phases (natural obs[9] terminations included)
    for i, phi in enumerate((0.0, 1.3, 2.9)):
        run_worker(rec, i, [gait_action(j, phi) for j in range(24)])
    # 3-4: the same gait through SnakeGymEnv.step directly (terminal obs, stale _observation at SnakeGymEnv.py:41-42)
    for i, phi in enumerate((0.4, 2.2)):
        run_single(rec, 3 + i, [gait_action(j, phi) for j in range(24)])
    # 5-6: out-of-range actions: checkBound clips the caller's array in place (SnakeGymEnv.py:82-88)
    wild = [rng.uniform(-2.5, 2.5, 8) for _ in range(5)]
    run_single(rec, 5, wild)
    run_worker(rec, 6, wild)
    # 7: targets already reached -> the servo loop is never entered (0 substeps); then a step back to zero
    run_single(rec, 7, [np.zeros(8), 0.3 * np.ones(8), np.zeros(8)])

    # 8: ... with a stale joint-0 force above 10 in the sensor cache: reward -10 on a 0-substep step
    def stale_force(client, env):
        tau, _, px = client.e.get_aux()
        client.e.set_aux(tau, 15.0, px)
    run_single(rec, 8, [np.zeros(8)], setup=stale_force)

    # 9-10: motors too weak to reach their targets: the loop stops at counter > 40 (41 substeps, snake.py:303)
    weak = 2e-5

    def weak_motors(client, env):
        env.robot.forces = [weak / client.e.params.dt] * env.robot.numMotors      # Snake.forces (snake.py:27)
    run_single(rec, 9, [np.ones(8), np.zeros(8)], setup=weak_motors, params=dict(max_motor_impulse=weak))
    run_worker(rec, 10, [np.ones(8), np.zeros(8)], setup=weak_motors, params=dict(max_motor_impulse=weak))

    # 11-12: snake in the air: checkSnakeHeight ends the loop after one substep, endDue2Height, done, -5
    def lifted(client, env):
        s = client.e.get_state()
        s[2] = 0.5
        client.e.set_state(s)
    run_single(rec, 11, [0.5 * np.ones(8), 0.2 * np.ones(8)], setup=lifted)
    run_worker(rec, 12, [0.5 * np.ones(8), 0.2 * np.ones(8)], setup=lifted)
    # 13-14: gait 0 (even slots) and identity (16 actions) through setParams (snake.py:34-53, 247-269)
    run_single(rec, 13, [gait_action(j, 0.7) for j in range(6)], gait=0)
    run_single(rec, 14, [0.6 * gait_action(j, 0.9, A=16) for j in range(6)], gait=2)
    # 15: test mode: per-substep observations and link positions in info (SnakeGymEnv.py:43-44, snake.py:292-293)
    run_single(rec, 15, [gait_action(j, 1.9) for j in range(4)], mode="test")
    # 16: default parameters given through an args object instead of defaultParams (same numbers: ppo/params.py)
    run_single(rec, 16, [gait_action(j, 0.0) for j in range(3)], use_args=True)

    R = rec.rows
    n, O, T = 16, 56, 41
    d = {
        "scenario": np.array([r["scen"] for r in R], dtype=np.int32),
        "vec_mode": np.array([r["vec"] for r in R], dtype=np.int32),
        "gait": np.array([r["gait"] for r in R], dtype=np.int32),
        "test_mode": np.array([r["mode"] for r in R], dtype=np.int32),
        "max_motor_impulse": np.array([r["mmi"] for r in R]),
        "state": np.stack([r["pre"]["state"] for r in R]),
        "aux": np.stack([r["pre"]["aux"] for r in R]),
        "manifold": np.stack([r["pre"]["manifold"] for r in R]),
        "act_dim": np.array([len(r["a_in"]) for r in R], dtype=np.int32),
        "action_in": np.stack([np.pad(r["a_in"], (0, 16 - len(r["a_in"]))) for r in R]),
        "action_out": np.stack([np.pad(r["a_out"], (0, 16 - len(r["a_out"]))) for r in R]),
        "obs": np.stack([r["obs"] for r in R]),
        "reward": np.array([r["rew"] for r in R]),
        "done": np.array([r["done"] for r in R], dtype=np.bool_),
        "substeps": np.array([r["k"] for r in R], dtype=np.int32),
        # servo error ||target - q|| after substep 1 .. k of the step (what checkFeedback compares with 0.05, snake.py:228-235)
        "servo_err": np.stack([r["err"] for r in R]),
    }
    tel_rows = [i for i, r in enumerate(R) if r["telem"] is not None]
    io = np.zeros((len(tel_rows), T, O))
    lp = np.zeros((len(tel_rows), T, 3 * (n + 1)))
    for t, i in enumerate(tel_rows):
        a, b = R[i]["telem"]
        io[t, :len(a)] = a
        lp[t, :len(b)] = b
    d["telemetry_rows"] = np.array(tel_rows, dtype=np.int32)
    d["internal_observations"] = io
    d["link_positions"] = lp
    out = os.path.join(HERE, "env_logic_vectors.npz")
    np.savez_compressed(out, **d)
    k = d["substeps"]
    print("wrote", out, os.path.getsize(out), "bytes:", len(R), "env-steps,", int(d["done"].sum()), "done,",
          "substeps min/max", int(k.min()), int(k.max()), "| 0-substep steps", int((k == 0).sum()), "| 41-substep steps",
          int((k == 41).sum()), "| clipped", int((np.abs(d["action_in"]) > 1).any(axis=1).sum()))


if __name__ == "__main__":
    main()
