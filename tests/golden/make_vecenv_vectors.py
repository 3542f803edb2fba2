"""Generates tests/golden/vecenv_vectors.npz by RUNNING the reference's OUTER seam and the trainers' own loops:

    /root/reference/ppo/multiprocessing_env.py    SubprocVecEnv itself: 16 forked workers, Pipes, np.stack order (:97-153)
    /root/reference/ars/train.py                  ARS.__init__ (create_env / create_envs), ARS.train_one_epoch():
                                                  test_envs twice (:74-116: (16, 8, 1) actions, `total_reward += reward`
                                                  on a list), update_weights; then ARS.train's evaluation call
                                                  test_env(self.env, policy, self.weights, normalizer, eval_policy=True)
                                                  on the trainer's single env (:43-71, :228: (8, 1) float64 actions)
    /root/reference/ppo/train.py                  train(args) for 40 frames: two 20-step rollouts (`envs.step(a)`,
                                                  `sum(reward)`, `1 - done`), compute_gae, ppo_update, and the policy
                                                  test at frame 40 (utils.test_env on the trainer's single env)

Both trainers hand the module they import as `pybullet` to snake.Snake as its client (ars/train.py:21-23, ppo/utils.py:
63-68).  Here that module is a stand-in whose calls are answered by the CPU oracle (OracleClient of
make_env_logic_vectors.py); `gym`, `tensorboardX`, `matplotlib` are stand-ins too (a Box, a SummaryWriter that records its
scalars, an empty module).  The forked workers each own a copy of the client, as each owns a copy of PyBullet's DIRECT
world in the reference.  The reference's files are imported as they are; three things of the environment are bridged:
  * numpy >= 1.24 refuses the ragged list `np.asarray([action[i]*SF - observation[i] ...])` that snake.py:228-231 builds
    from ARS's (8, 1) actions (a list mixing (1,) arrays and floats); the numpy of the reference's day made an object
    array of it.  snake.py's `np` is wrapped so that `asarray` does what it did then.  Nothing else of numpy is touched.
  * numpy >= 2 (NEP 50) keeps `float32 scalar * Python float` in float32; the numpy of the reference's day promoted it
    to float64 (snake.py:224, 229: `i*self.SCALING_FACTOR` on the elements of PPO's float32 actions).  Under the old
    rules every operation downstream of a float32 action element involves a Python float or the float64 observation,
    so converting the action to float64 where it enters SnakeGymEnv.step is the same arithmetic; that is done here,
    and checkBound's in-place clip is copied back into the caller's float32 array.
  * snake.py:296's 10-ms sleep per substep is skipped.
Instrumentation (adds records, changes nothing): SnakeGymEnv.step is wrapped to note, per process, the simulator state
an env-step started from, the substep count and the servo errors; VecEnv.step / reset are wrapped to note what the
trainer passed in and got back.

Stored (arrays only, no reference text): per vector step the actions as the trainer passed them (shape and dtype
kept), the stacked obs / rewards / dones as SubprocVecEnv returned them, the 16 workers' pre-step states / substep
counts / servo errors; the trainers' own results (ARS: the two `total_reward` lists, the updated weights; PPO: the scalars
it logs, the eval env's steps).  tests/test_vecenv_golden.py: 16 oracle envs reproduce all of it exactly (CPU), the
product's SubprocVecEnv on the HIP kernels reproduces it to the float32 tolerance (GPU).

Run here (the reference does not exist on the GPU box):  python tests/golden/make_vecenv_vectors.py
"""
import importlib.util
import os
import shutil
import sys
import tempfile
import types

import warnings

import numpy as np

warnings.filterwarnings("ignore", category=DeprecationWarning)      # float((1,) array), what PyBullet's C parser does
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_env_logic_vectors as base  # noqa: E402  (stand-ins for pybullet / gym, the reference's snake / SnakeGymEnv / multiprocessing_env, OracleClient)

orc, ref_snake, ref_env, ref_mp = base.orc, base.ref_snake, base.ref_env, base.ref_mp
NENV, N, O = 16, 16, 56
REC = 45 + 18 + 2 * N * 29 + 16 + 1 + 41 + 1 + O + 1  # state | aux | manifold | action as given (padded) | k | servo errors | done | obs, reward as SnakeGymEnv.step returned them
TMP = tempfile.mkdtemp(prefix="vecenv_vectors_")


# ---- numpy as the reference's day had it: ragged -> object array (snake.py:228-231 under (8, 1) actions) -------------
class _LegacyNumpy(types.ModuleType):
    def __init__(self):
        super().__init__("numpy")

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def asarray(x, *a, **k):
        try:
            return np.asarray(x, *a, **k)
        except ValueError:
            return np.asarray(x, dtype=object)


ref_snake.np = _LegacyNumpy()


# ---- the `pybullet` module the trainers import: one OracleClient behind module attributes ---------------------------
class PbClient(base.OracleClient):
    def setJointMotorControlArray(self, body, joints, mode, targetPositions, forces=None):
        # PyBullet's C parser takes each item through float(): ARS's targets are (1,) arrays and ints mixed
        return base.OracleClient.setJointMotorControlArray(self, body, joints, mode, [float(t) for t in targetPositions], forces=forces)


client = PbClient()
pb = sys.modules["pybullet"]
for name in dir(client):
    if not name.startswith("_"):
        setattr(pb, name, getattr(client, name))
pb.DIRECT, pb.GUI = 2, 1
pb.connect = lambda mode: 0
pb.client = client

sys.modules["gym"].wrappers = types.ModuleType("gym.wrappers")
sys.modules["gym.wrappers"] = sys.modules["gym"].wrappers

SCALARS = []


class _SummaryWriter(object):
    def __init__(self, log_dir=None):
        if log_dir:
            os.makedirs(log_dir, exist_ok=True)

    def add_scalar(self, tag, value, step):
        SCALARS.append((tag, float(value), int(step)))


_tbx = types.ModuleType("tensorboardX")
_tbx.SummaryWriter = _SummaryWriter
sys.modules["tensorboardX"] = _tbx
_mpl = types.ModuleType("matplotlib")
_mpl.pyplot = types.ModuleType("matplotlib.pyplot")
sys.modules.setdefault("matplotlib", _mpl)
sys.modules.setdefault("matplotlib.pyplot", _mpl.pyplot)


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ---- instrumentation ------------------------------------------------------------------------------------------------
_real_env_step = ref_env.SnakeGymEnv.step


def _env_step_recorded(self, action):
    c = self.robot._pybulletClient.client
    e = c.e
    tau, fz, _ = e.get_aux()
    prev_x = float(self._observation[3 * e.n])
    a_in = np.zeros(16)
    a_in[:len(action)] = np.asarray(action, dtype=np.float64).reshape(-1)
    pre = np.concatenate([e.get_state(), tau, [fz, prev_x], e.get_manifold().reshape(-1), a_in])
    c.servo_err = []
    given = action
    if isinstance(action, np.ndarray) and action.dtype == np.float32:
        action = action.astype(np.float64)                  # the promotion rules of the reference's numpy (see above)
    out = _real_env_step(self, action)
    if given is not action:
        given[...] = action                                 # checkBound's in-place clip, back in the caller's array
    k = self.robot.counter
    err = np.zeros(41)
    err[:k] = c.servo_err
    with open(os.path.join(TMP, "pid%d.f64" % os.getpid()), "ab") as f:
        f.write(np.concatenate([pre, [k], err, [1.0 if out[2] else 0.0], out[0], [out[1]]]).tobytes())
    return out


ref_env.SnakeGymEnv.step = _env_step_recorded

CALLS = []            # ("reset", obs) / ("step", actions as passed, obs, rews, dones, infos summary)
_real_vec_step, _real_vec_reset = ref_mp.VecEnv.step, ref_mp.SubprocVecEnv.reset


def _vec_step_recorded(self, actions):
    a = np.array(actions, copy=True)
    out = _real_vec_step(self, actions)
    obs, rews, dones, infos = out
    assert isinstance(infos, tuple) and all(type(i) is dict and i == {} for i in infos)
    CALLS.append(("step", a, obs.copy(), rews.copy(), dones.copy(), np.array(actions, copy=True)))
    return out


def _vec_reset_recorded(self):
    obs = _real_vec_reset(self)
    CALLS.append(("reset", obs.copy()))
    return obs


ref_mp.VecEnv.step = _vec_step_recorded
ref_mp.SubprocVecEnv.reset = _vec_reset_recorded


def worker_records(pids):
    out = []
    for pid in pids:
        raw = np.fromfile(os.path.join(TMP, "pid%d.f64" % pid), dtype=np.float64)
        out.append(raw.reshape(-1, REC))
    return out


def pack_records(R, prefix, d):
    """R [..., REC]: the per-env-step records of _env_step_recorded, split into named arrays."""
    o = 0
    for name, w in (("state", 45), ("aux", 18), ("manifold", 2 * N * 29), ("action_in", 16), ("substeps", 1), ("servo_err", 41), ("done_flag", 1),
                    ("env_obs", O), ("env_reward", 1)):
        d[prefix + name] = R[..., o:o + w] if w > 1 else R[..., o]
        o += w
    d[prefix + "substeps"] = d[prefix + "substeps"].astype(np.int32)
    d[prefix + "servo_err"] = d[prefix + "servo_err"].astype(np.float32)      # (only read for "how near the 0.05 boundary")
    # the contact caches are not stored (0.9 of the bytes): the oracle envs of tests/test_vecenv_golden.py, which
    # reproduce every stored state exactly from the reset on, hold them when the GPU test needs them
    d[prefix + "manifold_points"] = d.pop(prefix + "manifold").reshape(R.shape[:-1] + (2 * N, 29))[..., 0].sum(axis=-1).astype(np.int32)


def pack_calls(calls, recs, prefix, d):
    """calls: this trainer's slice of CALLS; recs: per worker [T, REC]."""
    steps = [c for c in calls if c[0] == "step"]
    T = len(steps)
    assert all(len(r) == T for r in recs), ([len(r) for r in recs], T)
    R = np.stack(recs, axis=1)                                  # [T, NENV, REC]
    d[prefix + "actions"] = np.stack([c[1] for c in steps])     # as the trainer passed them: shape and dtype kept
    # the trainer's array after the call: untouched (the actions are pickled to the workers, checkBound clips THEIR copies)
    assert all(np.array_equal(c[1], c[5]) for c in steps)
    d[prefix + "obs"] = np.stack([c[2] for c in steps])
    d[prefix + "rews"] = np.stack([c[3] for c in steps])
    d[prefix + "dones"] = np.stack([c[4] for c in steps])
    d[prefix + "resets"] = np.stack([c[1] for c in calls if c[0] == "reset"])
    d[prefix + "reset_before_step"] = np.array([sum(1 for c in calls[:i] if c[0] == "step") for i, c in enumerate(calls) if c[0] == "reset"], dtype=np.int32)
    pack_records(R, prefix, d)
    # what SubprocVecEnv returned, as types: the stacked arrays' dtypes (names), for the drop-in's dtype table
    d[prefix + "dtypes"] = np.array([str(steps[0][2].dtype), str(steps[0][3].dtype), str(steps[0][4].dtype)])
    assert np.array_equal(d[prefix + "done_flag"] != 0, d[prefix + "dones"])
    # SnakeGymEnv.step's own return inside the worker: the reward is what travels; the observation is replaced by the
    # reset one on done (multiprocessing_env.py:13-16) -- checked here, not stored twice
    assert np.array_equal(d[prefix + "env_reward"], d[prefix + "rews"])
    nd = ~d[prefix + "dones"]
    assert np.array_equal(d[prefix + "env_obs"][nd], d[prefix + "obs"][nd])
    d[prefix + "terminal_obs"] = d[prefix + "env_obs"][d[prefix + "dones"]]       # rows in (step, env) order of the dones
    a = np.stack([c[1] for c in steps]).reshape(T, NENV, -1)
    assert np.array_equal(d[prefix + "action_in"][..., :a.shape[-1]], a)          # each worker got ITS row, in env order
    for k in ("env_obs", "env_reward", "done_flag", "action_in", "aux"):
        del d[prefix + k]
    return T


def run_ars(d):
    os.chdir(TMP)
    ars = load("ref_ars_train", "/root/reference/ars/train.py")
    args = types.SimpleNamespace(v=0.03, N=16, b=16, lr=0.02, normalizer=True, log=os.path.join(TMP, "ars_log"), mode="train",
                                 alpha=1.0, beta=0.01, gamma=0.1, selfCollisionEnabled=True, motorVelocityLimit=np.inf,
                                 motorTorqueLimit=np.inf, kp=10.0, kd=0.1, gaitSelection=1, scaling_factor=6, cam_dist=5.0,
                                 cam_yaw=50, cam_pitch=-35, cam_roll=0, upAxisIndex=2, render_height=720, render_width=960,
                                 fov=60, nearVal=0.1, farVal=100)          # ars/train.py:235-262's defaults
    np.random.seed(11)
    n0 = len(CALLS)
    trainer = ars.ARS(args)
    pids = [p.pid for p in trainer.envs.ps]
    # the loop of ARS.train_one_epoch (ars/train.py:206-217), its pieces called one by one so that delta and both
    # reward lists can be kept
    delta = [ars.sample_delta(trainer.size) for _ in range(trainer.N)]
    weights_p = np.array([(trainer.weights + trainer.v * x) for x in delta])
    weights_n = np.array([(trainer.weights - trainer.v * x) for x in delta])
    reward_p = ars.test_envs(trainer.envs, ars.policy, weights_p, normalizer=trainer.normalizer)
    reward_n = ars.test_envs(trainer.envs, ars.policy, weights_n, normalizer=trainer.normalizer)
    new_w = ars.update_weights([reward_p, reward_n, delta], trainer.lr, trainer.b, trainer.weights.copy())
    trainer.envs.close()
    assert trainer.envs.closed and all(not p.is_alive() for p in trainer.envs.ps)
    T = pack_calls(CALLS[n0:], worker_records(pids), "ars_", d)
    assert T == 100
    d["ars_state_every5"] = d.pop("ars_state")[::5]       # spot checks of the pre-step states (the PPO run keeps all)
    d["ars_reward_p"] = np.array(reward_p, dtype=np.float64)          # a LIST of 16 + 50 * 16 floats (ars/train.py:82, 107)
    d["ars_reward_n"] = np.array(reward_n, dtype=np.float64)
    d["ars_new_weights"] = new_w
    d["ars_norm_n"], d["ars_norm_mean"], d["ars_norm_var"] = trainer.normalizer.n, trainer.normalizer.mean, trainer.normalizer.var
    d["ars_obs_space_shape"] = np.array(trainer.envs.observation_space.shape)
    d["ars_act_space_shape"] = np.array(trainer.envs.action_space.shape)
    print("ARS: %d vector steps, %d dones, substeps %d..%d, reward lists of %d entries"
          % (T, int(d["ars_dones"].sum()), d["ars_substeps"].min(), d["ars_substeps"].max(), len(reward_p)))
    # ---- the per-epoch evaluation on the trainer's SINGLE env (ars/train.py:228 -> :43-71), round 6:
    # ARS.train's next two lines after train_one_epoch, called as they stand.  env.reset() (a soft reset), the observation
    # plus np.random.random_sample noise, then <= 200 x SnakeGymEnv.step with the (8, 1) float64 column
    # policy(normalizer.normalize(state), weights) -- checkBound indexing a 2-D array (SnakeGymEnv.py:82-88) --, the
    # normaliser frozen (eval_policy=True).  Placed behind everything above so that no earlier array changes.
    trainer.weights = new_w

    def evaluate(prefix, weights, row0):
        noise, passed = [], []
        real_rs = np.random.random_sample

        def rs_recorded(*a, **k):
            out = real_rs(*a, **k)
            noise.append(np.array(out, copy=True))
            return out
        real_step = trainer.env.step

        def step_noting_the_callers_array(action):
            before = np.array(action, copy=True)
            out = real_step(action)
            passed.append((before, np.array(action, copy=True), action.shape, str(action.dtype)))
            return out
        np.random.random_sample = rs_recorded
        trainer.env.step = step_noting_the_callers_array
        try:
            test_reward, num_plays = ars.test_env(trainer.env, ars.policy, weights, normalizer=trainer.normalizer, eval_policy=True)
        finally:
            np.random.random_sample = real_rs
            del trainer.env.step
        ev = worker_records([os.getpid()])[0][row0:]
        assert len(ev) == num_plays == len(passed) and len(noise) == 1
        pack_records(ev, prefix, d)
        d[prefix + "done"] = d.pop(prefix + "done_flag") != 0
        assert all(sh == (8, 1) and dt == "float64" for _, _, sh, dt in passed)
        d[prefix + "action_passed"] = np.stack([b for b, _, _, _ in passed])           # [T, 8, 1] as policy() returned them
        d[prefix + "action_after"] = np.stack([a for _, a, _, _ in passed])            # the caller's array after step(): clipped in place
        d[prefix + "noise"] = noise[0]                                                 # what was added to the reset observation
        d[prefix + "weights"] = np.array(weights, copy=True)
        d[prefix + "test_reward"] = np.float64(test_reward)
        d[prefix + "num_plays"] = np.int32(num_plays)
        # the normaliser did not move (eval_policy=True)
        assert np.array_equal(trainer.normalizer.n, d["ars_norm_n"]) and np.array_equal(trainer.normalizer.mean, d["ars_norm_mean"])
        print("ARS eval %s(test_env on the trainer's single env): %d env-steps, done %s, test_reward %.6f, substeps %s..., "
              "|action| up to %.2f, %d components clipped"
              % (prefix, num_plays, bool(d[prefix + "done"][-1]), test_reward, d[prefix + "substeps"].tolist()[:8],
                 np.abs(d[prefix + "action_passed"]).max(), int((np.abs(d[prefix + "action_passed"]) > 1).sum())))
        return len(ev)
    # (1) the call as ARS.train makes it after the first epoch: weights one update away from zero, a near-idle snake --
    #     0-substep steps, the 200-step cap, no episode end
    n1 = evaluate("ars_eval_", trainer.weights, 0)
    # (2) the same function with the weights a later epoch would bring (x 12: commands beyond +-1, so checkBound clips the
    #     caller's (8, 1) array in place, the snake moves)
    n2 = evaluate("ars_eval2_", 12.0 * trainer.weights, n1)
    d["ars_eval_rows"] = np.int32(n1 + n2)


def run_ppo(d):
    import torch
    os.chdir(TMP)
    sys.path.insert(0, "/root/reference/ppo")
    ppo = load("ref_ppo_train", "/root/reference/ppo/train.py")
    args = sys.modules["params"].params(["--max_frames", "40", "--log_dir", os.path.join(TMP, "ppo_log")])
    torch.manual_seed(5)
    np.random.seed(5)
    n0, s0 = len(CALLS), len(SCALARS)
    made = []
    real_init = ref_mp.SubprocVecEnv.__init__

    def init_and_note(self, *a, **k):
        real_init(self, *a, **k)
        made.append(self)
    ref_mp.SubprocVecEnv.__init__ = init_and_note
    ppo.SubprocVecEnv.__init__ = init_and_note
    ppo.train(args)
    ref_mp.SubprocVecEnv.__init__ = real_init
    envs = made[0]
    pids = [p.pid for p in envs.ps]
    envs.close()
    T = pack_calls(CALLS[n0:], worker_records(pids), "ppo_", d)
    assert T == 40
    sc = SCALARS[s0:]
    d["ppo_scalar_tags"] = np.array([t for t, _, _ in sc])
    d["ppo_scalar_values"] = np.array([v for _, v, _ in sc])
    d["ppo_scalar_frames"] = np.array([f for _, _, f in sc], dtype=np.int32)
    # the trainer's own single env (ppo/train.py:93-94), driven by utils.test_env at frame 40: this process's records
    ev = worker_records([os.getpid()])[0][int(d["ars_eval_rows"]):]          # (this process's records start with ARS's evaluation)
    pack_records(ev, "ppo_eval_", d)
    d["ppo_eval_done"] = d.pop("ppo_eval_done_flag") != 0
    print("PPO: %d vector steps, %d dones, substeps %d..%d; scalars %s; eval env-steps %d"
          % (T, int(d["ppo_dones"].sum()), d["ppo_substeps"].min(), d["ppo_substeps"].max(),
             sorted(set(d["ppo_scalar_tags"].tolist())), len(ev)))


def main():
    d = {}
    try:
        run_ars(d)
        run_ppo(d)
    finally:
        os.chdir(HERE)
        shutil.rmtree(TMP, ignore_errors=True)
    out = os.path.join(HERE, "vecenv_vectors.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
