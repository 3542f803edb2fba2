"""Generates tests/golden/test_script_env_vectors.npz by RUNNING the reference's own /root/reference/test_script_env.py --
the one script in the reference that exercises the env end to end: Snake + SnakeGymEnv on the module-level PyBullet
client, reset, then 60 x env.step([0.5] * 8) with env.render() and the rewards summed up.

The script is executed as it is (runpy); its `import pybullet as p` resolves to the oracle-backed client of
make_env_logic_vectors.py (plus `connect` / `GUI`, which the script calls on the module), `time.sleep` does nothing, its
prints go nowhere.  SnakeGymEnv.step is wrapped to record, per env-step, the state the step started from and what the
reference returned.  What stepSimulation computes is the oracle's restatement of Bullet ([U], parity unpinned).

Run here:  python tests/golden/make_test_script_vectors.py
"""
import io
import os
import runpy
import sys
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_env_logic_vectors as base  # noqa: E402  (stand-in modules, OracleClient; imports the reference's files)


def main():
    client = base.OracleClient()
    client.GUI, client.DIRECT = 1, 2
    client.connect = lambda mode: 0
    sys.modules["pybullet"] = client                     # `import pybullet as p` in the script
    rows = []
    real_step = base.ref_env.SnakeGymEnv.step

    def recording_step(self, action):
        e = client.e
        tau, fz, _ = e.get_aux()
        pre = dict(state=e.get_state(), aux=np.concatenate([tau, [fz, float(self._observation[3 * e.n])]]), manifold=e.get_manifold())
        a_in = list(action)
        client.servo_err = []
        out = real_step(self, action)
        obs, rew, done, info = out
        assert info == {} and self.render().size == 0     # train mode: no frames (SnakeGymEnv.py:52-58)
        rows.append(dict(pre=pre, a_in=np.array(a_in, dtype=np.float64), a_out=np.array(list(action), dtype=np.float64),
                         obs=np.array(obs, dtype=np.float64), rew=float(rew), done=bool(done), k=int(self.robot.counter),
                         err=np.pad(np.array(client.servo_err), (0, 41 - len(client.servo_err)))))
        return out
    base.ref_env.SnakeGymEnv.step = recording_step
    import time
    real_sleep, time.sleep = time.sleep, (lambda s: None)
    out = io.StringIO()
    try:
        with contextlib.redirect_stdout(out):
            runpy.run_path("/root/reference/test_script_env.py", run_name="__main__")
    finally:
        time.sleep = real_sleep
        base.ref_env.SnakeGymEnv.step = real_step
    assert len(rows) == 60
    printed_total = float(out.getvalue().strip().splitlines()[-1].split(":")[1])     # the script's "Total Reward: ..."
    total = 0.0
    for r in rows:
        total += r["rew"]                                 # the script's own accumulation order (R += r)
    assert abs(total - printed_total) < 1e-12 * max(1.0, abs(total)), (total, printed_total)
    path = os.path.join(HERE, "test_script_env_vectors.npz")
    np.savez_compressed(path, state=np.stack([r["pre"]["state"] for r in rows]), aux=np.stack([r["pre"]["aux"] for r in rows]),
                        manifold=np.stack([r["pre"]["manifold"] for r in rows]), action_in=np.stack([r["a_in"] for r in rows]),
                        action_out=np.stack([r["a_out"] for r in rows]), obs=np.stack([r["obs"] for r in rows]),
                        reward=np.array([r["rew"] for r in rows]), done=np.array([r["done"] for r in rows]),
                        substeps=np.array([r["k"] for r in rows], dtype=np.int32), servo_err=np.stack([r["err"] for r in rows]),
                        total_reward=np.array(total))
    print("wrote", path, os.path.getsize(path), "bytes: 60 env-steps, substeps", [r["k"] for r in rows][:8], "...; dones",
          int(sum(r["done"] for r in rows)), "; total reward %.6f" % total)


if __name__ == "__main__":
    main()
