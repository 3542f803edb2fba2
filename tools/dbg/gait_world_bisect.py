"""Which parameter of the gait-test world makes the first substep differ between GPU and oracle?"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402

pkg = importlib.import_module("bullet-envs_amd")
d = np.load(os.path.join(ROOT, "tests", "golden", "gait_test_vectors.npz"))
cases = [dict(), dict(self_collision=0), dict(max_motor_impulse=0.04), dict(max_motor_impulse=4 / 240.), dict(dt=0.01),
         dict(self_collision=0, max_motor_impulse=0.04), dict(self_collision=0, dt=0.01), dict(dt=0.01, max_motor_impulse=0.04),
         dict(max_motor_impulse=0.04, residual_threshold=0.0), dict(max_motor_impulse=0.04, warm_start=0),
         dict(max_motor_impulse=0.04, contact_model=0, self_collision=0)]
for w in cases:
    for tscale in (1.0, 0.1):
        e = orc.OracleEnv(**w)
        e.hard_reset()
        st = pkg._lib.Stepper(1, 0, n_modules=16, **w)
        tg = d["targets"][0] * tscale
        e.substep(tg)
        info = st.substep(np.asarray(tg, np.float32)[None])
        s, _ = st.get_state()
        o = e.get_state()
        dqd = s[0, 29:45] - o[29:45]
        print("%-70s tscale %.1f: |dq| %.2e |dqd| %.2e (joint %d: gpu %.4f oracle %.4f) |d v| %.2e iters gpu %d oracle %d nc %d/%d" % (
            w, tscale, np.abs(s[0, 13:29] - o[13:29]).max(), np.abs(dqd).max(), np.abs(dqd).argmax(), s[0, 29 + np.abs(dqd).argmax()],
            o[29 + np.abs(dqd).argmax()], np.abs(s[0, 7:13] - o[7:13]).max(), info[0, 0], e.last_iterations,
            info[0, 1], e.last_num_contacts))
        st.close()
