"""GPU vs oracle, step by step, in the gait-test script's world (tests/test_gait_test_golden.py): where do they part?"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402
import test_gait_test_golden as T  # noqa: E402

pkg = importlib.import_module("bullet-envs_amd")
d = np.load(os.path.join(ROOT, "tests", "golden", "gait_test_vectors.npz"))
world = json.loads(str(d["world_json"]))
variants = [world] + [dict(world, **kv) for kv in (dict(obstacle=0), dict(dt=1 / 240., max_motor_impulse=4 / 240.), dict(gravity_z=-10.0))]
for w in variants:
    w = {k: v for k, v in w.items() if not (k == "obstacle_pos" and w.get("obstacle") == 0)}
    e = orc.OracleEnv(**w)
    e.hard_reset()
    st = pkg._lib.Stepper(1, 0, n_modules=16, **w)
    st.hard_reset() if hasattr(st, "hard_reset") else None
    s0, _ = st.get_state()
    print("world", w, "\n  initial |d state|", np.abs(s0[0] - e.get_state()).max())
    for k in range(16):
        tg = d["targets"][k]
        e.substep(tg)
        st.substep(np.asarray(tg, np.float32)[None])
        s, _ = st.get_state()
        o = e.get_state()
        print("  step %2d |dq| %.2e |dqd| %.2e |d head| %.2e |d quat| %.2e |d v| %.2e  fz gpu %.3f oracle %.3f" % (
            k, np.abs(s[0, 13:29] - o[13:29]).max(), np.abs(s[0, 29:45] - o[29:45]).max(), np.abs(s[0, :3] - o[:3]).max(),
            np.abs(s[0, 3:7] - o[3:7]).max(), np.abs(s[0, 7:13] - o[7:13]).max(), st.joint3_reaction_fz()[0] if hasattr(st, "joint3_reaction_fz") else -1,
            e.joint3_reaction_fz()))
