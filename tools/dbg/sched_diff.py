"""Which environments end a step on different bits under another slice length (tests/test_gpu_env.py::
test_schedule_does_not_change_results)?   [SD_OVER='{"self_collision": 0}'] python tools/dbg/sched_diff.py [16|32] [B]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_env import gait
pkg = importlib.import_module("bullet-envs_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
A = n // 2


def run(quantum):
    os.environ["SNK_QUANTUM"] = str(quantum)
    st = pkg.Stepper(B, n_modules=n, **__import__("json").loads(os.environ.get("SD_OVER", "{}")))
    st.reset()
    st.set_ground_friction((0.5 + np.arange(B) % 11 / 10.0).astype(np.float32))
    outs = []
    for j in range(4):
        a = (gait(range(B), j, A) * 1.2).astype(np.float32)
        o, r, d, s = st.step(a)
        outs.append((o.copy(), r.copy(), d.copy(), s.copy(), st.contact_overflow()))
    st.close()
    return outs


ref = run(0)
for q in (1, 3, 64, 0):
    got = run(q)
    for j, ((o, r, d, s, ov), (O, R, D, S, OV)) in enumerate(zip(got, ref)):
        bad = np.nonzero((o != O).any(axis=1))[0]
        print("quantum %2d step %d: %d envs differ %s; max |d obs| %.2e; overflow counters %s vs %s" % (
            q, j, len(bad), bad[:8], np.abs(o - O).max(), ov, OV))
if os.environ.get("SNK_POISON"):
    for j, (o, r, d, s, ov) in enumerate(ref):
        bad = np.nonzero(~np.isfinite(o).all(axis=1))[0]
        print("poisoned run, step %d: %d envs with non-finite observations %s" % (j, len(bad), bad[:10]))
