"""Twin environments: env i and env i + B/2 get the same friction, the same actions and start from the same state, but
sit in different slots -- other waves, other neighbours, other times.  They must stay bit-identical for the whole run
(the dynamics amplify any one-bit fault within a few env-steps).   python tools/dbg/twins_soak.py [steps]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_env import gait
pkg = importlib.import_module("bullet-envs_amd")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
CASES = [(16, 4096, {}), (32, 2048, {}), (16, 4096, dict(warm_start=1)), (16, 2048, dict(obstacle=2, obstacle_pos=[0.12, 0.0, 0.1])),
         (16, 2048, dict(obstacle=1, obstacle_pos=[0.12, 0.0, 0.1])), (16, 4096, dict(hull_sides=0, contact_model=0, relative_breaking_threshold=0)),
         (16, 4096, dict(contact_order=3)), (32, 2048, dict(contact_order=1))]       # round 6: another sweep order of the manifolds
bad = 0
rng = np.random.default_rng(3)
for n, B, over in CASES:
    A, H = n // 2, B // 2
    st = pkg.Stepper(B, n_modules=n, **over)
    st.reset()
    fr = (0.5 + np.arange(H) % 11 / 10.0).astype(np.float32)
    st.set_ground_friction(np.concatenate([fr, fr]))
    t0 = time.time()
    first = None
    ov = 0
    for j in range(STEPS):
        a = (gait(range(H), j, A) * 1.2).astype(np.float32)
        if j % 7 == 3:
            a = rng.uniform(-2, 2, (H, A)).astype(np.float32)
        o, r, d, s = st.step(np.concatenate([a, a]))
        if first is None and not (np.array_equal(o[:H], o[H:]) and np.array_equal(r[:H], r[H:]) and np.array_equal(s[:H], s[H:])):
            first = (j, int((o[:H] != o[H:]).any(axis=1).sum()))
    S, X = st.get_state()
    same = np.array_equal(S[:H], S[H:])
    bad += (first is not None) or not same
    print("%2d links %-70s %d steps x %d twin pairs: %s; fallback substeps %s; %.1f s" % (
        n, over, STEPS, H, "identical throughout" if first is None and same else "PARTED at step %d (%d pairs)" % first,
        st.contact_overflow(), time.time() - t0), flush=True)
    st.close()
print("twins:", "ok" if bad == 0 else "FAILURES")
