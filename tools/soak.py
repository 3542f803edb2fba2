"""Soak of the step kernel (scheduler included): long gait run, random and out-of-range actions, random
per-env friction, both chain lengths; every output finite, no scheduler alarm (a stalled queue would turn into
a RuntimeError within seconds).  python tools/soak.py [steps]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
from bench import gait_actions
pkg = importlib.import_module("bullet-envs_amd")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(0)
# (links, envs, parameter overrides): round 2 adds Bullet's contact model (a contact cache that moves between waves with
# the env-step's slices), the obstacle on the 16-link kernels and the wrap-around of the queue's tickets
CASES = ((16, 4096, {}), (32, 4096, {}), (16, 1537, {}), (16, 4096, dict(hull_sides=0, contact_model=0, relative_breaking_threshold=0)),
         (32, 2500, dict(hull_sides=0, contact_model=0, relative_breaking_threshold=0)), (16, 2048, dict(warm_start=1)), (16, 1200, dict(obstacle=1, obstacle_pos=[0.25, 0.0, 0.1])))
for case, (n, B, over) in enumerate(CASES):
    A = n // 2
    st = pkg.Stepper(B, n_modules=n, **over)
    if case == 2:
        st.debug_set_tickets(0xFFFFF000)       # the 32-bit tickets wrap a few steps into the run
    st.reset()
    st.set_ground_friction(rng.uniform(0.5, 1.5, B).astype(np.float32))
    steps = (STEPS if n == 16 else STEPS // 6) if not over else STEPS // 6
    tot_done = tot_sub = 0
    hist = np.zeros(64, np.int64)
    t0 = time.time()
    for j in range(steps):
        o, r, d, s = st.step(gait_actions(np.arange(B), j, A).astype(np.float32))
        assert np.isfinite(o).all() and np.isfinite(r).all(), j
        tot_done += int(d.sum()); tot_sub += int(s.sum()); hist += np.bincount(s, minlength=64)
    print(over if over else "default", end=": ")
    print("%d links, %d envs: %d gait steps finite, %d episode ends, mean substeps %.2f (min %d max %d), %.1f s"
          % (n, B, steps, tot_done, tot_sub / (steps * B), np.nonzero(hist)[0][0], np.nonzero(hist)[0][-1], time.time() - t0), flush=True)
    for j in range(steps // 4):
        a = rng.uniform(-3, 3, (B, A)).astype(np.float32)
        if j % 3 == 0:
            a[rng.random(B) < 0.5] = 0.0            # half the envs already on target: 0-substep env-steps
        o, r, d, s = st.step(a)
        assert np.isfinite(o).all() and np.isfinite(r).all(), j
    print("   %d random-action steps finite; max |qd| %.1f, max |obs fz| %.1f, substeps %d..%d; contact overflow "
          "(substeps, points, link-link/obstacle) %s"
          % (steps // 4, np.abs(o[:, n:2 * n]).max(), np.abs(o[:, -1]).max(), s.min(), s.max(), st.contact_overflow()), flush=True)
    assert st.contact_overflow()[1] == 0, "a ground-contact point was left without rows"     # (structurally zero)
    st.close()
print("soak ok")
