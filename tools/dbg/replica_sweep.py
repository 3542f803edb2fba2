"""Replicas of a few environments scattered among ordinary ones, several configurations and slice lengths: every replica
of an environment must end the env-step on the same bits (whatever wave ran it, whatever ran beside it)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_env import gait
pkg = importlib.import_module("bullet-envs_amd")
CASES = [(16, 5000, {}), (16, 5000, dict(warm_start=1)), (32, 2600, {}), (16, 3000, dict(obstacle=1, obstacle_pos=[0.12, 0.0, 0.1])),
         (16, 3000, dict(obstacle=2, obstacle_pos=[0.12, 0.0, 0.1])), (16, 5000, dict(hull_sides=0, contact_model=0, relative_breaking_threshold=0)),
         (32, 2600, dict(obstacle=1, obstacle_pos=[0.12, 0.0, 0.1]))]
bad_total = 0
for n, B, over in CASES:
    A = n // 2
    for quantum in (1, 0):
        os.environ["SNK_QUANTUM"] = str(quantum)
        st = pkg.Stepper(B, n_modules=n, **over)
        st.reset()
        fr = (0.5 + np.arange(B) % 11 / 10.0).astype(np.float32)
        st.set_ground_friction(fr)
        for j in range(2):
            st.step((gait(range(B), j, A) * 1.2).astype(np.float32))
        S, X = st.get_state(); Mf = st.get_manifold(); BX = st.get_box() if over.get("obstacle") == 2 else None
        a = (gait(range(B), 2, A) * 1.2).astype(np.float32)
        o0, r0, d0, s0 = st.step(a.copy())
        srcs = [int(np.argmax(s0)), int(np.argmin(s0 + 100 * (s0 == 0))), 1234 % B]
        idx = np.arange(7, B, 13)
        for e in srcs:
            S2, X2, a2, f2 = S.copy(), X.copy(), a.copy(), fr.copy()
            S2[idx], X2[idx], a2[idx], f2[idx] = S[e], X[e], a[e], fr[e]
            st.set_ground_friction(f2); st.set_state(S2, X2)
            if Mf is not None:
                M2 = Mf.copy(); M2[idx] = Mf[e]
                st.set_manifold(M2)
            if BX is not None:
                b0, b1 = BX[0].copy(), BX[1].copy()
                b0[idx], b1[idx] = BX[0][e], BX[1][e]
                st.set_box(b0, b1)
            o, r, d, s = st.step(a2.copy())
            camps = len(np.unique(o[idx], axis=0))
            same_as_source = np.array_equal(o[idx[0]], o0[e])
            bad_total += camps != 1 or not same_as_source
            print("%2d links %-62s quantum %d: source env %4d (%2d substeps): %d replicas, %d distinct outcomes; equal to the source's own outcome: %s"
                  % (n, over, quantum, e, s0[e], len(idx), camps, same_as_source), flush=True)
        st.close()
print("replica sweep:", "ok" if bad_total == 0 else "%d FAILURES" % bad_total)
