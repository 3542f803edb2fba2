import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("bullet-envs_amd")
k = 5
for B in (256, 1280):
    res = {}
    for nit in (1, 10, 50):
        st = pkg.Stepper(B, n_modules=32, n_iterations=nit, residual_threshold=0.0)
        st.reset()
        T = np.zeros((B, 32), np.float32); T[:, 1::2] = 0.3
        st.substep(T, 1)
        t0 = time.perf_counter(); info = st.substep(T, k); dt = time.perf_counter() - t0
        res[nit] = dt / k * 1e6
        st.close()
    per_it = (res[50] - res[10]) / 40
    print("N=32 B=%d: substep %.0f us at 50 it, %.0f us at 10 it, %.0f us at 1 it -> %.1f us per iteration, fixed %.0f us (contacts %d)" % (B, res[50], res[10], res[1], per_it, res[1] - per_it, info[0, 1]))
