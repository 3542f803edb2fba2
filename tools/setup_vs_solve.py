"""How a physics substep's time splits between the solve (50 Gauss-Seidel iterations) and everything else, in the real
two-waves-per-SIMD regime: substeps/s of the substep kernel on 4096 resting-on-the-ground envs for several iteration
counts (the solve is linear in them; the intercept is the setup).    python tools/setup_vs_solve.py [16|32]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("bullet-envs_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, K = 4096, 40
res = {}
for it in (1, 10, 25, 50):
    st = pkg.Stepper(B, n_modules=n, n_iterations=it, residual_threshold=0.0)
    st.reset()
    T = np.zeros((B, n), np.float32)
    T[:, 1::2] = 0.3 * np.sin(np.arange(B)[:, None] * 0.37 + np.arange(n // 2)[None, :])
    st.substep(T, 5)
    t0 = time.perf_counter()
    st.substep(T, K)
    dt = time.perf_counter() - t0
    res[it] = dt / K * 1e3
    print("iterations %2d: %.3f ms per batched substep (%.2f M substeps/s)" % (it, res[it], B / res[it] / 1e3))
    st.close()
slope = (res[50] - res[10]) / 40.0
print("per iteration %.4f ms; setup (intercept) %.3f ms = %.1f %% of a 50-iteration substep" % (slope, res[50] - 50 * slope, 100 * (res[50] - 50 * slope) / res[50]))
