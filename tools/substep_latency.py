"""Per-substep latency of the HIP stepper against the number of resident waves.
Usage (GPU box): python tools/substep_latency.py [k]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("bullet-envs_amd")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
for B in (256, 1024, 2048, 4096, 8192):
    st = pkg.Stepper(B)
    st.reset()
    T = rng.uniform(-0.5, 0.5, (B, 16)).astype(np.float32)
    st.substep(T, 2)                       # warm-up, snake settles on the ground
    t0 = time.perf_counter()
    info = st.substep(T, k)
    dt = time.perf_counter() - t0
    print("B=%5d  k=%d  %.2f ms total  %.1f us per substep-launch  %.3f M substeps/s  (iters %d, contacts %d)"
          % (B, k, dt * 1e3, dt / k * 1e6, B * k / dt / 1e6, info[0, 0], info[0, 1]))
    st.close()
