"""env-steps/s through the host-buffer C call (snk_step_host: H2D actions, step, D2H obs/reward/done/substeps,
synchronous) -- the PCIe-inclusive figure DESIGN.md 5 quotes next to bench.py's HBM-resident one."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module("bullet-envs_amd")
E, W, K = 4096, 10, 100
st = pkg.Stepper(E)
st.reset()
ids = np.arange(E)
acts = [bench.gait_actions(ids, j).astype(np.float32) for j in range(W + K)]
for j in range(W):
    st.step(acts[j])
t0 = time.perf_counter()
for j in range(W, W + K):
    st.step(acts[j])
dt = time.perf_counter() - t0
print("host-buffer path: %.1f env-steps/s (%.2f ms per 4096-env step)" % (E * K / dt, dt / K * 1e3))
