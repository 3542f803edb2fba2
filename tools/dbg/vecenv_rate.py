"""env-steps/s of the numpy-facing seams (what a host-side trainer calls): SnakeVecEnv.step with (N, 8) and ARS's (N, 8, 1)
actions, against the bare host-buffer C call (tools/host_api_rate.py)."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
from bench import gait_actions
pkg = importlib.import_module("bullet-envs_amd")
B, K = 4096, 60
for shape in ("N8", "N81"):
    env = pkg.SnakeVecEnv(B)
    env.reset()
    acts = [gait_actions(np.arange(B), j, 8).astype(np.float64) for j in range(K + 10)]
    if shape == "N81":
        acts = [a[:, :, None] for a in acts]
    for j in range(10):
        env.step(acts[j])
    t0 = time.perf_counter()
    for j in range(10, 10 + K):
        obs, rew, done, infos = env.step(acts[j])
    dt = time.perf_counter() - t0
    print("SnakeVecEnv.step, actions %s float64: %.1f k env-steps/s (%.2f ms per step)" % (shape, B * K / dt / 1e3, dt / K * 1e3))
    env.close()
