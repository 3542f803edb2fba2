"""Histogram of substeps per env-step in the bench workload (4096 envs, serpenoid gait)."""
import importlib, sys
import numpy as np
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module("bullet-envs_amd")
B = 4096
st = pkg.Stepper(B); st.reset()
ids = np.arange(B)
allsub = []
for j in range(40):
    o, r, d, s = st.step(bench.gait_actions(ids, j).astype(np.float32))
    if j >= 10: allsub.append(s)
s = np.concatenate(allsub)
h = np.bincount(s, minlength=42)
print("mean %.2f  max %d  p50 %d  p90 %d  p99 %d" % (s.mean(), s.max(), np.percentile(s, 50), np.percentile(s, 90), np.percentile(s, 99)))
print("hist:", {k: int(v) for k, v in enumerate(h) if v})
per_step_max = [a.max() for a in allsub]
print("max per env-step:", per_step_max[:10], " done frac %.3f" % d.mean())
