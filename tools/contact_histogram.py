"""Contact points per physics substep under the bench's gait (snk_contact_histogram): the distribution that sizes the
register-resident solve's row slots.
    python tools/contact_histogram.py [links=16] [env-steps=60] [--friction-seed 1] [--policy-like]   (on the GPU box)
Prints the percentiles and writes gpurun_out/contact_histogram_<links>.json."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module("bullet-envs_amd")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
NL = int(args[0]) if args else 16
K = int(args[1]) if len(args) > 1 else 60
E = 4096
st = pkg.Stepper(E, n_modules=NL)
if "--friction-seed" in sys.argv:
    st.set_ground_friction(bench.env_friction(np.arange(E), 1).astype(np.float32))
st.reset()
ids = np.arange(E)
rng = np.random.default_rng(0)
for j in range(10):
    st.step(bench.gait_actions(ids, j, NL // 2).astype(np.float32))
st.contact_histogram_enable(True)
st.contact_histogram(reset=True)
tot_sub = 0
for j in range(10, 10 + K):
    a = bench.gait_actions(ids, j, NL // 2).astype(np.float32)
    if "--policy-like" in sys.argv:
        a = rng.normal(size=a.shape).astype(np.float32)
    o, r, d, sub = st.step(a)
    tot_sub += int(sub.sum())
h = st.contact_histogram().astype(np.int64)
n = int(h.sum())
c = np.cumsum(h)
pct = {p: int(np.searchsorted(c, p / 100.0 * n)) for p in (1, 10, 25, 50, 75, 90, 95, 99, 99.9)}
mean = float((h * np.arange(len(h))).sum() / max(n, 1))
print("links %d: %d substeps (%d counted by the step outputs), mean %.2f contact points" % (NL, n, tot_sub, mean))
print("percentiles:", pct, "max", int(np.nonzero(h)[0].max()))
for cap in (36, 40, 44, 48, 52, 56, 64):
    print("  substeps with more than %d points: %.3f %%" % (cap, 100.0 * h[cap + 1:].sum() / max(n, 1)))
print("overflow counters", st.contact_overflow())
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"links": NL, "env_steps": K * E, "substeps": n, "mean": mean, "percentiles": pct, "histogram": h.tolist()},
          open("gpurun_out/contact_histogram_%d.json" % NL, "w"))
os._exit(0)
