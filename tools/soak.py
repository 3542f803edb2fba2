import importlib, sys, time
import numpy as np
sys.path.insert(0,'.')
from bench import gait_actions
pkg = importlib.import_module("bullet-envs_amd")
B=4096
st = pkg.Stepper(B); st.reset()
tot_done=0; tot_sub=0; t0=time.time()
for j in range(600):
    o,r,d,s = st.step(gait_actions(np.arange(B), j).astype(np.float32))
    assert np.isfinite(o).all() and np.isfinite(r).all(), j
    tot_done += int(d.sum()); tot_sub += int(s.sum())
print("600 env-steps x 4096 envs: finite everywhere, %d episode ends, mean substeps %.2f, %.1f s" % (tot_done, tot_sub/(600*B), time.time()-t0))
# random actions incl. out-of-range
rng=np.random.default_rng(0)
for j in range(100):
    o,r,d,s = st.step(rng.uniform(-3,3,(B,8)).astype(np.float32))
    assert np.isfinite(o).all() and np.isfinite(r).all(), j
print("100 random-action steps finite; max |qd| %.1f, max |obs55| %.1f" % (np.abs(o[:,16:32]).max(), np.abs(o[:,55]).max()))
