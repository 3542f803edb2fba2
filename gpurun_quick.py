# scratch timing script (not part of the product)
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("bullet-envs_amd")
B = 4096
st = pkg.Stepper(B)
st.reset()
k = np.arange(8)
phi = np.random.default_rng(0).uniform(0, 2*np.pi, B); phi[0] = 0
def act(j):
    return (-np.sin((2*k[None,:]+1)*4.0 + 2.0*(0.1*j) + phi[:,None])).astype(np.float32)
for j in range(5):
    st.step(act(j))
tot = 0
t0 = time.time()
for j in range(5, 25):
    o, r, d, s = st.step(act(j))
    tot += s.sum()
dt = time.time() - t0
print("20 steps of %d envs: %.3f s; env-steps/s %.0f; substeps/s %.0f; mean substeps %.2f; done frac %.3f" % (B, dt, 20*B/dt, tot/dt, tot/(20*B), d.mean()))
