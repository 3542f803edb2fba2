"""Distribution of GPU-float32 vs oracle-float64 single-substep errors on the ground (512 random
states), next to oracle-float32 vs oracle-float64 on the same inputs.  The maxima are heavy-tailed
(stick-slip states amplify round-off by 1e5); medians and 90th percentiles are what to compare
between builds."""
import importlib, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import oracle as orc
from conftest import random_state
pkg = importlib.import_module("bullet-envs_amd")
n, B = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(4321)
S = np.zeros((B, 13 + 2 * n))
for i in range(B):
    S[i] = random_state(rng, n, z=0.026, qamp=0.3, vamp=0.3, flat=True)
    S[i, 9] *= 0.1; S[i, 7:9] *= 0.1
S32 = S.astype(np.float32)
T = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32)
st = pkg.Stepper(B, residual_threshold=0.0)
st.set_state(S32); st.substep(T, 1)
G, _ = st.get_state()
o = orc.OracleEnv(residual_threshold=0.0); o32 = orc.OracleEnv(residual_threshold=0.0, f32=True)
eg, e32 = [], []
for i in range(B):
    o.set_state(S32[i].astype(np.float64)); o.substep(T[i].astype(np.float64)); r = o.get_state()
    o32.set_state(S32[i].astype(np.float64)); o32.substep(T[i].astype(np.float64)); r32 = o32.get_state()
    f = lambda x: (np.abs(x[13 + n:] - r[13 + n:]) / (1 + np.abs(r[13 + n:]))).max()
    eg.append(f(G[i])); e32.append(f(r32))
for name, e in (("GPU float32   ", np.array(eg)), ("oracle float32", np.array(e32))):
    print(name, "median %.3e  p90 %.3e  p99 %.3e  max %.3e" % (np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()))
