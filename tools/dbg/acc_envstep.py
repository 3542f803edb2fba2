"""Whole env-steps from random ground states and random actions: GPU float32 against the float64 oracle, next to the float32
oracle against the same -- substep-count / done mismatches (threshold decisions) and the distribution of observation and
reward errors over the steps where both agree.   python tools/dbg/acc_envstep.py [states] [links]"""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import oracle as orc
from conftest import random_state
pkg = importlib.import_module("bullet-envs_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
over = json.loads(os.environ.get("ACC_OVER", "{}"))
rng = np.random.default_rng(99)
S = np.zeros((B, 13 + 2 * n), np.float32)
for i in range(B):
    s = random_state(rng, n, z=0.026, qamp=0.3, vamp=0.3, flat=True)
    s[9] *= 0.1; s[7:9] *= 0.1
    S[i] = s
A = rng.uniform(-1.3, 1.3, (B, n // 2)).astype(np.float32)          # some beyond the clip
st = pkg.Stepper(B, n_modules=n, **over)
st.set_state(S)
obs, rew, done, sub = st.step(A.copy(), vec_mode=False)
fz3 = st.joint3_reaction_fz()
st.close()
o = orc.OracleEnv(n_modules=n, max_contacts=0, **over); o32 = orc.OracleEnv(n_modules=n, max_contacts=0, f32=True, **over)
mg = m32 = 0
eg, e32, rg, r32_ = [], [], [], []
for i in range(B):
    res = []
    for e in (o, o32):
        e.hard_reset()
        e.set_state(S[i].astype(np.float64))
        res.append(e.env_step(A[i].astype(np.float64), vec_mode=False))
    (oo, rr, dd, kk, _), (ob, rb, db, kb, _) = res
    f = lambda x: np.abs(x[:3 * n + 7] - oo[:3 * n + 7]).max()
    if kb != kk or db != dd: m32 += 1
    else: e32.append(f(ob)); r32_.append(abs(rb - rr))
    if sub[i] != kk or bool(done[i]) != dd: mg += 1
    else: eg.append(f(obs[i].astype(np.float64))); rg.append(abs(float(rew[i]) - rr))
for name, m, e, r in (("GPU float32   ", mg, np.array(eg), np.array(rg)), ("oracle float32", m32, np.array(e32), np.array(r32_))):
    print(name, "mismatches %d of %d | obs median %.3e p90 %.3e p99 %.3e | reward median %.3e p90 %.3e p99 %.3e"
          % (m, B, np.median(e), np.percentile(e, 90), np.percentile(e, 99), np.median(r), np.percentile(r, 90), np.percentile(r, 99)))
