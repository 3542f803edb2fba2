"""One (trainer, step, env) row of tests/golden/vecenv_vectors.npz: the servo error substep by substep in the float64
oracle, the float32 oracle and (with `gpu` as last argument) on the GPU through snk_substep_host, next to the substep
count of the fused env-step.  usage: vecenv_row_debug.py ars_|ppo_ STEP ENV [gpu]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import oracle as orc  # noqa: E402

orc.build()
v = np.load(os.path.join(ROOT, 'tests/golden/vecenv_vectors.npz'))
t, J, I = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
envs = [orc.OracleEnv() for _ in range(16)]
resets = [int(s) for s in v[t + 'reset_before_step']]
for j in range(J + 1):
    if j in resets:
        for e in envs:
            e.reset()
    if j == J:
        break
    a = v[t + 'actions'][j].reshape(16, -1).astype(np.float64)
    for i, e in enumerate(envs):
        e.env_step(a[i].copy(), vec_mode=True)
e = envs[I]
S = e.get_state()
tau, fz, px = e.get_aux()
X = np.concatenate([tau, [fz, px]])
M = e.get_manifold()
a = v[t + 'actions'][J].reshape(16, -1)[I].astype(np.float64)
print("action", a, "pts", M[:, 0].sum(), "ref k", v[t + 'substeps'][J, I], "ref errs", np.round(v[t + 'servo_err'][J, I][:v[t + 'substeps'][J, I] + 1], 5))
T = np.zeros(16)
T[1::2] = np.clip(a, -1, 1) * e.params.scaling_factor
for f32 in (False, True):
    x = orc.OracleEnv(f32=f32)
    x.hard_reset()
    x.sync(S, X, M)
    print("oracle f32=%s env_step k" % f32, x.env_step(a.copy(), vec_mode=True)[3])
    x.hard_reset()
    x.sync(S, X, M)
    errs = []
    for s in range(36):
        x.substep(T)
        errs.append(np.linalg.norm(T - x.get_state()[13:29]))
    print("  errs", np.round(errs, 5))
if sys.argv[-1] == "gpu":
    pkg = importlib.import_module("bullet-envs_amd")
    for B in (1, 16):
        st = pkg.Stepper(B)
        st.reset()
        st.set_state(np.repeat(S[None], B, 0), np.repeat(X[None], B, 0))
        st.set_manifold(np.repeat(M[None], B, 0))
        errs = []
        for s in range(36):
            info = st.substep(np.repeat(T[None], B, 0).astype(np.float32), 1)
            Sg, _ = st.get_state()
            errs.append(np.linalg.norm(T - Sg[0, 13:29]))
        print("gpu substep API B=%d errs" % B, np.round(errs, 5), "info", info[0])
        st.set_state(np.repeat(S[None], B, 0), np.repeat(X[None], B, 0))
        st.set_manifold(np.repeat(M[None], B, 0))
        o, r, d, k = st.step(np.repeat(a[None], B, 0).astype(np.float32))
        print("gpu env-step B=%d k" % B, k[:4], "ovf", st.contact_overflow())
        st.close()
