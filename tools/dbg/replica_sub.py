"""Replicas of one overflow environment among ordinary ones, one substep at a time: where do the replicas part?"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_env import gait
pkg = importlib.import_module("bullet-envs_amd")
B, n, A = 5000, 16, 8
fr = (0.5 + np.arange(B) % 11 / 10.0).astype(np.float32)
st = pkg.Stepper(B, n_modules=n)
st.reset(); st.set_ground_friction(fr)
a = np.clip(gait(range(B), 0, A) * 1.2, -1, 1).astype(np.float32)
S, X = st.get_state(); Mf = st.get_manifold()
e = int(sys.argv[1]) if len(sys.argv) > 1 else 637
idx = np.arange(7, B, 13)
S2, X2, M2, a2, f2 = S.copy(), X.copy(), Mf.copy(), a.copy(), fr.copy()
S2[idx], X2[idx], M2[idx], a2[idx], f2[idx] = S[e], X[e], Mf[e], a[e], fr[e]
tg = np.zeros((B, n), np.float32)
tg[:, 1::2] = a2 * np.float32(np.pi / 6)
st.set_ground_friction(f2); st.set_state(S2, X2); st.set_manifold(M2)
for k in range(32):
    c0 = st.contact_overflow()[0]
    info = st.substep(tg, 1)
    s, x = st.get_state()
    m = st.get_manifold()
    us, cs = np.unique(s[idx], axis=0, return_counts=True)
    um, cm = np.unique(m[idx].reshape(len(idx), -1), axis=0, return_counts=True)
    print("substep %2d: replicas: iters %s contacts %s | distinct states %d %s, distinct contact caches %d %s | fallback substeps in the whole handle %d"
          % (k, np.unique(info[idx, 0]), np.unique(info[idx, 1]), len(us), sorted(cs)[::-1][:4], len(um), sorted(cm)[::-1][:4], st.contact_overflow()[0] - c0))
    if len(us) > 1:
        maj, mino = us[np.argmax(cs)], us[np.argmin(cs)]
        d = np.abs(maj - mino)
        print("   majority vs minority state: max |d| %.3e at index %d; base %s joints q %s qd %s" % (d.max(), d.argmax(), d[:13].max(), d[13:29].max(), d[29:45].max()))
        mm, mi = um[np.argmax(cm)], um[np.argmin(cm)]
        dm = np.abs(mm - mi).reshape(32, -1)
        print("   contact caches differ in cylinders", np.nonzero(dm.max(axis=1) > 0)[0], "max", dm.max())
        break
