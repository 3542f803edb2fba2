"""Replicas of one overflow env-step scattered among ordinary environments: do the replicas agree?"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_env import gait
pkg = importlib.import_module("bullet-envs_amd")
B, n, A = 5000, 16, 8
os.environ["SNK_QUANTUM"] = os.environ.get("SNK_QUANTUM", "0")
fr = (0.5 + np.arange(B) % 11 / 10.0).astype(np.float32)
st = pkg.Stepper(B, n_modules=n)
st.reset(); st.set_ground_friction(fr)
a = (gait(range(B), 0, A) * 1.2).astype(np.float32)
S, X = st.get_state(); Mf = st.get_manifold()
o, r, d, s = st.step(a.copy())
# which envs overflowed in step 0?  replay each alone to find out
cand = []
one = pkg.Stepper(1, n_modules=n)
for e in (637, 1011, 1121, 1461, 1495, 1605, 1648, 1869, 2573, 2913, 3023, 3354, 3541, 4212, 4586):
    one.reset(); one.set_ground_friction(fr[e:e + 1]); one.set_state(S[e:e + 1], X[e:e + 1]); one.set_manifold(Mf[e:e + 1])
    c0 = one.contact_overflow()[0]
    one.step(a[e:e + 1].copy())
    if one.contact_overflow()[0] > c0:
        cand.append(e)
print("envs with an overflow substep in step 0:", cand)
for e in cand[:2]:
    idx = np.arange(7, B, 13)                     # 384 replicas scattered over the handle
    S2, X2, M2, a2, f2 = S.copy(), X.copy(), Mf.copy(), a.copy(), fr.copy()
    S2[idx], X2[idx], M2[idx], a2[idx], f2[idx] = S[e], X[e], Mf[e], a[e], fr[e]
    outs = []
    for rep in range(3):
        st.set_ground_friction(f2); st.set_state(S2, X2); st.set_manifold(M2)
        o2, r2, d2, s2 = st.step(a2.copy())
        outs.append(o2[idx].copy())
        print("   non-finite replicas:", int((~np.isfinite(o2[idx]).all(axis=1)).sum()), "non-finite ordinary envs:", int((~np.isfinite(o2).all(axis=1)).sum()) - int((~np.isfinite(o2[idx]).all(axis=1)).sum()))
        uniq, cnt = np.unique(np.nan_to_num(o2[idx], nan=12345.0), axis=0, return_counts=True)
        if rep == 0 and len(uniq) > 1:
            minority = uniq[np.argmin(cnt)]
            mn = idx[(o2[idx] == minority).all(axis=1)]
            G = 2048
            print("   minority replicas: position in the workgroup's sequence (index // %d) %s; workgroup %% 8 %s; workgroup %% 2 %s; first few %s"
                  % (G, np.bincount(mn // G, minlength=3), np.bincount(mn % G % 8, minlength=8), np.bincount(mn % 2, minlength=2), mn[:10]))
            print("   all replicas:      position %s; workgroup %% 8 %s" % (np.bincount(idx // G, minlength=3), np.bincount(idx % G % 8, minlength=8)))
            print("   |minority - majority| max %.3e" % np.abs(minority - uniq[np.argmax(cnt)]).max())
        print("   env %d: %d replicas among %d ordinary envs, pass %d: %d distinct outcomes, counts %s" % (e, len(idx), B - len(idx), rep, len(uniq), sorted(cnt)[::-1][:6]))
