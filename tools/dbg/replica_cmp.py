"""Replica outcomes of two builds (SNK_LIB_A / SNK_LIB_B), same inputs: are they the same bits?"""
import os, subprocess, sys
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import importlib
    sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
    from test_gpu_env import gait
    pkg = importlib.import_module("bullet-envs_amd")
    B, n, A = 5000, 16, 8
    os.environ["SNK_QUANTUM"] = "0"
    fr = (0.5 + np.arange(B) % 11 / 10.0).astype(np.float32)
    st = pkg.Stepper(B, n_modules=n)
    st.reset(); st.set_ground_friction(fr)
    a = (gait(range(B), 0, A) * 1.2).astype(np.float32)
    S, X = st.get_state(); Mf = st.get_manifold()
    outs = []
    for e in (637, 1011, 1121, 1461):
        idx = np.arange(7, B, 13)
        S2, X2, M2, a2, f2 = S.copy(), X.copy(), Mf.copy(), a.copy(), fr.copy()
        S2[idx], X2[idx], M2[idx], a2[idx], f2[idx] = S[e], X[e], Mf[e], a[e], fr[e]
        st.set_ground_friction(f2); st.set_state(S2, X2); st.set_manifold(M2)
        o, r, d, s = st.step(a2.copy())
        outs.append(o[idx])
    np.save(sys.argv[2], np.stack(outs))
    sys.exit(0)
res = {}
for tag in ("A", "B"):
    env = dict(os.environ, SNK_LIB=os.environ["SNK_LIB_" + tag], SNK_NO_PLAN="1")
    subprocess.check_call([sys.executable, __file__, "child", "/tmp/rep_%s.npy" % tag], env=env)
    res[tag] = np.load("/tmp/rep_%s.npy" % tag)
for i, e in enumerate((637, 1011, 1121, 1461)):
    a, b = res["A"][i], res["B"][i]
    print("env %d: build A distinct outcomes %d, build B %d; A's first replica == B's first replica: %s (max |d| %.3e)"
          % (e, len(np.unique(a, axis=0)), len(np.unique(b, axis=0)), np.array_equal(a[0], b[0]), np.abs(a[0] - b[0]).max()))
