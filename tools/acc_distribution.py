"""Distribution of GPU-float32 vs oracle-float64 errors after K physics substeps on the ground (512 random
states; K = 1 and 3: under the default contact model the cache starts empty and fills over the first substeps), next to
oracle-float32 vs oracle-float64 on the same inputs.   python tools/acc_distribution.py [states] [links]  The maxima are heavy-tailed
(stick-slip states amplify round-off by 1e5); medians and 90th percentiles are what to compare
between builds."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import oracle as orc
from conftest import random_state
pkg = importlib.import_module("bullet-envs_amd")
n, B = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), (int(sys.argv[1]) if len(sys.argv) > 1 else 512)
rng = np.random.default_rng(4321)
S = np.zeros((B, 13 + 2 * n))
for i in range(B):
    S[i] = random_state(rng, n, z=float(os.environ.get("ACC_Z", "0.026")), qamp=float(os.environ.get("ACC_QAMP", "0.3")),
                        vamp=float(os.environ.get("ACC_VAMP", "0.3")), flat=os.environ.get("ACC_FLAT", "1") == "1")
    S[i, 9] *= 0.1; S[i, 7:9] *= 0.1
S32 = S.astype(np.float32)
T = rng.uniform(-0.5, 0.5, (B, n)).astype(np.float32)
for K in (1, 3):
    over = dict(self_collision=int(os.environ.get("ACC_SELF_COLLISION", "1")))       # (experiments: link-link contacts off)
    over.update(json.loads(os.environ.get("ACC_OVER", "{}")))                       # (... any other parameter set)
    st = pkg.Stepper(B, n_modules=n, residual_threshold=0.0, **over)
    MU = rng.uniform(0.5, 1.5, B).astype(np.float32) if os.environ.get("ACC_MU") else None     # per-env plane friction
    if MU is not None:
        st.set_ground_friction(MU)
    st.set_state(S32); st.substep(T, K)
    G, GX = st.get_state()
    st.close()
    o = orc.OracleEnv(n_modules=n, residual_threshold=0.0, max_contacts=0, **over); o32 = orc.OracleEnv(n_modules=n, residual_threshold=0.0, max_contacts=0, f32=True, **over)
    eg, e32, fg, f32_ = [], [], [], []
    for i in range(B):
        o.hard_reset(); o32.hard_reset()   # (an empty contact cache, as the device's after set_state on a fresh handle)
        if MU is not None:
            o.set_plane_friction(float(MU[i])); o32.set_plane_friction(float(MU[i]))
        o.set_state(S32[i].astype(np.float64)); o32.set_state(S32[i].astype(np.float64))
        for _ in range(K):
            o.substep(T[i].astype(np.float64)); o32.substep(T[i].astype(np.float64))
        r = o.get_state(); r32 = o32.get_state()
        f = lambda x: (np.abs(x[13 + n:] - r[13 + n:]) / (1 + np.abs(r[13 + n:]))).max()
        eg.append(f(G[i])); e32.append(f(r32))
        # the joint-0 force sensor (obs[55] / obs[103]) and the motor torques of the last substep: aux = [torques n, fz, ..]
        ta, fza, _ = o.get_aux(); tb, fzb, _ = o32.get_aux()
        sc = 1.0 + np.abs(ta).max() + abs(fza)
        fg.append(max(np.abs(GX[i, :n] - ta).max(), abs(GX[i, n] - fza)) / sc)
        f32_.append(max(np.abs(tb - ta).max(), abs(fzb - fza)) / sc)
    for name, e, ff in (("GPU float32   ", np.array(eg), np.array(fg)), ("oracle float32", np.array(e32), np.array(f32_))):
        print("K = %d  " % K + name, "median %.3e  p90 %.3e  p99 %.3e  max %.3e" % (np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()),
              "| torques + force sensor: median %.3e  p90 %.3e" % (np.median(ff), np.percentile(ff, 90)))
