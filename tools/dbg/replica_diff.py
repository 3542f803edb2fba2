"""One environment's env-step replicated over a whole handle: do the replicas end on the same bits?  (A difference is a
timing-dependent fault: a hazard, a race, a read of something never written.)
   python tools/dbg/replica_diff.py [B replicas]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_env import gait
pkg = importlib.import_module("bullet-envs_amd")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B, n, A = 5000, 16, 8
os.environ["SNK_QUANTUM"] = "0"
fr = (0.5 + np.arange(B) % 11 / 10.0).astype(np.float32)


def run(upto):
    st = pkg.Stepper(B, n_modules=n)
    st.reset(); st.set_ground_friction(fr)
    pre = None
    for j in range(upto + 1):
        a = (gait(range(B), j, A) * 1.2).astype(np.float32)
        if j == upto:
            pre = (st.get_state(), st.get_manifold(), a.copy())
        o, r, d, s = st.step(a)
    st.close()
    return o, pre


for step in (0, 1):
    o1, pre = run(step)
    o2, _ = run(step)
    bad = np.nonzero((o1 != o2).any(axis=1))[0]
    print("step %d: envs that differ between two runs: %s" % (step, bad[:12]))
    for e in bad[:3]:
        (S, X), Mf, a = pre
        rp = pkg.Stepper(R, n_modules=n)
        rp.reset()
        rp.set_ground_friction(np.full(R, fr[e], np.float32))
        rp.set_state(np.repeat(S[e:e + 1], R, 0), np.repeat(X[e:e + 1], R, 0))
        rp.set_manifold(np.repeat(Mf[e:e + 1], R, 0))
        o, r, d, s = rp.step(np.repeat(a[e:e + 1], R, 0).copy())
        ov = rp.contact_overflow()
        uniq = np.unique(o, axis=0)
        cnt = [int((o == u).all(axis=1).sum()) for u in uniq]
        print("   env %d replicated %d x: %d distinct outcomes, counts %s; substeps %s; overflow counters %s; spread %.2e"
              % (e, R, len(uniq), cnt[:8], np.unique(s), ov, np.abs(o - o[0]).max()))
        rp.close()
