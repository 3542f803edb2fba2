"""Hash of the state and outputs after a few gait steps (to compare two builds bit for bit):
    SNK_LIB=<lib> python tools/state_hash.py [16|32]"""
import hashlib, importlib, sys
import numpy as np
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module("bullet-envs_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = 2000
st = pkg.Stepper(B, n_modules=n)
st.reset()
st.set_ground_friction((0.5 + np.arange(B) % 11 / 10.0).astype(np.float32))
h = hashlib.sha256()
for j in range(4):
    o, r, d, s = st.step(bench.gait_actions(np.arange(B), j, n // 2).astype(np.float32))
    for a in (o, r, d, s):
        h.update(np.ascontiguousarray(a).tobytes())
S, X = st.get_state()
h.update(S.tobytes()); h.update(X.tobytes())
print(n, "links:", h.hexdigest()[:16], "mean substeps", s.mean())
