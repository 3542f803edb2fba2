"""Phase split of a physics substep on the states the bench's gait produces (the -DSNK_PROFILE build writes the phase
ticks of an env-step's last substep where the motor torques live):
    SNK_LIB=$PWD/bullet-envs_amd/libsnk_prof.so python tools/dbg/phases_gait.py [16|32]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.')
assert os.environ.get("SNK_LIB", "").endswith("libsnk_prof.so")
from bench import gait_actions
pkg = importlib.import_module("bullet-envs_amd")
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 16
names16 = ["contacts", "bias + ABA (factor, solve)", "sensor pass 1, v += a dt", "rows: motors (ABA deltas)",
           "rows: friction A (ABA deltas)", "  load 64 slots (A)", "rows: friction B (ABA deltas)", "  load 64 slots (B)",
           "rows: normals (ABA deltas)", "  load 32 slots + motor registers", "coupling scalars", "limit rows",
           "PGS, 50 iterations", "sensor pass 2: contact wrenches, FK", "sensor pass 2: bias + ABA", "integrate + FK"]
names32 = ["ground contacts", "link-link contacts (GJK)", "bias + ABA (factor, solve)", "sensor pass 1, v += a dt",
           "rows: M^-1 columns + assembly", "PGS, 50 iterations", "sensor pass 2", "integrate", "FK of the new pose"]
names = names16 if NL == 16 else names32
B = 4096
st = pkg.Stepper(B, n_modules=NL)
st.reset()
for j in range(8):
    o, r, d, s = st.step(gait_actions(np.arange(B), j, NL // 2).astype(np.float32))
_, aux = st.get_state()
t = aux[:, :len(names)].astype(np.float64)
ok = s > 0
m = t[ok].mean(axis=0)
print("%d links, gait, 4096 envs: ticks per phase of the last substep of env-step 8, mean over %d envs; total %.0f" % (NL, ok.sum(), m.sum()))
for n_, v in zip(names, m):
    print("   %-40s %9.0f  %5.1f %%" % (n_, v, 100 * v / m.sum()))
