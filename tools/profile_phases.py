"""Phase durations inside one physics substep (s_memtime ticks, wave 0 of each env), from the
-DSNK_PROFILE build:  python bullet-envs_amd/build.py --profile  (here), then on the GPU box
SNK_LIB=bullet-envs_amd/libsnk_prof.so python tools/profile_phases.py [16|32]
(16: the register-resident solve's substep; 32: the streamed-row solve's)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.')
assert os.environ.get("SNK_LIB", "").endswith("libsnk_prof.so"), "run with SNK_LIB=bullet-envs_amd/libsnk_prof.so"
pkg = importlib.import_module("bullet-envs_amd")
names = ["contacts", "bias + ABA (factor, solve)", "sensor pass 1, v += a dt", "rows: motors (ABA deltas)",
         "rows: friction A (ABA deltas)", "  load 64 slots (A)", "rows: friction B (ABA deltas)", "  load 64 slots (B)",
         "rows: normals (ABA deltas)", "  load 32 slots + motor registers", "coupling scalars", "limit rows",
         "PGS, 50 iterations", "sensor pass 2: contact wrenches, FK", "sensor pass 2: bias + ABA", "integrate + FK"]
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if NL == 32:
    names = ["ground contacts", "link-link contacts (GJK)", "bias + ABA (factor, solve)", "sensor pass 1, v += a dt",
             "rows: M^-1 columns (38 delta sweeps) + assembly, lane = dof", "PGS, 50 iterations", "sensor pass 2", "integrate", "FK of the new pose"]
for B in (1024, 2048):
    st = pkg.Stepper(B, n_modules=NL, residual_threshold=0.0)
    st.reset()
    T = np.zeros((B, NL), np.float32); T[:, 1::2] = 0.3
    st.substep(T, 3)
    _, aux = st.get_state()
    t = aux[:, :len(names)].astype(np.float64)
    m = t.mean(axis=0)
    print("B=%d (%d wave(s)/SIMD): ticks per phase, mean over envs; total %.0f" % (B, B // 1024, m.sum()))
    for n_, v in zip(names, m):
        print("   %-36s %9.0f  %5.1f %%" % (n_, v, 100 * v / m.sum()))
    st.close()
