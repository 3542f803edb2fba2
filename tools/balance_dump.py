"""Dump what a scheduling study of env_step_kernel needs: for the bench workload, per env-step the
substep count of every env and the servo error the plan kernel sorts by.  Run on the GPU box:
    python tools/balance_dump.py gpurun_out/balance.npz
tests/golden/bench_substeps.npy (the fixture of tests/test_sched_model.py) is the first four launches of that
file's `substeps`, stored as int8.
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("bullet-envs_amd")
E, K, W = 4096, 24, 4
st = pkg.Stepper(E)
st.reset()
ids = np.arange(E)
subs, errs = [], []
scale = np.pi / 6
for j in range(W + K):
    a = bench.gait_actions(ids, j).astype(np.float32)
    S, _ = st.get_state()
    q = S[:, 13:29]
    tgt = np.zeros((E, 16), np.float32)
    tgt[:, 1::2] = np.clip(a, -1, 1) * scale
    err2 = ((tgt - q) ** 2).sum(axis=1)
    _, _, _, sub = st.step(a)
    if j >= W:
        subs.append(sub.copy()); errs.append(err2.copy())
np.savez(sys.argv[1], substeps=np.array(subs), err2=np.array(errs))
print("mean substeps", np.mean(subs), "max", np.max(subs), "min", np.min(subs))
