"""Per-wave accounting of the scheduled step kernel (env_step_sched_kernel): time alive, time waiting in the
queue, slices, substeps, and the carry-on / requeue decisions at slice boundaries.
    python bullet-envs_amd/build.py --sched-debug      (here: builds libsnk_dbg.so with -DSNK_SCHED_DEBUG)
    SNK_LIB=bullet-envs_amd/libsnk_dbg.so [SNK_QUANTUM=q] python tools/sched_stats.py [16|32]      (on the GPU box)"""
import importlib, sys, os, ctypes as C
import numpy as np
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module("bullet-envs_amd")
E = 4096
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 16      # chain length: 16 or 32
st = pkg.Stepper(E, n_modules=NL)
st.reset()
ids = np.arange(E)
lib = st.lib
for j in range(6):
    a = bench.gait_actions(ids, j, NL // 2).astype(np.float32)
    o, r, d, sub = st.step(a)
    if j >= 3:
        buf = np.zeros((4096, 8), np.int64); n = C.c_int32()
        lib.snk_sched_stats(st.h, buf.ctypes.data_as(C.c_void_p), C.byref(n))
        w = buf[:n.value].astype(float)
        wait, alive, slices, subs = w[:, 0] / 100e3, w[:, 1] / 100e3, w[:, 2], w[:, 3]      # ms
        busy = alive - wait
        print("step %d: waves %d  env substeps mean %.2f | per wave: alive ms mean %.2f min %.2f max %.2f | waiting ms mean %.3f max %.3f | "
              "slices mean %.1f | substeps mean %.1f min %d max %d | ms per substep (busy/substeps) mean %.4f"
              % (j, n.value, sub.mean(), alive.mean(), alive.min(), alive.max(), wait.mean(), wait.max(), slices.mean(),
                 subs.mean(), subs.min(), subs.max(), (busy / np.maximum(subs, 1)).mean()))
        print("        boundary checks/wave %.1f  requeues/wave %.1f  mean top %.2f  mean remaining %.2f"
              % (w[:, 4].mean(), w[:, 5].mean(), w[:, 6].sum() / max(w[:, 4].sum(), 1), w[:, 7].sum() / max(w[:, 4].sum(), 1)))
os._exit(0)
