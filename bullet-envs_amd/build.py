"""Builds libsnk.so (HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box
with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsnk.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "snk.h")


def sources():
    """Every file under csrc/ is a dependency of the one translation unit (snk_api.hip includes the rest): a stale
    libsnk.so after an edit of ANY of them would travel to the GPU box unnoticed (it is git-ignored, not gpurun-ignored)."""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + [HEADER, os.path.abspath(__file__)])


def build(force=False, verbose=False, defines=(), out=None):
    """defines/out: instrumented variants (e.g. -DSNK_PROFILE -> libsnk_prof.so, tools/profile_phases.py)."""
    if out is None and not defines and not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    # -greedy-regclass-priority-trumps-globalness: the register-resident solve keeps 192 row registers + 16 motor columns
    # alive across its loop; with the allocator's default priorities 4-12 of them ended up in scratch memory, reloaded
    # in EVERY Gauss-Seidel iteration, and which ones changed with every edit of unrelated code (round 3: 314 k ...
    # 341 k env-steps/s for the same arithmetic).  With this priority rule: two reloads per iteration, 342.8 k.
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm",
           "-greedy-regclass-priority-trumps-globalness=1", "-shared", "-fPIC",
           os.path.join(CSRC, "snk_api.hip"), "-o", out or LIB] + ["-D" + d for d in defines]
    cmd += os.environ.get("SNK_EXTRA_FLAGS", "").split()        # compiler experiments
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    if "--profile" in sys.argv:
        print(build(force=True, defines=("SNK_PROFILE",), out=os.path.join(HERE, "libsnk_prof.so")))
    elif "--sched-debug" in sys.argv:      # per-wave accounting of the step kernel's scheduler (tools/sched_stats.py)
        print(build(force=True, defines=("SNK_SCHED_DEBUG",), out=os.path.join(HERE, "libsnk_dbg.so")))
    else:
        build(force="--force" in sys.argv, verbose="-v" in sys.argv)
        print(LIB)
