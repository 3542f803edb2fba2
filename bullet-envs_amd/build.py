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


# The flags every build of the translation unit uses (tools/dbg/loop_spills.sh asks for them: `build.py --print-flags`).
# -greedy-regclass-priority-trumps-globalness: the register-resident solve keeps 192 row registers + 16 motor columns
# alive across its loop; with the allocator's default priorities 4-12 of them ended up in scratch memory, reloaded
# in EVERY Gauss-Seidel iteration, and which ones changed with every edit of unrelated code (round 3: 314 k ...
# 341 k env-steps/s for the same arithmetic).  With this priority rule: two reloads per iteration, 342.8 k.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm",
         "-greedy-regclass-priority-trumps-globalness=1"]


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def _command(out, defines=(), extra=()):
    return [_hipcc()] + FLAGS + ["-shared", "-fPIC", os.path.join(CSRC, "snk_api.hip"), "-o", out] + \
        ["-D" + d for d in defines] + list(extra)


def _stamp(out):
    return out + ".cmd"


def toolchain_text():
    """What compiled the library: hipcc's own version lines (HIP version, clang version), kept next to the library
    (libsnk.so.toolchain) -- tests/test_gpu_bits.py keys its pinned hashes by it."""
    try:
        out = subprocess.check_output([_hipcc(), "--version"], text=True, stderr=subprocess.STDOUT)
    except (OSError, subprocess.CalledProcessError):
        return "unknown"
    keep = [ln.strip() for ln in out.splitlines() if ln.startswith("HIP version") or "clang version" in ln]
    return " | ".join(keep) if keep else "unknown"


def _stamp_text(cmd):
    """The command line as it is kept next to the library: paths relative to this directory, so that the copy of the tree
    on a GPU box (another absolute path) does not look like another build."""
    return " ".join(os.path.relpath(a, HERE) if os.path.isabs(a) and a.startswith(os.path.dirname(HERE)) else a for a in cmd)


def needs_build(cmd=None):
    """Stale when a source is newer than the library -- or when the library was built by ANOTHER command line (other
    flags or -D defines: mtimes cannot see that, and a libsnk.so from an experiment would travel to the GPU box as the
    product).  The command of the last build is kept next to the library (libsnk.so.cmd, git-ignored with it)."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(d) > t for d in sources() + [HEADER, os.path.abspath(__file__)]):
        return True
    try:
        with open(_stamp(LIB)) as f:
            return f.read() != _stamp_text(cmd if cmd is not None else _command(LIB))
    except OSError:
        return True


def build(force=False, verbose=False, defines=(), out=None):
    """defines/out: instrumented variants (e.g. -DSNK_PROFILE -> libsnk_prof.so, tools/profile_phases.py).  The default
    output (libsnk.so) is only ever built with the default command: variants and compiler experiments (SNK_EXTRA_FLAGS)
    need an `out` of their own."""
    extra = os.environ.get("SNK_EXTRA_FLAGS", "").split()        # compiler experiments
    if out is None and (defines or extra):
        raise RuntimeError("bullet-envs_amd/build.py: -D defines / SNK_EXTRA_FLAGS need an output of their own (out=...): "
                           "libsnk.so is the product and is built with the default flags only")
    target = out or LIB
    cmd = _command(target, defines, extra)
    if out is None and not force and not needs_build(cmd):
        if not os.path.exists(LIB + ".toolchain"):          # (a library from before round 6: same image, same compiler)
            with open(LIB + ".toolchain", "w") as f:
                f.write(toolchain_text())
        return LIB
    run = list(cmd)
    if verbose:
        run.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(run))
    subprocess.check_call(run)
    with open(_stamp(target), "w") as f:
        f.write(_stamp_text(cmd))
    with open(target + ".toolchain", "w") as f:
        f.write(toolchain_text())
    return target


if __name__ == "__main__":
    if "--print-flags" in sys.argv:
        print(" ".join(FLAGS))
    elif "--profile" in sys.argv:
        print(build(force=True, defines=("SNK_PROFILE",), out=os.path.join(HERE, "libsnk_prof.so")))
    elif "--sched-debug" in sys.argv:      # per-wave accounting of the step kernel's scheduler (tools/sched_stats.py)
        print(build(force=True, defines=("SNK_SCHED_DEBUG",), out=os.path.join(HERE, "libsnk_dbg.so")))
    else:
        build(force="--force" in sys.argv, verbose="-v" in sys.argv)
        print(LIB)
