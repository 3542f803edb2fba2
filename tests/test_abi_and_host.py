"""CPU-side tests of the boundary: the C-ABI library loads and exports every symbol that
include/snk.h declares, defaults mirror the reference, and the product never falls back to
a CPU path.  No compute calls are made (there is no GPU here)."""
import ctypes
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "snk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(snk_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libsnk.so does not export %s" % name
    # and the binding table in _lib.py covers the header exactly
    from importlib import import_module
    _lib = import_module("bullet-envs_amd._lib")
    assert sorted(_lib.SYMBOLS) == declared


def test_default_params_mirror_reference(pkg):
    p = pkg.default_params()
    assert p.n_modules == 16 and p.gait == 1                       # snake.py:16,62
    assert abs(p.scaling_factor - np.pi / 6) < 1e-15               # snake.py:63
    assert p.dt == 1.0 / 240.0 and p.gravity_z == -9.8             # F2; snake.py:8
    assert (p.kp, p.kd) == (0.1, 1.0) and np.isinf(p.max_motor_impulse)   # F3
    assert p.mu_link == 2.0 and list(p.aniso) == [1.0, 0.1, 0.01]  # snake.py:25,104-106
    assert (p.alpha, p.beta, p.gamma) == (1.0, 0.01, 0.1)          # SnakeGymEnv.py:14-16
    assert p.servo_tol == 0.05 and p.max_counter == 40 and p.height_threshold == 0.1
    assert p.term_index == 9 and p.term_angle == 0.5 and p.done_penalty == -5.0
    assert p.energy_dt == 0.01 and p.collision_force == 10.0 and p.collision_penalty == -10.0


def test_params_struct_layout_matches_c(pkg):
    """sizeof(snk_params) AND the offset of every field as ctypes sees them equal the C compiler's (a field
    inserted in the middle shifts every later one without changing what a size check sees in the fields before it)."""
    names = [n for n, _ in pkg.SnkParams._fields_]
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "snk.h"\nint main(){printf("%zu %d", sizeof(snk_params), SNK_ABI_VERSION);\n'
           + "".join('printf(" %%zu", offsetof(snk_params, %s));\n' % n for n in names) + "return 0;}\n")
    exe = os.path.join("/tmp", "snk_sizeof_%d" % os.getpid())
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe],
                   input=src.encode(), check=True)
    out = [int(x) for x in subprocess.check_output([exe]).split()]
    os.remove(exe)
    assert ctypes.sizeof(pkg.SnkParams) == out[0]
    for n, off in zip(names, out[2:]):
        assert getattr(pkg.SnkParams, n).offset == off, (n, getattr(pkg.SnkParams, n).offset, off)
    assert names[:2] == ["struct_size", "abi_version"]
    # snk_default_params stamps the struct with this library's size and version (snk_create refuses anything else)
    p = pkg.default_params()
    assert p.struct_size == out[0] and p.abi_version == out[1]


def test_oracle_and_product_defaults_agree(pkg, oracle_mod):
    a, b = pkg.default_params(), oracle_mod.default_params()
    for name, _ in pkg.SnkParams._fields_:
        if name.startswith("reserved"):
            assert getattr(a, name) == 0
            continue
        if name in ("struct_size", "abi_version"):      # the C ABI's layout guard: the oracle has its own struct
            continue
        va, vb = getattr(a, name), getattr(b, name)
        if name in ("aniso", "obstacle_pos", "obstacle_half"):
            assert list(va) == list(vb), name
        else:
            assert va == vb, name


def test_no_cpu_fallback(pkg):
    """Without a GPU the stepper must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError) as ei:
        pkg.Stepper(4)
    assert "snk_create" in str(ei.value)
    with pytest.raises(RuntimeError):
        pkg.SnakeVecEnv(4)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkgdir = os.path.join(ROOT, "bullet-envs_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                for line in txt.splitlines():
                    s = line.strip()
                    if s.startswith(("import ", "from ", "#include")):
                        assert "oracle" not in s, (f, s)


def test_action_matrix_shapes(pkg):
    se = importlib.import_module("bullet-envs_amd.snake_env")
    a = se._as_action_matrix(np.zeros((4, 8)), 4, 8)                 # PPO  (ppo/train.py:122)
    assert a.shape == (4, 8) and a.dtype == np.float32
    a = se._as_action_matrix(np.zeros((4, 8, 1)), 4, 8)              # ARS  (ars/train.py:95-99)
    assert a.shape == (4, 8)
    a = se._as_action_matrix([np.zeros((8, 1)) for _ in range(4)], 4, 8)
    assert a.shape == (4, 8)
    with pytest.raises(ValueError):
        se._as_action_matrix(np.zeros((4, 7)), 4, 8)


def test_params_from_reference_args(pkg):
    class Args:  # the fields ppo/params.py:5-46 defines that the hot path reads
        alpha, beta, gamma = 2.0, 0.02, 0.3
        gaitSelection, scaling_factor, mode = 0, 4.0, 'train'
        motorVelocityLimit, motorTorqueLimit = np.inf, np.inf
        kp, kd = 10, 0.1      # dead in the reference (F3): must NOT reach the physics
    p = pkg.params_from_args(Args())
    assert (p.alpha, p.beta, p.gamma, p.gait) == (2.0, 0.02, 0.3, 0)
    assert abs(p.scaling_factor - np.pi / 4) < 1e-15
    assert (p.kp, p.kd) == (0.1, 1.0)


def test_snake_facade_spaces(pkg):
    robot = pkg.Snake(None, "ignored.urdf", None)
    assert robot.motorList == list(range(3, 49, 3))                 # snake.py:80
    ub = robot.getObservationUpperBound()
    assert ub.shape == (56,) and ub[0] == np.pi and np.isinf(ub[16]) and ub[48] == 1.0


def test_bench_action_stream():
    sys.path.insert(0, ROOT)
    import bench
    ids = np.arange(8)
    a = bench.gait_actions(ids, 3)
    assert a.shape == (8, 8)
    k = np.arange(8)
    assert np.allclose(a[0], -np.sin((2 * k + 1) * 4.0 + 0.6))      # env 0 is the canonical gait
    assert np.array_equal(bench.gait_actions(ids, 3), a)            # counter-based: reproducible
    assert np.array_equal(bench.gait_actions(np.array([5]), 3)[0], a[5])   # keyed by global index
    assert np.abs(a).max() <= 1.0


def test_contact_threshold_follows_the_urdf_geometry(pkg):
    """The relative contact breaking threshold (the default [U]) is 0.02 x the link collider's angular-motion disc; the
    product derives it from the SAME cylinder constants its model and hull use, and those are the generated URDF's
    (oracle/urdf_gen.py = snake.urdf:806-811): 0.02 x (0.0183 + |(0.027, 0.027, 0.0175)|) = 1.2061 mm."""
    import ctypes as C
    import xml.etree.ElementTree as ET
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import urdf_gen
    lib = pkg.load()
    from importlib import import_module
    _lib = import_module("bullet-envs_amd._lib")
    cyl = [c for c in ET.fromstring(urdf_gen.snake_urdf(16)).iter("collision") if c.find("geometry/cylinder") is not None][0]
    r, length = float(cyl.find("geometry/cylinder").get("radius")), float(cyl.find("geometry/cylinder").get("length"))
    z = float(cyl.find("origin").get("xyz").split()[2])
    for n in (16, 32):
        p = _lib.default_params(n_modules=n)
        out = (C.c_double * 6)()
        assert lib.snk_params_derived(C.byref(p), out) == 0
        thr, cr, chl, cz, mg, bthr = list(out)
        assert abs(cr - r) < 1e-7 and abs(chl - length / 2) < 1e-7 and abs(cz - z) < 1e-7 and abs(mg - 0.001) < 1e-9
        want = 0.02 * (z + np.sqrt(2 * (r + 0.001) ** 2 + (length / 2 + 0.001) ** 2))
        assert abs(thr - want) < 1e-9 and abs(thr - 1.20606e-3) < 1e-8
        assert abs(bthr - 0.02 * np.sqrt(0.1 ** 2 + 0.4 ** 2 + 0.1 ** 2)) < 1e-8
        p2 = _lib.default_params(n_modules=n, relative_breaking_threshold=0)
        assert lib.snk_params_derived(C.byref(p2), out) == 0 and abs(out[0] - 0.02) < 1e-9


def test_vec_env_module_surface(pkg):
    """ppo/multiprocessing_env.py's module-level names a caller can import: VecEnv (its four calls left empty, step =
    step_async + step_wait, :31-80), CloudpickleWrapper (:83-94: .x, pickled through cloudpickle), SubprocVecEnv."""
    import pickle
    v = pkg.VecEnv(3, "obs-space", "act-space")
    assert (v.num_envs, v.observation_space, v.action_space) == (3, "obs-space", "act-space")
    assert v.reset() is None and v.step_async([0]) is None and v.step_wait() is None and v.close() is None
    assert v.step([0]) is None
    k = 7
    w = pkg.CloudpickleWrapper(lambda: k + 1)                 # a closure: plain pickle refuses it
    with pytest.raises(Exception):
        pickle.dumps(w.x)
    w2 = pickle.loads(pickle.dumps(w))
    assert w2.x() == 8 and w2() == 8 and w.x() == 8
    assert issubclass(pkg.SubprocVecEnv, pkg.VecEnv)


def test_bench_plain_multi_gpu_invocation_refuses_cleanly_without_the_gpus():
    """The same plain invocation where the GPUs are not there (this container: none; a 1-GPU box: one): the ranks
    refuse with a message naming the counts, the parent relays the launcher's non-zero exit code and prints no JSON."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs here: the invocation would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SNK_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "--gpus 2 but this node shows" in out.stderr


def test_subproc_vec_env_refuses_thunks_that_differ(pkg):
    """The reference forks one process per thunk and honours each one's own settings (ppo/multiprocessing_env.py:106-111);
    one GPU handle runs ONE parameter set, so thunks that differ are refused -- naming the thunk and the first differing
    field -- instead of being replaced silently by the first.  No device needed: the refusal comes
    before the handle is created; the stand-ins carry what a bullet-envs_amd SnakeGymEnv carries (params, mode, close)."""
    from importlib import import_module
    _lib = import_module("bullet-envs_amd._lib")
    closed = []

    class Proto(object):
        def __init__(self, mode="train", **over):
            self.params = _lib.default_params(**over)
            self.mode = mode

        def close(self):
            closed.append(self)

    def thunk(**kw):
        return lambda: Proto(**kw)

    with pytest.raises(ValueError, match=r"env_fns\[3\] differs from env_fns\[0\] in `gait` \(0 vs 1\)"):
        pkg.SubprocVecEnv([thunk(), thunk(), thunk(), thunk(gait=0)])
    assert len(closed) == 4                       # every probed thunk's env was closed again
    with pytest.raises(ValueError, match=r"env_fns\[1\] differs from env_fns\[0\] in `alpha`"):
        pkg.SubprocVecEnv([thunk(), pkg.CloudpickleWrapper(thunk(alpha=2.0)), thunk()])
    with pytest.raises(ValueError, match=r"env_fns\[1\] differs from env_fns\[0\] in `aniso`"):
        pkg.SubprocVecEnv([thunk(), thunk(aniso=[1.0, 0.1, 0.02])])
    with pytest.raises(ValueError, match=r"in `mode` \('test' vs 'train'\)"):
        pkg.SubprocVecEnv([thunk(), thunk(mode="test")])
    # many thunks: the first, the last and evenly spaced ones in between are probed -- a different LAST one is caught
    closed.clear()
    with pytest.raises(ValueError, match=r"env_fns\[299\] differs from env_fns\[0\] in `beta`"):
        pkg.SubprocVecEnv([thunk()] * 299 + [thunk(beta=0.5)])
    assert len(closed) <= pkg.SubprocVecEnv.kHeteroProbe + 2
    with pytest.raises(TypeError, match="must build bullet-envs_amd SnakeGymEnv objects"):
        pkg.SubprocVecEnv([lambda: object()])
    with pytest.raises(ValueError, match="no env_fns"):
        pkg.SubprocVecEnv([])
    # identical thunks pass the guard and reach the handle's creation: without a GPU that fails loudly (no fallback)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            pkg.SubprocVecEnv([thunk(), thunk()])
