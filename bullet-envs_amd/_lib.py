"""ctypes binding of libsnk.so (include/snk.h).

There is NO CPU fallback: if the HIP library cannot be loaded, importing the stepper
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNK_LIB selects an instrumented build of the same ABI (tools/profile_phases.py); default: libsnk.so
LIB_PATH = os.environ.get("SNK_LIB") or os.path.join(_HERE, "libsnk.so")


class SnkParams(C.Structure):
    """Mirror of `snk_params` (include/snk.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("n_modules", C.c_int32), ("inertia_from_file", C.c_int32),
        ("default_mass", C.c_double), ("collision_margin", C.c_double),
        ("hull_sides", C.c_int32), ("contact_model", C.c_int32),
        ("self_collision", C.c_int32), ("obstacle", C.c_int32),
        ("obstacle_pos", C.c_double * 3), ("obstacle_half", C.c_double * 3), ("mu_obstacle", C.c_double),
        ("obstacle_mass", C.c_double),
        ("dt", C.c_double), ("gravity_z", C.c_double),
        ("lin_damping", C.c_double), ("ang_damping", C.c_double),
        ("joint_damping", C.c_double), ("max_coord_vel", C.c_double),
        ("kp", C.c_double), ("kd", C.c_double), ("max_motor_impulse", C.c_double),
        ("joint_lo", C.c_double), ("joint_hi", C.c_double),
        ("limit_erp", C.c_double), ("limit_max_impulse", C.c_double),
        ("mu_link", C.c_double), ("aniso", C.c_double * 3),
        ("contact_erp", C.c_double), ("linear_slop", C.c_double),
        ("breaking_threshold", C.c_double), ("relative_breaking_threshold", C.c_int32), ("cone_friction", C.c_int32),
        ("n_iterations", C.c_int32), ("residual_threshold", C.c_double),
        ("warm_start", C.c_int32), ("warmstarting_factor", C.c_double), ("friction_directions", C.c_int32),
        ("scaling_factor", C.c_double), ("gait", C.c_int32),
        ("servo_tol", C.c_double), ("max_counter", C.c_int32),
        ("height_threshold", C.c_double), ("energy_dt", C.c_double),
        ("alpha", C.c_double), ("beta", C.c_double), ("gamma", C.c_double),
        ("term_angle", C.c_double), ("term_index", C.c_int32),
        ("collision_force", C.c_double), ("collision_penalty", C.c_double),
        ("done_penalty", C.c_double),
        ("contact_order", C.c_int32), ("reserved0", C.c_int32),
    ]


# every symbol include/snk.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_F = C.POINTER(C.c_float)
_U8 = C.POINTER(C.c_uint8)
_I32 = C.POINTER(C.c_int32)
_D = C.POINTER(C.c_double)
SYMBOLS = {
    "snk_default_params": (None, [C.POINTER(SnkParams)]),
    "snk_create": (C.c_int, [C.POINTER(SnkParams), C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "snk_destroy": (C.c_int, [_vp]),
    "snk_num_envs": (C.c_int32, [_vp]),
    "snk_obs_dim": (C.c_int32, [_vp]),
    "snk_act_dim": (C.c_int32, [_vp]),
    "snk_state_dim": (C.c_int32, [_vp]),
    "snk_record_floats": (C.c_int32, [_vp]),
    "snk_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "snk_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "snk_step_packed": (C.c_int, [_vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp]),
    "snk_reset_host": (C.c_int, [_vp, _U8, _F]),
    "snk_step_host": (C.c_int, [_vp, _F, _F, _F, _U8, _I32, C.c_int32]),
    "snk_substep_host": (C.c_int, [_vp, _F, C.c_int32, _I32]),
    "snk_get_state": (C.c_int, [_vp, _F, _F]),
    "snk_set_state": (C.c_int, [_vp, _F, _F]),
    "snk_manifold_floats": (C.c_int32, [_vp]),
    "snk_get_manifold": (C.c_int, [_vp, _F]),
    "snk_set_manifold": (C.c_int, [_vp, _F]),
    "snk_contact_overflow": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "snk_contact_histogram_bins": (C.c_int32, []),
    "snk_contact_histogram": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int32]),
    "snk_contact_histogram_enable": (C.c_int, [_vp, C.c_int32]),
    "snk_get_box": (C.c_int, [_vp, _F, _F]),
    "snk_set_box": (C.c_int, [_vp, _F, _F]),
    "snk_get_obs": (C.c_int, [_vp, _F]),
    "snk_mean_height": (C.c_int, [_vp, _F]),
    "snk_link_positions": (C.c_int, [_vp, _F]),
    "snk_joint3_reaction_fz": (C.c_int, [_vp, _F]),
    "snk_set_ground_friction": (C.c_int, [_vp, _F]),
    "snk_get_ground_friction": (C.c_int, [_vp, _F]),
    "snk_debug_set_tickets": (C.c_int, [_vp, C.c_uint32]),
    "snk_debug_raise_alarm": (C.c_int, [_vp]),
    "snk_selftest": (C.c_int, [C.c_int32]),
    "snk_timing_enable": (C.c_int, [_vp, C.c_int32]),
    "snk_timing_read": (C.c_int, [_vp, _D, _I32]),
    "snk_model_describe": (C.c_int, [_vp, _D, _D]),
    "snk_params_derived": (C.c_int, [C.c_void_p, _D]),
    "snk_last_error": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libsnk.so; raises RuntimeError with the reason when that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "bullet-envs_amd: %s is missing -- build it with `python bullet-envs_amd/build.py` "
            "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime (same SONAME libamdhip64.so.7 as
    # /opt/rocm's).  Two HSA runtimes cannot share a GPU in one process, so when torch is
    # installed it is imported FIRST: libsnk.so's NEEDED entry then binds to the copy torch
    # already mapped and the process has exactly one runtime.  Without torch, /opt/rocm's is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError("bullet-envs_amd: cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().snk_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, last_error()))


def default_params(**over):
    p = SnkParams()
    load().snk_default_params(C.byref(p))
    for k, v in over.items():
        if k in ("aniso", "obstacle_pos", "obstacle_half"):
            for i in range(3):
                getattr(p, k)[i] = v[i]
        else:
            if not hasattr(p, k):
                raise AttributeError("snk_params has no field %r" % k)
            setattr(p, k, v)
    return p


def fptr(a):
    return a.ctypes.data_as(_F)


class Stepper:
    """Thin owner of one `snk_handle`: N environments on one GPU (host-buffer API)."""

    def __init__(self, n_envs, device=0, params=None, **over):
        self.lib = load()
        self.params = params if params is not None else default_params(**over)
        h = _vp()
        check(self.lib.snk_create(C.byref(self.params), int(n_envs), int(device), C.byref(h)), "snk_create")
        self.h = h
        self.n_envs = int(n_envs)
        self.device = int(device)
        self.n = self.params.n_modules
        self.obs_dim = self.lib.snk_obs_dim(h)
        self.act_dim = self.lib.snk_act_dim(h)
        self.state_dim = self.lib.snk_state_dim(h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.snk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer calls ----
    def reset(self, mask=None):
        obs = np.zeros((self.n_envs, self.obs_dim), dtype=np.float32)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert m.shape == (self.n_envs,)
        check(self.lib.snk_reset_host(self.h, m.ctypes.data_as(_U8) if m is not None else None, fptr(obs)),
              "snk_reset_host")
        return obs

    def step(self, actions, vec_mode=True):
        """actions float32 [n_envs, act_dim], clipped in place.  Returns obs, rew, done, substeps."""
        assert actions.dtype == np.float32 and actions.flags.c_contiguous
        assert actions.shape == (self.n_envs, self.act_dim)
        obs = np.empty((self.n_envs, self.obs_dim), dtype=np.float32)
        rew = np.empty(self.n_envs, dtype=np.float32)
        done = np.empty(self.n_envs, dtype=np.uint8)
        sub = np.empty(self.n_envs, dtype=np.int32)
        check(self.lib.snk_step_host(self.h, fptr(actions), fptr(obs), fptr(rew), done.ctypes.data_as(_U8),
                                     sub.ctypes.data_as(_I32), 1 if vec_mode else 0), "snk_step_host")
        return obs, rew, done.astype(bool), sub

    def substep(self, targets, k=1):
        t = np.ascontiguousarray(targets, dtype=np.float32)
        assert t.shape == (self.n_envs, self.n)
        info = np.zeros((self.n_envs, 2), dtype=np.int32)
        check(self.lib.snk_substep_host(self.h, fptr(t), int(k), info.ctypes.data_as(_I32)), "snk_substep_host")
        return info

    def get_state(self):
        s = np.zeros((self.n_envs, self.state_dim), dtype=np.float32)
        a = np.zeros((self.n_envs, self.n + 2), dtype=np.float32)
        check(self.lib.snk_get_state(self.h, fptr(s), fptr(a)), "snk_get_state")
        return s, a

    def set_state(self, state=None, aux=None):
        s = a = None
        if state is not None:
            s = np.ascontiguousarray(state, dtype=np.float32)
            assert s.shape == (self.n_envs, self.state_dim)
        if aux is not None:
            a = np.ascontiguousarray(aux, dtype=np.float32)
            assert a.shape == (self.n_envs, self.n + 2)
        check(self.lib.snk_set_state(self.h, fptr(s) if s is not None else None, fptr(a) if a is not None else None),
              "snk_set_state")

    def get_manifold(self):
        """contact_model 1: [n_envs, 2n, 29] contact cache (see snk.h); None for a contact_model 0 handle."""
        if self.lib.snk_manifold_floats(self.h) == 0:
            return None
        m = np.zeros((self.n_envs, 2 * self.n, 29), dtype=np.float32)
        check(self.lib.snk_get_manifold(self.h, fptr(m)), "snk_get_manifold")
        return m

    def set_manifold(self, m):
        m = np.ascontiguousarray(m, dtype=np.float32)
        assert m.shape == (self.n_envs, 2 * self.n, 29)
        check(self.lib.snk_set_manifold(self.h, fptr(m)), "snk_set_manifold")

    def get_box(self):
        """obstacle 2: (state [n_envs, 13], manifold with the plane [n_envs, 29]) of the free box."""
        s = np.zeros((self.n_envs, 13), dtype=np.float32)
        m = np.zeros((self.n_envs, 29), dtype=np.float32)
        check(self.lib.snk_get_box(self.h, fptr(s), fptr(m)), "snk_get_box")
        return s, m

    def set_box(self, state=None, manifold=None):
        s = m = None
        if state is not None:
            s = np.ascontiguousarray(state, dtype=np.float32)
            assert s.shape == (self.n_envs, 13)
        if manifold is not None:
            m = np.ascontiguousarray(manifold, dtype=np.float32)
            assert m.shape == (self.n_envs, 29)
        check(self.lib.snk_set_box(self.h, fptr(s) if s is not None else None, fptr(m) if m is not None else None),
              "snk_set_box")

    def contact_overflow(self):
        """snk_contact_overflow since the handle was created: (substeps with more contact points than the solve's slots
        -- on a register-resident 16-link handle those are solved by the streamed-row solve with every point, so this is
        a count of slower substeps --, points left without rows (structurally zero: the streamed-row solve has a slot for
        every point the manifolds can hold), link-link / obstacle contacts left out).  The last two at zero mean Bullet's
        'no limit' held."""
        out = (C.c_uint64 * 3)()
        check(self.lib.snk_contact_overflow(self.h, out), "snk_contact_overflow")
        return int(out[0]), int(out[1]), int(out[2])

    def contact_histogram_enable(self, on=True):
        """Switches the per-substep counting on or off (off after creation: one atomic per substep, 0.7 % of the rate)."""
        check(self.lib.snk_contact_histogram_enable(self.h, 1 if on else 0), "snk_contact_histogram_enable")

    def contact_histogram(self, reset=False):
        """snk_contact_histogram: out[k] = physics substeps that ran with k contact points while the counting was enabled
        (since the last call with reset=True); the last bin collects everything beyond it."""
        n = int(self.lib.snk_contact_histogram_bins())
        out = (C.c_uint64 * n)()
        check(self.lib.snk_contact_histogram(self.h, out, 1 if reset else 0), "snk_contact_histogram")
        return np.array(out[:], dtype=np.uint64)

    def get_obs(self):
        o = np.zeros((self.n_envs, self.obs_dim), dtype=np.float32)
        check(self.lib.snk_get_obs(self.h, fptr(o)), "snk_get_obs")
        return o

    def mean_height(self):
        o = np.zeros(self.n_envs, dtype=np.float32)
        check(self.lib.snk_mean_height(self.h, fptr(o)), "snk_mean_height")
        return o

    def link_positions(self):
        """[n_envs, 3(n+1)]: getLinkPositions of every env ([x.., y.., z..] of links 0,3,...,3n)."""
        o = np.zeros((self.n_envs, 3 * (self.n + 1)), dtype=np.float32)
        check(self.lib.snk_link_positions(self.h, fptr(o)), "snk_link_positions")
        return o

    def joint3_reaction_fz(self):
        """Fz of the first motor joint's reaction (Bullet joint 3) after the last substep, [n_envs]."""
        o = np.zeros(self.n_envs, dtype=np.float32)
        check(self.lib.snk_joint3_reaction_fz(self.h, fptr(o)), "snk_joint3_reaction_fz")
        return o

    def set_ground_friction(self, mu):
        m = np.ascontiguousarray(mu, dtype=np.float32)
        assert m.shape == (self.n_envs,)
        check(self.lib.snk_set_ground_friction(self.h, fptr(m)), "snk_set_ground_friction")

    def get_ground_friction(self):
        """The effective per-env plane friction (ones unless set), read back from the device."""
        m = np.zeros(self.n_envs, dtype=np.float32)
        check(self.lib.snk_get_ground_friction(self.h, fptr(m)), "snk_get_ground_friction")
        return m

    def debug_set_tickets(self, base):
        check(self.lib.snk_debug_set_tickets(self.h, int(base) & 0xFFFFFFFF), "snk_debug_set_tickets")

    def debug_raise_alarm(self):
        """Test hook: the scheduler's alarm raised from the host (include/snk.h); the handle is poisoned afterwards."""
        check(self.lib.snk_debug_raise_alarm(self.h), "snk_debug_raise_alarm")

    def model_describe(self):
        bodies = np.zeros((self.n + 1, 10))
        origins = np.zeros((self.n + 1, 3))
        check(self.lib.snk_model_describe(self.h, bodies.ctypes.data_as(_D), origins.ctypes.data_as(_D)),
              "snk_model_describe")
        return bodies, origins

    # ---- device-pointer calls (torch tensors own the memory) ----
    def step_device(self, actions_ptr, obs_ptr, rew_ptr, done_ptr, sub_ptr=0, vec_mode=True, stream=0):
        check(self.lib.snk_step(self.h, actions_ptr, obs_ptr, rew_ptr, done_ptr, sub_ptr or None,
                                1 if vec_mode else 0, stream or None), "snk_step")

    def step_packed_device(self, actions_ptr, packed_ptr, row_stride, sub_ptr=0, vec_mode=True, stream=0):
        """snk_step_packed: rows [obs | reward f32 | done u32] of `row_stride` floats in one device buffer."""
        check(self.lib.snk_step_packed(self.h, actions_ptr, packed_ptr, int(row_stride), sub_ptr or None,
                                       1 if vec_mode else 0, stream or None), "snk_step_packed")

    def reset_device(self, mask_ptr=0, obs_ptr=0, stream=0):
        check(self.lib.snk_reset(self.h, mask_ptr or None, obs_ptr or None, stream or None), "snk_reset")

    def timing_enable(self, capacity):
        check(self.lib.snk_timing_enable(self.h, int(capacity)), "snk_timing_enable")

    def timing_read(self):
        ms = C.c_double()
        cnt = C.c_int32()
        check(self.lib.snk_timing_read(self.h, C.byref(ms), C.byref(cnt)), "snk_timing_read")
        return ms.value, cnt.value
