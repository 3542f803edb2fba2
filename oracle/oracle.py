"""ctypes binding of the CPU oracle (oracle/snake_oracle.cpp).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.  PARITY UNPINNED: see
snake_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcParams(C.Structure):
    _fields_ = [
        ("n_modules", C.c_int32), ("inertia_from_file", C.c_int32),
        ("default_mass", C.c_double), ("collision_margin", C.c_double),
        ("hull_sides", C.c_int32), ("contact_model", C.c_int32), ("max_contacts", C.c_int32),
        ("self_collision", C.c_int32), ("max_self_contacts", C.c_int32), ("obstacle", C.c_int32), ("pair_manifold", C.c_int32),
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
        ("warm_start", C.c_int32), ("warmstarting_factor", C.c_double),
        ("friction_directions", C.c_int32), ("contact_erp_rule", C.c_int32),
        ("scaling_factor", C.c_double), ("gait", C.c_int32),
        ("servo_tol", C.c_double), ("max_counter", C.c_int32),
        ("height_threshold", C.c_double), ("energy_dt", C.c_double),
        ("alpha", C.c_double), ("beta", C.c_double), ("gamma", C.c_double),
        ("term_angle", C.c_double), ("term_index", C.c_int32),
        ("collision_force", C.c_double), ("collision_penalty", C.c_double),
        ("done_penalty", C.c_double),
        ("noncontact_order", C.c_int32), ("contact_order", C.c_int32),
    ]


def build(force=False):
    """Compile liboracle.so / liboracle32.so in place (g++, seconds)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, f)) for f in ("liboracle.so", "liboracle32.so"))
    if not need:
        src = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("snake_oracle.cpp", "snake_oracle.h"))
        need = any(os.path.getmtime(os.path.join(_HERE, f)) < src for f in ("liboracle.so", "liboracle32.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


_libs = {}


def _load(f32=False):
    key = bool(f32)
    if key in _libs:
        return _libs[key]
    path = os.path.join(_HERE, "liboracle32.so" if f32 else "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    D = C.POINTER(C.c_double)
    I = C.POINTER(C.c_int32)
    vp = C.c_void_p
    sig = {
        "orc_default_params": (None, [C.POINTER(OrcParams)]),
        "orc_create": (vp, [C.POINTER(OrcParams)]),
        "orc_destroy": (None, [vp]),
        "orc_set_plane_friction": (None, [vp, C.c_double]),
        "orc_num_links": (C.c_int32, [vp]), "orc_num_dofs": (C.c_int32, [vp]),
        "orc_obs_dim": (C.c_int32, [vp]), "orc_state_dim": (C.c_int32, [vp]),
        "orc_get_state": (None, [vp, D]), "orc_set_state": (None, [vp, D]),
        "orc_get_aux": (None, [vp, D, D, D]), "orc_set_aux": (None, [vp, D, C.c_double, C.c_double]),
        "orc_hard_reset": (None, [vp]), "orc_reset": (None, [vp, D]),
        "orc_get_obs": (None, [vp, D]), "orc_mean_height": (C.c_double, [vp]),
        "orc_joint3_reaction_fz": (C.c_double, [vp]),
        "orc_substep": (None, [vp, D]),
        "orc_manifold_floats": (C.c_int32, [vp]),
        "orc_get_manifold": (None, [vp, D]), "orc_set_manifold": (None, [vp, D]),
        "orc_get_box_state": (None, [vp, D]), "orc_set_box_state": (None, [vp, D]),
        "orc_get_box_manifold": (None, [vp, D]), "orc_set_box_manifold": (None, [vp, D]),
        "orc_last_iterations": (C.c_int32, [vp]), "orc_last_num_contacts": (C.c_int32, [vp]),
        "orc_env_step": (None, [vp, D, C.c_int32, D, D, I, I]),
        "orc_link_com_world": (None, [vp, D]), "orc_joint_axes_world": (None, [vp, D, D]),
        "orc_link_inertials": (None, [vp, D]), "orc_link_parents": (None, [vp, I]),
        "orc_forward_dynamics": (None, [vp, D, C.c_int32, C.c_int32, D]),
        "orc_minv_mul": (None, [vp, D, D]),
        "orc_momentum": (None, [vp, D, D, D]),
        "orc_contacts": (C.c_int32, [vp, D, C.c_int32]),
        "orc_last_normal_impulses": (C.c_int32, [vp, D, C.c_int32]),
        "orc_cylinder_frames": (C.c_int32, [vp, D]),
        "orc_debug_gjk": (C.c_double, [vp, D, D, D]),
        "orc_contacts_full": (C.c_int32, [vp, D, C.c_int32]),
        "orc_last_contacts_full": (C.c_int32, [vp, D, C.c_int32]),
        "orc_bench_gait": (C.c_double, [C.POINTER(OrcParams), C.c_int32, D, D, C.c_int32, C.c_int32, C.c_int32,
                                        C.POINTER(C.c_int64), D]),
        "orc_quicksort_equal_keys": (None, [C.c_int32, I]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _libs[key] = lib
    return lib


def default_params(**over):
    p = OrcParams()
    _load().orc_default_params(C.byref(p))
    for k, v in over.items():
        if k in ("aniso", "obstacle_pos", "obstacle_half"):
            for i in range(3):
                getattr(p, k)[i] = v[i]
        else:
            setattr(p, k, v)
    return p


def quicksort_equal_keys(n):
    """The permutation btAlignedObjectArray::quickSort leaves on n equal keys (snake_oracle.cpp): out[k] = the original
    index of the element at position k."""
    out = (C.c_int32 * n)()
    _load().orc_quicksort_equal_keys(n, out)
    return list(out)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def bench_gait(n_envs, phases, warmup, steps, n_threads, mu_plane=None, params=None, want_agg=False, **over):
    """Times the C++ gait driver (orc_bench_gait): returns (seconds for `steps` batched steps, physics substeps)
    and, with want_agg, a dict of rollout aggregates over the timed steps."""
    p = params if params is not None else default_params(**over)
    ph = np.ascontiguousarray(phases, dtype=np.float64)
    assert ph.shape == (n_envs,)
    mu = None if mu_plane is None else np.ascontiguousarray(mu_plane, dtype=np.float64)
    sub = C.c_int64()
    agg = np.zeros(4)
    sec = _load().orc_bench_gait(C.byref(p), int(n_envs), _dp(ph), _dp(mu) if mu is not None else None,
                                 int(warmup), int(steps), int(n_threads), C.byref(sub), _dp(agg))
    if want_agg:
        tot = float(n_envs * steps)
        return sec, sub.value, dict(mean_substeps=sub.value / tot, episode_end_rate=agg[0] / tot,
                                    mean_reward=agg[1] / tot, mean_dx=agg[2] / tot, mean_contacts=agg[3] / tot)
    return sec, sub.value


class OracleEnv:
    """One environment of the CPU oracle."""

    def __init__(self, params=None, f32=False, **over):
        self.lib = _load(f32)
        self.params = params if params is not None else default_params(**over)
        self.h = self.lib.orc_create(C.byref(self.params))
        self.n = self.params.n_modules
        self.L = self.lib.orc_num_links(self.h)
        self.nd = self.lib.orc_num_dofs(self.h)
        self.obs_dim = self.lib.orc_obs_dim(self.h)
        self.state_dim = self.lib.orc_state_dim(self.h)
        self.act_dim = self.n // 2 if self.params.gait in (0, 1) else self.n

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass

    def set_plane_friction(self, mu):
        self.lib.orc_set_plane_friction(self.h, float(mu))

    def get_state(self):
        s = np.zeros(self.state_dim)
        self.lib.orc_get_state(self.h, _dp(s))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        assert s.shape == (self.state_dim,)
        self.lib.orc_set_state(self.h, _dp(s))

    def get_aux(self):
        tau = np.zeros(self.n)
        fz = C.c_double()
        px = C.c_double()
        self.lib.orc_get_aux(self.h, _dp(tau), C.byref(fz), C.byref(px))
        return tau, fz.value, px.value

    def set_aux(self, tau, fz, prev_x):
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        self.lib.orc_set_aux(self.h, _dp(tau), float(fz), float(prev_x))

    def hard_reset(self):
        self.lib.orc_hard_reset(self.h)

    def sync(self, state, aux=None, manifold=None):
        """Everything the simulator carries from one substep to the next, taken from the product's accessors of one env:
        state (snk_get_state), aux = [motor torques n, joint-0 Fz, prev obs x] and, for contact_model 1, the env's
        contact cache (snk_get_manifold)."""
        self.set_state(np.asarray(state, dtype=np.float64))
        if aux is not None:
            aux = np.asarray(aux, dtype=np.float64)
            self.set_aux(aux[:self.n], float(aux[self.n]), float(aux[self.n + 1]))
        if manifold is not None and self.params.contact_model == 1:
            self.set_manifold(np.asarray(manifold, dtype=np.float64))

    def get_manifold(self):
        """Contact cache of contact_model 1: [2n, 29] = per cylinder [count, 4 x (local point 3, ground point 3, applied
        normal impulse)]."""
        m = np.zeros((2 * self.n, 29))
        self.lib.orc_get_manifold(self.h, _dp(m))
        return m

    def set_manifold(self, m):
        m = np.ascontiguousarray(m, dtype=np.float64)
        assert m.shape == (2 * self.n, 29)
        self.lib.orc_set_manifold(self.h, _dp(m))

    def get_box(self):
        """obstacle 2: (state [pos3, quat xyzw 4, omega 3, vel 3], manifold with the plane [29])."""
        s, m = np.zeros(13), np.zeros(29)
        self.lib.orc_get_box_state(self.h, _dp(s))
        self.lib.orc_get_box_manifold(self.h, _dp(m))
        return s, m

    def set_box(self, state, manifold=None):
        s = np.ascontiguousarray(state, dtype=np.float64)
        assert s.shape == (13,)
        self.lib.orc_set_box_state(self.h, _dp(s))
        if manifold is not None:
            m = np.ascontiguousarray(manifold, dtype=np.float64)
            assert m.shape == (29,)
            self.lib.orc_set_box_manifold(self.h, _dp(m))

    def reset(self):
        o = np.zeros(self.obs_dim)
        self.lib.orc_reset(self.h, _dp(o))
        return o

    def get_obs(self):
        o = np.zeros(self.obs_dim)
        self.lib.orc_get_obs(self.h, _dp(o))
        return o

    def mean_height(self):
        return self.lib.orc_mean_height(self.h)

    def joint3_reaction_fz(self):
        return self.lib.orc_joint3_reaction_fz(self.h)

    def substep(self, targets):
        t = np.ascontiguousarray(targets, dtype=np.float64)
        assert t.shape == (self.n,)
        self.lib.orc_substep(self.h, _dp(t))

    @property
    def last_iterations(self):
        return self.lib.orc_last_iterations(self.h)

    @property
    def last_num_contacts(self):
        return self.lib.orc_last_num_contacts(self.h)

    def env_step(self, action, vec_mode=False):
        a = np.ascontiguousarray(action, dtype=np.float64).reshape(-1).copy()
        assert a.shape == (self.act_dim,)
        o = np.zeros(self.obs_dim)
        r = C.c_double()
        d = C.c_int32()
        k = C.c_int32()
        self.lib.orc_env_step(self.h, _dp(a), 1 if vec_mode else 0, _dp(o), C.byref(r), C.byref(d), C.byref(k))
        return o, r.value, bool(d.value), k.value, a

    def link_com_world(self):
        out = np.zeros((self.L, 3))
        self.lib.orc_link_com_world(self.h, _dp(out))
        return out

    def joint_axes_world(self):
        ax = np.zeros((self.n, 3))
        org = np.zeros((self.n, 3))
        self.lib.orc_joint_axes_world(self.h, _dp(ax), _dp(org))
        return ax, org

    def link_inertials(self):
        out = np.zeros((self.L, 7))
        self.lib.orc_link_inertials(self.h, _dp(out))
        return out

    def link_parents(self):
        out = np.zeros(self.L, dtype=np.int32)
        self.lib.orc_link_parents(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)))
        return out

    def forward_dynamics(self, tau=None, gravity=True, damping=True):
        tau = np.zeros(self.n) if tau is None else np.ascontiguousarray(tau, dtype=np.float64)
        acc = np.zeros(self.nd)
        self.lib.orc_forward_dynamics(self.h, _dp(tau), int(gravity), int(damping), _dp(acc))
        return acc

    def minv_mul(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.nd)
        self.lib.orc_minv_mul(self.h, _dp(x), _dp(y))
        return y

    def momentum(self):
        lin = np.zeros(3)
        ang = np.zeros(3)
        k = C.c_double()
        self.lib.orc_momentum(self.h, _dp(lin), _dp(ang), C.byref(k))
        return lin, ang, k.value

    def contacts(self, maxc=256):
        out = np.zeros((maxc, 5))
        nc = self.lib.orc_contacts(self.h, _dp(out), maxc)
        return out[:nc]

    def cylinder_frames(self):
        """[2n, 12]: world centre (3) and rotation (9, row-major) of every collision cylinder, link order."""
        out = np.zeros((2 * self.n, 12))
        self.lib.orc_cylinder_frames(self.h, _dp(out))
        return out

    def debug_gjk(self, fa, fb):
        """Core distance (no margins) between two cylinders given as [centre 3, rotation 9]; (dist, pa, pb)."""
        fa = np.ascontiguousarray(fa, dtype=np.float64); fb = np.ascontiguousarray(fb, dtype=np.float64)
        out = np.zeros(6)
        d = self.lib.orc_debug_gjk(self.h, _dp(fa), _dp(fb), _dp(out))
        return d, out[:3], out[3:]

    def contacts_full(self, maxc=512):
        """[nc, 12]: P, dist, link, linkB, normal, PB of every contact of the current pose."""
        out = np.zeros((maxc, 12))
        nc = self.lib.orc_contacts_full(self.h, _dp(out), maxc)
        return out[:nc]

    def last_contacts_full(self, maxc=512):
        """[nc, 12]: the contacts the last substep solved (same record as contacts_full; linkB -1 ground, -2 the box)."""
        out = np.zeros((maxc, 12))
        nc = self.lib.orc_last_contacts_full(self.h, _dp(out), maxc)
        return out[:nc]

    def last_normal_impulses(self, maxc=256):
        out = np.zeros(maxc)
        nc = self.lib.orc_last_normal_impulses(self.h, _dp(out), maxc)
        return out[:nc]
