"""Simulator-state checkpoint / restore.

The reference checkpoints only trainer state (`ppo/train.py:155-167`, `ars/train.py:171-173`);
a PyBullet world cannot be resumed mid-rollout.  Here the whole simulator state of N
environments is the per-env record (`snk_get_state`): pose, velocities, joint state, last
motor torques, joint-0 force, previous x for the reward -- plus the per-env plane friction and
the model parameters it was produced with.  Restoring it continues bit-for-bit.
`save_state` / `load_state` synchronise the device (the state accessors of the C ABI do).
"""
import numpy as np


def _stepper(obj):
    for name in ("stepper", "_stepper"):
        if hasattr(obj, name):
            return getattr(obj, name)
    return obj


FORMAT = 2      # 1 (rounds 2-5): params as the raw bytes of the C struct; 2: params field by field


def _params_fields(p):
    """snk_params as {field: value / list}, without the layout guard (struct_size, abi_version)."""
    out = {}
    for name, _ctype in p._fields_:
        if name in ("struct_size", "abi_version"):
            continue
        v = getattr(p, name)
        out[name] = [float(x) for x in v] if hasattr(v, "__len__") else v
    return out


def save_state(env, path):
    """env: Stepper, SnakeVecEnv or DeviceVecEnv.  Writes one .npz."""
    import json
    st = _stepper(env)
    state, aux = st.get_state()
    # the EFFECTIVE friction, read back from the device (also covers values set through the C ABI directly)
    mf = st.get_manifold()           # the contact cache is simulator state too (contact_model 1)
    box = st.get_box() if st.params.obstacle == 2 else None      # ... and so is a free obstacle box
    # The parameters travel field by field (JSON: names and values), not as the raw bytes of the C struct: a field
    # appended to snk_params in a later build must not make every older checkpoint unreadable.
    np.savez_compressed(path, state=state, aux=aux, n_envs=np.int64(st.n_envs), format=np.int64(FORMAT),
                        params_json=np.frombuffer(json.dumps(_params_fields(st.params), sort_keys=True).encode(), dtype=np.uint8),
                        ground_friction=st.get_ground_friction(),
                        manifold=np.zeros(0, np.float32) if mf is None else mf,
                        box_state=np.zeros(0, np.float32) if box is None else box[0],
                        box_manifold=np.zeros(0, np.float32) if box is None else box[1])


def _check_params(z, st):
    import json
    from . import _lib
    if "params_json" not in z.files:
        # format 1: the raw bytes of the struct as that build laid it out
        raw = z["params"].tobytes() if "params" in z.files else b""
        if raw == bytes(st.params):
            return
        raise ValueError("checkpoint format 1 (rounds 2-5): its parameters are the raw bytes of another build's snk_params "
                         "(%d bytes; this build's struct has %d) -- a LAYOUT mismatch, which cannot be told from a change "
                         "of values; re-save the checkpoint with the build that wrote it at format %d"
                         % (len(raw), len(bytes(st.params)), FORMAT))
    saved = json.loads(z["params_json"].tobytes().decode())
    mine = _params_fields(st.params)
    defaults = _params_fields(_lib.default_params(n_modules=int(saved.get("n_modules", st.params.n_modules))))
    unknown = sorted(set(saved) - set(mine))
    if unknown:
        raise ValueError("checkpoint carries parameters this build does not know: %s" % ", ".join(unknown))
    for name in mine:
        # a field the checkpoint's build did not have yet: it ran with what is the default now
        want = saved[name] if name in saved else defaults[name]
        if want != mine[name] and not (want != want and mine[name] != mine[name]):
            raise ValueError("checkpoint was written with different model parameters: `%s` is %r there%s, %r in this handle"
                             % (name, want, "" if name in saved else " (absent: that build's default)", mine[name]))


def load_state(env, path):
    """Restores a checkpoint into an env created with the same parameters and size."""
    st = _stepper(env)
    with np.load(path if str(path).endswith(".npz") else str(path) + ".npz") as z:
        if int(z["n_envs"]) != st.n_envs:
            raise ValueError("checkpoint holds %d environments, this handle %d" % (int(z["n_envs"]), st.n_envs))
        _check_params(z, st)
        mu = z["ground_friction"]
        # always restored: a checkpoint of a default-friction world must also undo the target's custom friction
        st.set_ground_friction(mu if mu.size else np.ones(st.n_envs, np.float32))
        st.set_state(z["state"], z["aux"])
        if "manifold" in z.files and z["manifold"].size:
            st.set_manifold(z["manifold"])
        if "box_state" in z.files and z["box_state"].size:
            st.set_box(z["box_state"], z["box_manifold"])
