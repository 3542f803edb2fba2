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


def save_state(env, path):
    """env: Stepper, SnakeVecEnv or DeviceVecEnv.  Writes one .npz."""
    st = _stepper(env)
    state, aux = st.get_state()
    # the EFFECTIVE friction, read back from the device (also covers values set through the C ABI directly)
    mf = st.get_manifold()           # the contact cache is simulator state too (contact_model 1)
    box = st.get_box() if st.params.obstacle == 2 else None      # ... and so is a free obstacle box
    np.savez_compressed(path, state=state, aux=aux, n_envs=np.int64(st.n_envs),
                        params=np.frombuffer(bytes(st.params), dtype=np.uint8),
                        ground_friction=st.get_ground_friction(),
                        manifold=np.zeros(0, np.float32) if mf is None else mf,
                        box_state=np.zeros(0, np.float32) if box is None else box[0],
                        box_manifold=np.zeros(0, np.float32) if box is None else box[1])


def load_state(env, path):
    """Restores a checkpoint into an env created with the same parameters and size."""
    st = _stepper(env)
    with np.load(path if str(path).endswith(".npz") else str(path) + ".npz") as z:
        if int(z["n_envs"]) != st.n_envs:
            raise ValueError("checkpoint holds %d environments, this handle %d" % (int(z["n_envs"]), st.n_envs))
        if z["params"].tobytes() != bytes(st.params):
            raise ValueError("checkpoint was written with different model parameters")
        mu = z["ground_friction"]
        # always restored: a checkpoint of a default-friction world must also undo the target's custom friction
        st.set_ground_friction(mu if mu.size else np.ones(st.n_envs, np.float32))
        st.set_state(z["state"], z["aux"])
        if "manifold" in z.files and z["manifold"].size:
            st.set_manifold(z["manifold"])
        if "box_state" in z.files and z["box_state"].size:
            st.set_box(z["box_state"], z["box_manifold"])
