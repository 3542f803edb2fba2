"""Simulator-state checkpoint / restore.

The reference checkpoints only trainer state (`ppo/train.py:155-167`, `ars/train.py:171-173`);
a PyBullet world cannot be resumed mid-rollout.  Here the whole simulator state of N
environments is the per-env record (`snk_get_state`): pose, velocities, joint state, last
motor torques, joint-0 force, previous x for the reward -- plus the per-env plane friction and
the model parameters it was produced with.  Restoring it continues bit-for-bit.
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
    mu = getattr(st, "ground_friction", None)
    np.savez_compressed(path, state=state, aux=aux, n_envs=np.int64(st.n_envs),
                        params=np.frombuffer(bytes(st.params), dtype=np.uint8),
                        ground_friction=np.zeros(0, np.float32) if mu is None else np.asarray(mu, np.float32))


def load_state(env, path):
    """Restores a checkpoint into an env created with the same parameters and size."""
    st = _stepper(env)
    with np.load(path if str(path).endswith(".npz") else str(path) + ".npz") as z:
        if int(z["n_envs"]) != st.n_envs:
            raise ValueError("checkpoint holds %d environments, this handle %d" % (int(z["n_envs"]), st.n_envs))
        if z["params"].tobytes() != bytes(st.params):
            raise ValueError("checkpoint was written with different model parameters")
        if z["ground_friction"].size:
            st.set_ground_friction(z["ground_friction"])
        st.set_state(z["state"], z["aux"])
