"""bullet-envs_amd -- MI355X-native batched stepper for the SnakeGymEnv step/reset path.

The directory name carries the reference's hyphen, so import it through importlib
    pkg = importlib.import_module("bullet-envs_amd")
or through the top-level shim:  import bullet_envs_amd as pkg
"""
from ._lib import SnkParams, Stepper, default_params, load, LIB_PATH  # noqa: F401
from .snake_env import (CloudpickleWrapper, Snake, SnakeGymEnv, SnakeVecEnv, SubprocVecEnv, VecEnv,  # noqa: F401
                        params_from_args)
from .device_env import DeviceVecEnv, ShardedVecEnv  # noqa: F401
from .pybullet_client import BulletClient  # noqa: F401  (the reference's inner seam: Snake(pybullet_client, ...))
from . import checkpoint  # noqa: F401  (simulator-state checkpoints)
from .checkpoint import save_state, load_state  # noqa: F401
try:        # the trainer-side pieces need torch; the env itself does not
    from . import rollout  # noqa: F401  (on-device policy inference / rollout buffer)
    from . import ars  # noqa: F401      (ARS normaliser + linear policies on device)
except ImportError:  # pragma: no cover
    rollout = ars = None
