"""bullet-envs_amd -- MI355X-native batched stepper for the SnakeGymEnv step/reset path.

Import through importlib (the directory name carries the reference's hyphen):
    pkg = importlib.import_module("bullet-envs_amd")
or use the top-level shim `import bullet_envs_amd`.
"""
from ._lib import SnkParams, Stepper, default_params, load, LIB_PATH  # noqa: F401
