"""Minimal stand-in for gym.spaces.Box (gym is not a dependency of the stepper).

The reference builds `gym.spaces.Box(low, high)` for observation and action spaces
(/root/reference/SnakeGymEnv.py:60-79); callers only read `.shape`, `.low`, `.high`
(ppo/train.py:72-73, ars/train.py:195-196).  If gym is importable its Box is used instead.
"""
import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box as _GymBox
except Exception:  # noqa: BLE001
    _GymBox = None


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        assert self.low.shape == self.high.shape
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return "Box%s" % (self.shape,)


def make_box(low, high):
    if _GymBox is not None:
        return _GymBox(np.asarray(low, dtype=np.float32), np.asarray(high, dtype=np.float32))
    return Box(low, high)
