"""Minimal `gym.spaces.Box` stand-in.

The reference builds `gym.spaces.Box` objects (engine.py:296,393-413) and the
learners test `isinstance(action_space, gym.spaces.Box)`
(safe_rl_libX/trpo/trpo_core.py:150-164).  When `gym` is importable the real
class is used so that check keeps working; otherwise this stand-in carries the
attributes the learners read (`shape`, `low`, `high`, `dtype`).
"""
import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box as Box  # type: ignore
    HAVE_GYM = True
except Exception:  # noqa: BLE001
    HAVE_GYM = False

    class Box:  # type: ignore[no-redef]
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            if shape is None:
                low = np.asarray(low, dtype=self.dtype)
                high = np.asarray(high, dtype=self.dtype)
                shape = low.shape
            else:
                shape = tuple(int(s) for s in shape)
                low = np.full(shape, low, dtype=self.dtype)
                high = np.full(shape, high, dtype=self.dtype)
            assert low.shape == high.shape == tuple(shape)
            self.low, self.high, self.shape = low, high, tuple(shape)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return np.random.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

        def __eq__(self, other):
            return (isinstance(other, Box) and self.shape == other.shape
                    and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))
