"""Minimal action / observation space containers (what the reference takes from gym.spaces,
multiagent/environment.py:47-68).  gym's classes are used when gym is installed."""
import numpy as np

try:                                     # pragma: no cover - gym is optional
    from gym.spaces import Box, Discrete, Tuple  # noqa: F401
except Exception:
    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.int64

        def contains(self, x):
            return 0 <= int(x) < self.n

        def sample(self):
            return int(np.random.randint(self.n))

        def __repr__(self):
            return "Discrete(%d)" % self.n

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.low = np.full(self.shape, low, dtype=dtype)
            self.high = np.full(self.shape, high, dtype=dtype)
            self.dtype = np.dtype(dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return "Box%s" % (self.shape,)

    class Tuple:
        def __init__(self, spaces):
            self.spaces = tuple(spaces)

        def contains(self, x):
            return len(x) == len(self.spaces) and all(s.contains(v) for s, v in zip(self.spaces, x))

        def sample(self):
            return tuple(s.sample() for s in self.spaces)

        def __repr__(self):
            return "Tuple(%s)" % ", ".join(map(repr, self.spaces))
