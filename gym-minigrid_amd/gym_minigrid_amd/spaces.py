"""Minimal stand-ins for gym.spaces (gym is not a dependency): just the attributes the reference's
callers read from `env.action_space` / `env.observation_space` (minigrid.py:792-807)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low, self.high, self.shape, self.dtype)


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype("int64")

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def __repr__(self):
        return "Dict(%r)" % (self.spaces,)
