import numpy as np


class Space:
    shape = None
    dtype = None


class Box(Space):
    def __init__(self, low, high, shape=None, dtype="float32"):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low)
        self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype("int64")

    def contains(self, x):
        return 0 <= int(x) < self.n

    def sample(self):
        return int(np.random.randint(self.n))


class Dict(Space):
    def __init__(self, spaces=None, **kw):
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)

    def __getitem__(self, key):
        return self.spaces[key]

    def __contains__(self, key):
        return key in self.spaces
