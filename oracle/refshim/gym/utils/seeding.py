"""Legacy (gym < 0.22) seeding, restated from its published behaviour:
seed -> (seed mod 2**64) -> sha512(str(seed))[:8] read as little-endian uint32
words -> numpy RandomState.seed([lo, hi, ...]) (MT19937 init_by_array), with
trailing zero words dropped and 0 -> [0].  seed=None -> 8 bytes of os.urandom.
"""
import hashlib
import os
import struct

import numpy as np


def _bigint_from_bytes(b):
    b = b + b"\0" * (4 - len(b) % 4)
    words = struct.unpack("<%dI" % (len(b) // 4), b)
    return sum(w << (32 * i) for i, w in enumerate(words))


def _int_list_from_bigint(v):
    if v == 0:
        return [0]
    out = []
    while v > 0:
        v, m = divmod(v, 2 ** 32)
        out.append(m)
    return out


def create_seed(a=None, max_bytes=8):
    if a is None:
        return _bigint_from_bytes(os.urandom(max_bytes))
    if isinstance(a, (int, np.integer)):
        return int(a) % 2 ** (8 * max_bytes)
    raise TypeError("unsupported seed %r" % (a,))


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(h[:max_bytes])


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError("seed must be a non-negative integer or None")
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
