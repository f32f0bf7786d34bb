"""The counter-based synthetic action stream used by bench.py and the parity tests:
a(seed, env, t) = mix64(seed + env*PHI + t*C) mapped to 0..6.  NumPy replica of
`action_of` in csrc/k_state.hip (device) so CPU-side checkers see the same actions."""
import numpy as np

_M = (1 << 64) - 1


def action_stream(seed, envs, ts):
    """envs, ts: broadcastable integer arrays -> uint8 actions."""
    with np.errstate(over="ignore"):
        e = np.asarray(envs, dtype=np.uint64)
        t = np.asarray(ts, dtype=np.uint64)
        z = np.uint64(seed & _M) + e * np.uint64(0x9E3779B97F4A7C15) + t * np.uint64(0xD1B54A32D192ED03)
        z = z ^ (z >> np.uint64(30))
        z = z * np.uint64(0xBF58476D1CE4E5B9)
        z = z ^ (z >> np.uint64(27))
        z = z * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return (((z >> np.uint64(32)) * np.uint64(7)) >> np.uint64(32)).astype(np.uint8)
