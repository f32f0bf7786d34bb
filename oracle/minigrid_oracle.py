"""ctypes front-end of the CPU oracle (oracle/minigrid_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.  State lives in plain
numpy arrays in the reference's own encoding (see the C file's header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmgoracle.so")


class _Cfg(ctypes.Structure):
    _fields_ = [("W", ctypes.c_int), ("H", ctypes.c_int), ("max_steps", ctypes.c_int),
                ("see_through", ctypes.c_int), ("lava_v1", ctypes.c_int), ("view", ctypes.c_int), ("extended", ctypes.c_int), ("alt_vis", ctypes.c_int), ("task", ctypes.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "minigrid_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/libmgoracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.mgo_step_batch.restype = ctypes.c_int
        _lib.mgo_rollout.restype = ctypes.c_int64
        _lib.mgo_obs_batch.restype = None
        _lib.mgo_set_task.restype = None
        _lib.mgo_set_contains.restype = None
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class OracleEnvs:
    """N independent reference-semantics envs of one family, stepped on the CPU."""

    def __init__(self, W, H, max_steps, see_through, lava_v1=False, view=7, extended=False, alt_vis=False, task=0):
        self.W, self.H, self.V = int(W), int(H), int(view)
        assert 1 <= self.V <= 15
        self.cfg = _Cfg(self.W, self.H, int(max_steps), int(bool(see_through)), int(bool(lava_v1)), self.V, int(bool(extended)), int(bool(alt_vis)), int(task))
        self.task = None
        self.n = 0

    def set_state(self, grid, agent, aux=None, carry=None, steps=None, carry_aux=None):
        grid = np.ascontiguousarray(grid, np.uint8)
        n = grid.shape[0]
        assert grid.shape == (n, self.W, self.H, 3), grid.shape
        self.n = n
        self.grid = grid.copy()
        self.aux = np.zeros((n, self.W, self.H), np.uint8) if aux is None else np.ascontiguousarray(aux, np.uint8).copy()
        self.agent = np.ascontiguousarray(agent, np.int32).reshape(n, 3).copy()
        if carry is None:
            carry = np.tile(np.array([1, 0, 0], np.uint8), (n, 1))
        self.carry = np.ascontiguousarray(carry, np.uint8).reshape(n, 3).copy()
        self.carry_aux = np.zeros(n, np.uint8) if carry_aux is None else np.ascontiguousarray(carry_aux, np.uint8).copy()
        self.steps = np.zeros(n, np.int32) if steps is None else np.ascontiguousarray(steps, np.int32).copy()
        self.grid0, self.aux0, self.agent0 = self.grid.copy(), self.aux.copy(), self.agent.copy()
        self.contains = None          # optional Box.contains plane (n, W, H, 3); set_contains() enables it
        self.carry_contains = None

    def set_contains(self, contains, carry_contains=None):
        self.contains = np.ascontiguousarray(contains, np.uint8).copy()
        assert self.contains.shape == (self.n, self.W, self.H, 3)
        if carry_contains is None:
            carry_contains = np.tile(np.array([1, 0, 0], np.uint8), (self.n, 1))
        self.carry_contains = np.ascontiguousarray(carry_contains, np.uint8).reshape(self.n, 3).copy()
        self.contains0 = self.contains.copy()

    def observe(self, full=False):
        obs = np.zeros((self.n, self.V, self.V, 3), np.uint8)
        fo = np.zeros((self.n, self.W, self.H, 3), np.uint8) if full else None
        lib().mgo_obs_batch(ctypes.byref(self.cfg), ctypes.c_int64(self.n), _p(self.grid), _p(self.aux),
                            _p(self.agent), _p(self.carry), _p(obs), _p(fo))
        return (obs, fo) if full else obs

    def step(self, actions, full=False):
        a = np.ascontiguousarray(actions, np.uint8)
        assert a.shape == (self.n,)
        obs = np.zeros((self.n, self.V, self.V, 3), np.uint8)
        fo = np.zeros((self.n, self.W, self.H, 3), np.uint8) if full else None
        reward = np.zeros(self.n, np.float64)
        done = np.zeros(self.n, np.uint8)
        err = np.zeros(self.n, np.int32)
        lib().mgo_set_contains(_p(self.contains), _p(self.carry_contains))
        if self.cfg.task:
            self._task = np.ascontiguousarray(self.task if self.task is not None else np.zeros(self.n), np.uint32).copy()
            lib().mgo_set_task(_p(self._task))
        lib().mgo_step_batch(ctypes.byref(self.cfg), ctypes.c_int64(self.n), _p(self.grid), _p(self.aux),
                             _p(self.agent), _p(self.carry), _p(self.carry_aux), _p(self.steps), _p(a),
                             _p(obs), _p(fo), _p(reward), _p(done), _p(err))
        self.err = err
        if self.cfg.task == 11:      # TwoGoals: the task word is the running goal count, updated by the step
            self.task = self._task
        if full:
            return obs, fo, reward, done
        return obs, reward, done

    def reset_where(self, mask):
        """Caller-side reset on done with the SAME seed (ReseedWrapper(seeds=[s])): restore episode start."""
        m = np.asarray(mask, bool)
        self.grid[m] = self.grid0[m]
        self.aux[m] = self.aux0[m]
        self.agent[m] = self.agent0[m]
        self.carry[m] = (1, 0, 0)
        self.carry_aux[m] = 0
        self.steps[m] = 0
        if self.cfg.task == 11 and self.task is not None:
            self.task[m] = 0
        if self.contains is not None:
            self.contains[m] = self.contains0[m]
            self.carry_contains[m] = (1, 0, 0)

    def rollout(self, actions, with_obs=True, full=False):
        """actions u8[T][n]; restores the initial state on done.  Returns env-steps executed."""
        a = np.ascontiguousarray(actions, np.uint8)
        T = a.shape[0]
        assert a.shape == (T, self.n)
        obs = np.zeros((self.n, self.V, self.V, 3), np.uint8) if with_obs else None
        fo = np.zeros((self.n, self.W, self.H, 3), np.uint8) if full else None
        reward = np.zeros(self.n, np.float64)
        done = np.zeros(self.n, np.uint8)
        r = lib().mgo_rollout(ctypes.byref(self.cfg), ctypes.c_int64(self.n), ctypes.c_int64(T), _p(self.grid),
                              _p(self.aux), _p(self.agent), _p(self.carry), _p(self.carry_aux), _p(self.steps),
                              _p(self.grid0), _p(self.aux0), _p(self.agent0), _p(a), _p(obs), _p(fo),
                              _p(reward), _p(done))
        self.last = (obs, fo, reward, done)
        return int(r)


def flat_obs(image, mission, max_str_len=96, num_char_codes=27):
    """FlatObsWrapper.observation (wrappers.py:556-577) restated: image bytes followed by the one-hot mission string,
    float32 (np.concatenate of uint8 and float32).  A character outside a-z/space re-uses the previous character's
    code, as the reference's un-reset `chNo` does."""
    assert len(mission) <= max_str_len
    arr = np.zeros((max_str_len, num_char_codes), np.float32)
    code = None
    for i, ch in enumerate(mission.lower()):
        if "a" <= ch <= "z":
            code = ord(ch) - ord("a")
        elif ch == " ":
            code = 26
        arr[i, code] = 1
    return np.concatenate((np.asarray(image, np.uint8).flatten(), arr.flatten()))
