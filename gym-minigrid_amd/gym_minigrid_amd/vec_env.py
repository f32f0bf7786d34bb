"""VecMiniGrid -- the batched, MI355X-resident counterpart of the reference's gym.Env surface.

Mirrors, for N lockstep env instances, the methods and attributes callers use on
`gym_minigrid.minigrid.MiniGridEnv` (/root/reference/gym_minigrid/minigrid.py):
    reset() -> obs                      :831-858   (reseed=True: ReseedWrapper(seeds=[s_i]) -- seed(s_i) then reset(), wrappers.py:24-28;
                                                    reseed=False: the plain reset(), the env's RNG stream continues;
                                                    set_seed_schedule(): ReseedWrapper with a list of seeds per env, wrappers.py:12-28)
    seed(seed)                          :860-863
    step(actions) -> obs, reward, done, info   :1227-1325
    action_space = Discrete(7)          :792
    observation_space['image']          :799-807   (FullyObsWrapper: (W, H, 3), wrappers.py:316-324)
    max_steps, width, height, mission
with observations as the `image` array only (ImgObsWrapper semantics, wrappers.py:156-166), batched to
uint8 (N,7,7,3) / (N,W,H,3).  All simulation runs in libmgx.so's HIP kernels; this file is plumbing.
"""
import ctypes

import numpy as np

from . import _lib
from .spaces import Box, Dict, Discrete


def _ptr(a):
    """numpy array / torch tensor / None -> c_void_p"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return ctypes.c_void_p(a.ctypes.data)
    assert a.is_contiguous()
    return ctypes.c_void_p(a.data_ptr())


class VecMiniGrid:
    """N independent MiniGrid envs of one registered id (or an explicit config), stepped on one GPU.

    obs_mode: 'partial' (V,V,3) | 'full' (W,H,3, FullyObsWrapper) | 'partial_onehot' (V,V,21, OneHotPartialObsWrapper
              formula) | 'full_onehot' (W,H,22, FullyObsOneHotWrapper(flatten=False)) | 'full_onehot_nocolor' (W,H,15).
    backend='torch': outputs are torch tensors on cuda:<device> (zero-copy device pointers, asynchronous on
                     torch's current stream).  backend='numpy': host arrays (staged copies, synchronous).
    auto_reset=True: an env that reports done is restored to its episode start inside the same step and `obs`
                     is the first observation of the new episode (VecEnv convention; ReseedWrapper(seeds=[s_i])
                     layout semantics).  auto_reset=False: exact reference semantics, the caller resets.
    new_level_each_episode=True (with auto_reset): plain reference behaviour -- `seed(s_i)` once at reset(), then every
                     episode draws a NEW level from the env's own RNG stream (generated on the GPU); False: every
                     episode replays the level of reset() (ReseedWrapper(seeds=[s_i])).
    object_state=True: keep the hidden Goal/Box state (toggletimes, triage_color, Box.contains; minigrid.py:156-181,332-364)
                     per cell; injected with set_state(aux=...) + set_object_state(...).  Default: what the env id needs
                     (on for the ObstructedMaze levels that hide keys in boxes, off elsewhere).
    default_vis=False: the fork's alternative occlusion model (minigrid.py:649-709).
    extended_actions: ExtendedActions (minigrid.py:747-764): actions 7 / 8 strafe left / right.
    agent_view_size: ViewSizeWrapper (wrappers.py:579-608): 3, 5, 7 (default), 9 or 11.
    seeds: int (env i gets seed+i+env_offset) or an array of N uint64 seeds.
    """

    def __init__(self, env_id=None, num_envs=1, device=0, seeds=0, obs_mode="partial", auto_reset=True,
                 config=None, backend="torch", env_offset=0, check_actions=False, new_level_each_episode=False,
                 agent_view_size=7, extended_actions=False, default_vis=True, object_state=None):
        L = _lib.lib()
        if config is None:
            if env_id is None:
                raise ValueError("give env_id or config")
            config = _lib.env_config(env_id)
        self.env_id = env_id
        cfg = _lib.Config()
        ctypes.memmove(ctypes.byref(cfg), ctypes.byref(config), ctypes.sizeof(cfg))
        cfg.obs_mode = {"partial": _lib.OBS_PARTIAL, "full": _lib.OBS_FULL, "partial_onehot": _lib.OBS_PARTIAL_ONEHOT,
                        "full_onehot": _lib.OBS_FULL_ONEHOT, "full_onehot_nocolor": _lib.OBS_FULL_ONEHOT_NOCOLOR,
                        "flat": _lib.OBS_PARTIAL_FLAT, "full_flat": _lib.OBS_FULL_FLAT}[obs_mode]
        cfg.auto_reset = int(bool(auto_reset))
        cfg.new_level_each_episode = int(bool(new_level_each_episode))
        cfg.agent_view_size = int(agent_view_size)
        cfg.extended_actions = int(bool(extended_actions))
        cfg.alt_visibility = int(not default_vis)
        if object_state is not None:  # None: the id's own need (ObstructedMaze's boxed keys switch it on)
            cfg.object_state = int(bool(object_state))
        self.cfg = cfg
        self.num_envs = int(num_envs)
        self.device = int(device)
        self.backend = backend
        self.env_offset = int(env_offset)
        self.check_actions = bool(check_actions)
        self.width, self.height, self.max_steps = cfg.width, cfg.height, cfg.max_steps
        self.obs_mode = obs_mode
        self.agent_view_size = int(agent_view_size)
        chan = {"partial": 3, "full": 3, "partial_onehot": 21, "full_onehot": 22, "full_onehot_nocolor": 15, "flat": 3, "full_flat": 3}[obs_mode]
        self.obs_shape = ((self.agent_view_size,) * 2 if obs_mode in ("partial", "partial_onehot", "flat") else (cfg.width, cfg.height)) + (chan,)
        self.obs_dtype = "uint8"
        if obs_mode in ("flat", "full_flat"):  # FlatObsWrapper (wrappers.py:528-577): image ++ one-hot mission, float32
            self.obs_shape = (int(np.prod(self.obs_shape)) + 27 * 96,)
            self.obs_dtype = "float32"
        self.n_actions = 9 if extended_actions else 7
        self.action_space = Discrete(self.n_actions)  # minigrid.py:788-792
        if cfg.task_kind == _lib.TASK_DYNOBS:  # Dynamic-Obstacles: Discrete(3), larger actions fold to 0 (envs/dynamicobstacles.py:32-33,61-63)
            self.action_space = Discrete(3)
            self.n_actions = 256
        self.observation_space = Dict({"image": Box(0, 255, self.obs_shape, "uint8")})
        if self.obs_dtype == "float32":
            self.observation_space = Box(0, 255, (1,) + self.obs_shape, "uint8")  # wrappers.py:543-548
        self.reward_range = (-1, 1) if cfg.task_kind == _lib.TASK_DYNOBS else (0, 1)
        try:  # families whose mission names per-episode objects (Fetch, GoToObject, UnlockPickup, KeyCorridor, LockedRoom)
            self.mission = self._mission_of(0) if cfg.task_kind not in _lib.TASKS_WITH_EPISODE_MISSION else "per episode: see missions()"
        except _lib.MgxError:
            self.mission = "per episode: see missions()"
        self._h = ctypes.c_void_p()
        _lib.check(L.mgx_create(ctypes.byref(cfg), self.num_envs, self.device, ctypes.byref(self._h)))
        self._torch = None
        if backend == "torch":
            import torch
            self._torch = torch
            self._dev = torch.device("cuda", self.device)
            self._bind_stream()
        elif backend != "numpy":
            raise ValueError("backend must be 'torch' or 'numpy'")
        self._obs = self._new((self.num_envs,) + self.obs_shape, self.obs_dtype)
        self._reward = self._new((self.num_envs,), "float32")
        self._done = self._new((self.num_envs,), "uint8")
        # step()'s output pointers never change (the buffers belong to this object): looked up once, passed as plain integers.
        # Below ~200 k envs a step is bound by the host side of the call, of which these lookups were a fifth.
        self._out_ptrs = tuple(_ptr(x).value for x in (self._obs, self._reward, self._done))
        self._step_fn = L.mgx_step
        self._act_shape = (self.num_envs,)
        self.seed(seeds)

    # ------------------------------------------------------------------ plumbing
    def _new(self, shape, dtype):
        if self._torch is not None:
            return self._torch.empty(shape, dtype=getattr(self._torch, dtype), device=self._dev)
        return np.empty(shape, dtype=dtype)

    def _bind_stream(self):
        s = self._torch.cuda.current_stream(self._dev).cuda_stream
        if getattr(self, "_stream", -1) != s:
            _lib.check(_lib.lib().mgx_set_stream(self._h, ctypes.c_void_p(s)))
            self._stream = s

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().mgx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _actions(self, actions):
        if self._torch is not None:
            t = self._torch
            if not isinstance(actions, t.Tensor):
                actions = t.as_tensor(np.asarray(actions), device=self._dev)
            if self.check_actions and bool((actions >= self.n_actions).any() | (actions < 0).any()):
                raise AssertionError("unknown action")  # minigrid.py:1318
            if actions.dtype != t.uint8 or actions.device != self._dev:
                actions = actions.to(device=self._dev, dtype=t.uint8)
            actions = actions.contiguous()
        else:
            actions = np.asarray(actions)
            if self.check_actions and ((actions >= self.n_actions).any() or (actions < 0).any()):
                raise AssertionError("unknown action")
            actions = np.ascontiguousarray(actions, dtype=np.uint8)
        if tuple(actions.shape) != (self.num_envs,):
            raise ValueError("actions must have shape (%d,)" % self.num_envs)
        return actions

    # ------------------------------------------------------------------ gym.Env surface
    def seed(self, seed=1337):
        """env i will be reset with seed_i: an int gives seed + env_offset + i, an array gives seeds[i]."""
        if np.ndim(seed) == 0:
            s = (np.uint64(int(seed) % (1 << 64)) + np.arange(self.env_offset, self.env_offset + self.num_envs, dtype=np.uint64))
        else:
            s = np.ascontiguousarray(seed, dtype=np.uint64)
            if s.shape != (self.num_envs,):
                raise ValueError("seeds must have shape (%d,)" % self.num_envs)
        self.seeds = s
        self._seeds_dev = None  # device copy (torch backend), made on first use
        return [seed]

    def set_seed_schedule(self, seeds, seed_idx=0):
        """ReseedWrapper(env_i, seeds=seeds[i], seed_idx=seed_idx) for every env (wrappers.py:12-28): seeds (N, K) uint64.  From now on every
        reset of env i -- the in-kernel one of auto_reset=True and reset(reseed=False) -- re-seeds it with the next entry of its list,
        cyclically.  Like the wrapper's constructor this does not reset: call reset(reseed=False) next (step() refuses until then).
        seeds=None removes the schedule."""
        if self._torch is not None:
            self._bind_stream()
        if seeds is None:
            _lib.check(_lib.lib().mgx_set_seed_schedule(self._h, None, 0, 0))
            return
        s = np.ascontiguousarray(seeds, dtype=np.uint64)
        if s.ndim != 2 or s.shape[0] != self.num_envs:
            raise ValueError("seeds must have shape (%d, K)" % self.num_envs)
        _lib.check(_lib.lib().mgx_set_seed_schedule(self._h, _ptr(s), int(s.shape[1]), int(seed_idx)))

    BONUS_KINDS = {"action": 1, "state": 2}   # MGX_BONUS_ACTION / MGX_BONUS_STATE (include/mgx.h)

    def add_bonus(self, kind):
        """env = ActionBonus(env) ("action") / StateBonus(env) ("state") for every env (wrappers.py:87-153): each step adds
        1 / sqrt(visits of this env to (agent_pos, agent_dir, action) / to agent_pos) to the reward; the counts persist across episodes.
        Calls stack in call order, innermost first, one of each at most; kind=None removes both.  Every call zeroes the counts."""
        if self._torch is not None:
            self._bind_stream()
        _lib.check(_lib.lib().mgx_add_bonus(self._h, 0 if kind is None else self.BONUS_KINDS[kind]))

    def set_dac(self, on=True):
        """env = DACWrapper(env) for every env (wrappers.py:35-84, the fork's absorbing-state wrapper): an env that is done before its
        max_steps-th step is not reset; until that step its image is all ones, its reward 0, its actions ignored; that step reports done."""
        if self._torch is not None:
            self._bind_stream()
        _lib.check(_lib.lib().mgx_set_dac(self._h, 1 if on else 0))

    def bonus_counts(self, kind):
        """The wrapper's self.counts as a dense uint32 array: (N, W, H, 4, A) for "action" (A = 7, or 9 with extended actions), (N, W, H) for "state"."""
        A = 9 if self.cfg.extended_actions else 7
        shape = (self.num_envs, self.width, self.height) + ((4, A) if kind == "action" else ())
        out = np.zeros(shape, np.uint32)
        if self._torch is not None:
            self._bind_stream()
        _lib.check(_lib.lib().mgx_get_bonus_counts(self._h, self.BONUS_KINDS[kind], _ptr(out)))
        return out

    def reset(self, mask=None, reseed=True):
        """reseed=True: seed(seed_i); reset() for every env (or those with mask[i] != 0) -- ReseedWrapper(seeds=[seed_i]) semantics.
        reseed=False: the reference's plain reset() (minigrid.py:831-858; `if done: env.reset()`, run_tests.py:64-66): the env's own RNG
        stream continues and a NEW level is drawn from it -- or, after set_seed_schedule(), ReseedWrapper.reset(): the next seed of the
        env's list.  Returns the obs buffer of ALL envs (the one step() returns: with a mask only the tiles holding a reset env are
        rewritten, the rest still hold the last step)."""
        seeds = self.seeds if reseed else None
        if self._torch is not None:
            self._bind_stream()
            if reseed and self._seeds_dev is None:  # keep the seeds on the GPU: a masked reset per step must not upload 8 B per env
                self._seeds_dev = self._torch.from_numpy(self.seeds.view(np.int64)).to(self._dev)
            seeds = self._seeds_dev if reseed else None
        if mask is None or isinstance(mask, np.ndarray) or self._torch is None or not isinstance(mask, self._torch.Tensor):
            m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        else:                            # e.g. the `done` tensor step() returned: used in place
            m = mask if mask.dtype == self._torch.uint8 else mask.to(self._torch.uint8)
            if m.shape != (self.num_envs,) or not m.is_contiguous():
                raise ValueError("mask must be a contiguous tensor of shape (num_envs,)")
        _lib.check(_lib.lib().mgx_reset(self._h, _ptr(seeds), _ptr(m), _ptr(self._obs)))
        return self._obs

    def step(self, actions):
        t = self._torch
        if t is not None:
            self._bind_stream()
            # the common case -- a contiguous uint8 tensor of the right shape on the handle's device -- skips the conversions
            if (type(actions) is t.Tensor and actions.dtype is t.uint8 and actions.device == self._dev and tuple(actions.shape) == self._act_shape
                    and actions.is_contiguous() and not self.check_actions):
                a = actions
            else:
                a = self._actions(actions)
            ap = a.data_ptr()
        else:
            a = self._actions(actions)
            ap = a.ctypes.data
        rc = self._step_fn(self._h, ap, *self._out_ptrs)
        if rc:
            _lib.check(rc)
        self._last_actions = a  # keep alive until the async kernel has consumed it
        return self._obs, self._reward, self._done, {}

    def rollout(self, actions, with_obs=True):
        """T steps in one call: actions uint8 (T, N) on the device -> obs (T, N, ...), reward (T, N), done (T, N).
        One hipGraph launch per call once (T, buffers) repeat: the output buffers are cached per T for that reason and
        are overwritten by the next rollout of the same length (torch backend only)."""
        if self._torch is None:
            raise ValueError("rollout needs the torch backend (device buffers)")
        torch = self._torch
        self._bind_stream()
        T = int(actions.shape[0])
        if actions.dtype != torch.uint8 or tuple(actions.shape) != (T, self.num_envs) or not actions.is_contiguous() or not actions.is_cuda:
            raise ValueError("actions must be a contiguous uint8 cuda tensor of shape (T, num_envs)")
        cache = self.__dict__.setdefault("_roll", {})
        if T not in cache:
            cache[T] = (torch.empty((T, self.num_envs) + self.obs_shape, dtype=getattr(torch, self.obs_dtype), device=self._dev),
                        torch.empty((T, self.num_envs), dtype=torch.float32, device=self._dev),
                        torch.empty((T, self.num_envs), dtype=torch.uint8, device=self._dev))
        obs, reward, done = cache[T]
        _lib.check(_lib.lib().mgx_rollout(self._h, T, _ptr(actions), _ptr(obs) if with_obs else None, _ptr(reward), _ptr(done)))
        return (obs if with_obs else None), reward, done

    def observe(self):
        if self._torch is not None:
            self._bind_stream()
        _lib.check(_lib.lib().mgx_observe(self._h, _ptr(self._obs)))
        return self._obs

    def direction(self):
        """obs['direction'] of the reference (minigrid.py:1375-1379): agent_dir per env, uint8 (N,)."""
        if getattr(self, "_dir", None) is None:
            self._dir = self._new((self.num_envs,), "uint8")
        if self._torch is not None:
            self._bind_stream()
        _lib.check(_lib.lib().mgx_get_direction(self._h, _ptr(self._dir)))
        return self._dir

    def pose(self):
        """env.agent_pos / agent_dir per env, int32 (N, 3) = (x, y, dir): the 'pos' and 'dir' AgentExtraInfoWrapper adds to the
        observation (wrappers.py:169-187).  No device sync with the torch backend."""
        if getattr(self, "_pose", None) is None:
            self._pose = self._new((self.num_envs, 3), "int32")
        if self._torch is not None:
            self._bind_stream()
        _lib.check(_lib.lib().mgx_get_pose(self._h, _ptr(self._pose)))
        return self._pose

    # ------------------------------------------------------------------ state injection / inspection
    def set_state(self, grid, agent, aux=None, carry=None, steps=None):
        """Reference-encoded state (host numpy arrays): grid (N,W,H,3) u8, agent (N,3) i32, aux (N,W,H) u8,
        carry (N,3) u8, steps (N,) i32.  Also becomes the episode start used by auto-reset -- except on a seeded
        new_level_each_episode handle, where it replaces the current episode only and the next one is the env's next level."""
        n, W, H = self.num_envs, self.width, self.height
        g = np.ascontiguousarray(grid, np.uint8)
        assert g.shape == (n, W, H, 3), g.shape
        ag = np.ascontiguousarray(agent, np.int32)
        assert ag.shape == (n, 3)
        ax = None if aux is None else np.ascontiguousarray(aux, np.uint8)
        ca = None if carry is None else np.ascontiguousarray(carry, np.uint8)
        st = None if steps is None else np.ascontiguousarray(steps, np.int32)
        if self._torch is not None:
            self._bind_stream()
        _lib.check(_lib.lib().mgx_set_state(self._h, _ptr(g), _ptr(ax), _ptr(ag), _ptr(ca), _ptr(st)))

    def get_state(self):
        n, W, H = self.num_envs, self.width, self.height
        out = dict(grid=np.empty((n, W, H, 3), np.uint8), aux=np.empty((n, W, H), np.uint8),
                   agent=np.empty((n, 3), np.int32), carry=np.empty((n, 3), np.uint8), steps=np.empty(n, np.int32))
        if self._torch is not None:
            self._bind_stream()
        _lib.check(_lib.lib().mgx_get_state(self._h, _ptr(out["grid"]), _ptr(out["aux"]), _ptr(out["agent"]),
                                            _ptr(out["carry"]), _ptr(out["steps"])))
        return out

    def set_object_state(self, contains=None, carry_aux=None, carry_contains=None):
        """object_state handles: Box.contains plane (N,W,H,3) and the carried object's hidden state (aux (N,), contains (N,3))."""
        c = None if contains is None else np.ascontiguousarray(contains, np.uint8)
        a = None if carry_aux is None else np.ascontiguousarray(carry_aux, np.uint8)
        cc = None if carry_contains is None else np.ascontiguousarray(carry_contains, np.uint8)
        _lib.check(_lib.lib().mgx_set_object_state(self._h, _ptr(c), _ptr(a), _ptr(cc)))

    def get_object_state(self):
        n, W, H = self.num_envs, self.width, self.height
        out = dict(contains=np.empty((n, W, H, 3), np.uint8), carry_aux=np.empty(n, np.uint8), carry_contains=np.empty((n, 3), np.uint8))
        _lib.check(_lib.lib().mgx_get_object_state(self._h, _ptr(out["contains"]), _ptr(out["carry_aux"]), _ptr(out["carry_contains"])))
        return out

    def _mission_of(self, task):
        buf = ctypes.create_string_buffer(128)
        n = _lib.lib().mgx_mission(ctypes.byref(self.cfg), int(task), buf, 128)
        if n < 0:
            _lib.check(n)
        return buf.value.decode()

    def missions(self):
        """obs['mission'] of every env (minigrid.py:1373-1379); only Fetch missions differ between envs."""
        if self.cfg.task_kind == _lib.TASK_NONE:
            return [self.mission] * self.num_envs
        cache = {}
        return [cache.setdefault(int(t), self._mission_of(int(t))) for t in self.get_task()]

    def set_task(self, task):
        """Per-env task word (Fetch: target cell code = type | color << 4), host uint32 array (N,)."""
        t = np.ascontiguousarray(task, np.uint32)
        assert t.shape == (self.num_envs,)
        _lib.check(_lib.lib().mgx_set_task(self._h, _ptr(t)))

    def get_task(self):
        t = np.empty(self.num_envs, np.uint32)
        _lib.check(_lib.lib().mgx_get_task(self._h, _ptr(t)))
        return t

    def sync(self):
        """Wait for all enqueued work; raises AssertionError subclasses for recorded faults (like the reference)."""
        _lib.check(_lib.lib().mgx_sync(self._h))

    def clear_faults(self):
        _lib.check(_lib.lib().mgx_clear_faults(self._h))

    def stats(self):
        s = _lib.Stats()
        _lib.check(_lib.lib().mgx_get_stats(self._h, ctypes.byref(s)))
        return dict(steps=s.steps, episodes=s.episodes, reward_sum=s.reward_sum,
                    invalid_actions=s.invalid_actions, out_of_bounds=s.out_of_bounds)

    def read_stats_async(self, out2):
        """Enqueue (episodes, reward_sum) -> out2 (float64 cuda tensor of 2 elements); no host sync."""
        self._bind_stream()
        _lib.check(_lib.lib().mgx_read_stats_async(self._h, _ptr(out2)))
        return out2

    def fill_actions(self, seed, t0, T, out=None):
        """actions[T][N] of the synthetic counter-based stream (actions.action_stream), generated on the GPU."""
        if out is None:
            out = self._new((T, self.num_envs), "uint8")
        if self._torch is not None:
            self._bind_stream()
        _lib.check(_lib.lib().mgx_fill_actions(self._h, ctypes.c_uint64(seed), self.env_offset, int(t0), int(T), _ptr(out)))
        return out

    def step_kernel_name(self):
        """The step-kernel instantiation this handle launches, e.g. 'k_step<8,8,0,7>' (include/mgx.h)."""
        buf = ctypes.create_string_buffer(64)
        n = _lib.lib().mgx_step_kernel_name(self._h, buf, 64)
        if n < 0:
            _lib.check(n)
        return buf.value.decode()

    def profile_begin(self, stride=8):
        """Start timing (include/mgx.h): the stream span + every `stride`-th step-kernel launch on its own."""
        _lib.check(_lib.lib().mgx_profile_begin_sampled(self._h, int(stride)))

    def profile_stop(self):
        """Enqueue the span's end marker now, without waiting (profile_end() after the caller's own synchronise reads the times)."""
        _lib.check(_lib.lib().mgx_profile_stop(self._h))

    def profile_end(self):
        """-> (step-kernel launches in the span, span ms).  profile_kernel() then gives the per-launch samples."""
        n, ms = ctypes.c_int64(), ctypes.c_double()
        _lib.check(_lib.lib().mgx_profile_end(self._h, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def profile_kernel(self):
        """-> (sampled launches, summed duration ms) of the step kernel alone over the last profiled span."""
        n, ms = ctypes.c_int64(), ctypes.c_double()
        _lib.check(_lib.lib().mgx_profile_kernel(self._h, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value


def generate_levels(env_id_or_cfg, seeds, with_task=False, with_contains=False):
    """Host-side `env.seed(s); env.reset()` of a built-in family -> (grid (n,W,H,3) u8, agent (n,3) i32[, task (n,) u32]
    [, contains (n,W,H,3) u8: Box.contains of every cell, (1,0,0) = nothing])."""
    cfg = _lib.env_config(env_id_or_cfg) if isinstance(env_id_or_cfg, str) else env_id_or_cfg
    s = np.ascontiguousarray(seeds, dtype=np.uint64)
    n = s.shape[0]
    grid = np.empty((n, cfg.width, cfg.height, 3), np.uint8)
    agent = np.empty((n, 3), np.int32)
    task = np.zeros(n, np.uint32)
    contains = np.empty((n, cfg.width, cfg.height, 3), np.uint8) if with_contains else None
    _lib.check(_lib.lib().mgx_generate_levels_full(ctypes.byref(cfg), n, _ptr(s), _ptr(grid), _ptr(agent), _ptr(task), _ptr(contains)))
    out = (grid, agent) + ((task,) if with_task else ()) + ((contains,) if with_contains else ())
    return out


def generate_level_stream(env_id_or_cfg, seed, K, with_task=False, with_contains=False):
    """Host-side `env.seed(seed)` then K consecutive `env.reset()`s (the RNG stream continues across episodes)."""
    cfg = _lib.env_config(env_id_or_cfg) if isinstance(env_id_or_cfg, str) else env_id_or_cfg
    grid = np.empty((K, cfg.width, cfg.height, 3), np.uint8)
    agent = np.empty((K, 3), np.int32)
    task = np.zeros(K, np.uint32)
    contains = np.empty((K, cfg.width, cfg.height, 3), np.uint8) if with_contains else None
    _lib.check(_lib.lib().mgx_generate_level_stream_full(ctypes.byref(cfg), ctypes.c_uint64(int(seed)), K, _ptr(grid), _ptr(agent), _ptr(task), _ptr(contains)))
    return (grid, agent) + ((task,) if with_task else ()) + ((contains,) if with_contains else ())
