"""new_level_each_episode with the level generator running BESIDE the steps (a ring of four next-level buffers per env, k_levelgen on a stream
of its own every second step) against the form with one buffer and k_levelgen behind every step (MGX_LG_RING=off), which
tests/test_gpu_stream.py pins to the reference's traces and to host generator + oracle (and which the default form passes there too).  Both forms run side by side here: every observation, reward, done and state equal,
through caller-side resets of all three kinds, injected states and captured rollouts of odd and even length."""
import numpy as np
import pytest

import gym_minigrid_amd as mg
from helpers import to_np

pytestmark = pytest.mark.gpu

FAMILIES = ["MiniGrid-LavaCrossingS9N1-v0", "MiniGrid-DoorKey-5x5-v0", "MiniGrid-LavaGapS7-v1", "MiniGrid-Empty-Random-6x6-v0",
            "MiniGrid-SimpleCrossingS11N5-v0", "MiniGrid-DoorKey-8x8-v0", "MiniGrid-LavaCrossingS9N3-v0",
            # task words, the second MT19937 block, hidden object state (three planes per buffer), the gather form
            "MiniGrid-Fetch-8x8-N3-v0", "MiniGrid-Unlock-v0", "MiniGrid-GoToDoor-5x5-v0", "MiniGrid-KeyCorridorS3R2-v0",
            "MiniGrid-ObstructedMaze-1Dlhb-v0", "MiniGrid-MemoryS13Random-v0", "MiniGrid-RedBlueDoors-6x6-v0"]
RAW_ACTIONS = ("DoorKey", "Fetch", "Unlock", "GoToDoor", "KeyCorridor", "ObstructedMaze", "RedBlueDoors")


def pair(env_id, N, monkeypatch, auto_reset=True, seed=5):
    envs = []
    for f in ("16", "off"):
        monkeypatch.setenv("MGX_LG_RING", f)
        envs.append(mg.VecMiniGrid(env_id, num_envs=N, seeds=seed, auto_reset=auto_reset, new_level_each_episode=True, backend="torch"))
    monkeypatch.delenv("MGX_LG_RING")
    return envs


def same_state(a, b):
    sa, sb = a.get_state(), b.get_state()
    ok = all(np.array_equal(sa[k], sb[k]) for k in ("grid", "agent", "steps", "carry"))
    if a.cfg.task_kind:
        ok = ok and np.array_equal(a.get_task(), b.get_task())
    if a.cfg.object_state:
        oa, ob = a.get_object_state(), b.get_object_state()
        ok = ok and all(np.array_equal(oa[k], ob[k]) for k in oa)
    return ok


@pytest.mark.parametrize("merge", ["0", "1"])
@pytest.mark.parametrize("env_id", FAMILIES)
def test_ring_generator_equals_split(env_id, merge, monkeypatch):
    """merge: a generator launch per step's flag array, or one per run of four steps (k_levelgen<true>: an env that finished twice within the
    run draws its levels in step order, pass by pass) -- the family rule picks one, both must agree with the one-buffer form."""
    monkeypatch.setenv("MGX_LG_MERGE", merge)
    N, T = (1500, 400) if "Memory" not in env_id else (400, 300)
    T = {"MiniGrid-SimpleCrossingS11N5-v0": 520, "MiniGrid-DoorKey-8x8-v0": 700}.get(env_id, T)   # (time-outs at 484 / 640 steps)
    a, b = pair(env_id, N, monkeypatch)
    assert np.array_equal(to_np(a.reset()), to_np(b.reset()))
    acts = to_np(a.fill_actions(3, 0, T))
    if not any(f in env_id for f in RAW_ACTIONS):
        acts = np.where(acts > 2, 2, acts).astype(np.uint8)       # mostly forward: episodes a few steps long, back-to-back dones
    for t in range(T):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        assert np.array_equal(to_np(da), to_np(db)), t
        assert np.array_equal(to_np(oa), to_np(ob)), t
        assert np.array_equal(to_np(ra), to_np(rb)), t
        if t % 97 == 0:
            assert same_state(a, b), t
    assert same_state(a, b)
    assert a.stats()["episodes"] == b.stats()["episodes"] and a.stats()["episodes"] > 0
    a.close(); b.close()


def test_ring_generator_caller_side_boundaries(monkeypatch):
    """Masked seeded resets, plain resets (the ring's consumer on the caller's side) and injected states in between steps."""
    env_id, N, T = "MiniGrid-LavaCrossingS9N1-v0", 1100, 240
    a, b = pair(env_id, N, monkeypatch)
    assert np.array_equal(to_np(a.reset()), to_np(b.reset()))
    acts = to_np(a.fill_actions(11, 0, T))
    acts = np.where(acts > 2, 2, acts).astype(np.uint8)
    rs = np.random.RandomState(2)
    g0, a0 = mg.generate_levels(env_id, np.arange(N, dtype=np.uint64) + 900)
    for t in range(T):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        assert np.array_equal(to_np(da), to_np(db)) and np.array_equal(to_np(oa), to_np(ob)) and np.array_equal(to_np(ra), to_np(rb)), t
        kind = t % 12
        if kind in (3, 7, 10):
            m = (rs.rand(N) < (0.3 if kind != 10 else 1.1)).astype(np.uint8)
            mask = None if kind == 10 and t % 24 == 10 else m
            reseed = kind == 7
            assert np.array_equal(to_np(a.reset(mask, reseed=reseed)), to_np(b.reset(mask, reseed=reseed))), (t, kind)
            assert same_state(a, b), t
        elif kind == 5 and t < 100:
            a.set_state(g0, a0); b.set_state(g0, a0)
            assert np.array_equal(to_np(a.observe()), to_np(b.observe())), t
    assert same_state(a, b)
    assert a.stats()["episodes"] == b.stats()["episodes"]
    a.close(); b.close()


@pytest.mark.parametrize("T", [1, 2, 7, 32])
def test_ring_generator_in_captured_rollouts(T, monkeypatch):
    """mgx_rollout captures the per-step launches: the graph starts from "no flags waiting" and drains behind its last step, so replays,
    single steps and resets may follow each other in any order."""
    import torch
    env_id, N = "MiniGrid-DoorKey-5x5-v0", 912
    a, b = pair(env_id, N, monkeypatch)
    a.reset(); b.reset()
    for rnd in range(5):
        acts = a.fill_actions(20 + rnd, 0, T)
        acts = torch.where(acts > 2, torch.full_like(acts, 2), acts).contiguous()
        ra = a.rollout(acts)
        ob, rb, db = [], [], []
        for t in range(T):
            o, r, d, _ = b.step(acts[t])
            ob.append(to_np(o).copy()); rb.append(to_np(r).copy()); db.append(to_np(d).copy())
        assert np.array_equal(to_np(ra[0]), np.stack(ob)) and np.array_equal(to_np(ra[1]), np.stack(rb)) and np.array_equal(to_np(ra[2]), np.stack(db)), rnd
        assert same_state(a, b), rnd
        if rnd % 2:
            o1 = a.step(acts[0]); o2 = b.step(acts[0])
            assert np.array_equal(to_np(o1[0]), to_np(o2[0]))
        if rnd == 2:
            assert np.array_equal(to_np(a.reset(reseed=False)), to_np(b.reset(reseed=False)))
    a.close(); b.close()


@pytest.mark.parametrize("ring", ["16", "off"])
@pytest.mark.parametrize("env_id", ["MiniGrid-LavaCrossingS9N1-v0", "MiniGrid-Fetch-6x6-N2-v0"])
def test_injected_state_draws_nothing(env_id, ring, monkeypatch):
    """set_state / set_task on a new_level_each_episode handle are attribute assignments in the reference (env.grid, env.agent_pos, ...): the
    env's RNG stream is untouched, so the episode after the injected one is level 1 of the env's stream -- generate_level_stream(seed_i)[1]."""
    monkeypatch.setenv("MGX_LG_RING", ring)
    N, seed = 300, 40
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seed, auto_reset=True, new_level_each_episode=True, backend="numpy")
    env.reset()
    levels = [mg.generate_level_stream(env_id, seed + i, 3, with_task=True) for i in range(N)]
    G = np.stack([lv[0] for lv in levels]); A = np.stack([lv[1] for lv in levels]); K = np.stack([lv[2] for lv in levels])
    g_other, a_other = mg.generate_levels(env_id, np.arange(N, dtype=np.uint64) + 7000)
    env.set_state(g_other, a_other, steps=np.full(N, env.cfg.max_steps - 1, np.int32))   # the injected episode ends with the next step
    if env.cfg.task_kind:
        env.set_task(np.full(N, 0x0205, np.uint32))
    obs, rew, done, _ = env.step(np.zeros(N, np.uint8))
    assert done.all()
    st = env.get_state()
    assert np.array_equal(st["grid"], G[:, 1]) and np.array_equal(st["agent"], A[:, 1])
    if env.cfg.task_kind:
        assert np.array_equal(env.get_task() & 0xFF, K[:, 1] & 0xFF)
    env.step(np.zeros(N, np.uint8))
    env.set_state(g_other, a_other, steps=np.full(N, env.cfg.max_steps - 1, np.int32))
    obs, rew, done, _ = env.step(np.zeros(N, np.uint8))
    assert done.all()
    st = env.get_state()
    assert np.array_equal(st["grid"], G[:, 2]) and np.array_equal(st["agent"], A[:, 2])
    env.close()


def test_ring_is_the_default_form(monkeypatch):
    """The ring is what a handle gets without MGX_LG_RING (and what the side-by-side tests above therefore compare): its 16 next-level buffers
    per env show in the device's free memory; FullyObs handles (and the families of mgx_create's rule) keep one buffer."""
    import torch
    N = 262144

    def cost(env_id, **kw):
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        env = mg.VecMiniGrid(env_id, num_envs=N, seeds=1, auto_reset=True, new_level_each_episode=True, backend="torch", **kw)
        env.reset(); env.sync()
        used = free0 - torch.cuda.mem_get_info()[0]
        env.close()
        return used
    per_level = N * (84 + 8)                                 # LavaCrossingS9N1: ceil4(81) cells + the agent record
    default = cost("MiniGrid-LavaCrossingS9N1-v0")
    monkeypatch.setenv("MGX_LG_RING", "off")
    single = cost("MiniGrid-LavaCrossingS9N1-v0")
    monkeypatch.delenv("MGX_LG_RING")
    assert 13 * per_level < default - single < 20 * per_level, (default, single, per_level)
    full = cost("MiniGrid-LavaCrossingS9N1-v0", obs_mode="full")
    monkeypatch.setenv("MGX_LG_RING", "off")
    full_single = cost("MiniGrid-LavaCrossingS9N1-v0", obs_mode="full")
    assert abs(full - full_single) < 6 * per_level, (full, full_single)         # (allocator noise; a ring would be 15)


def test_ring_generator_under_the_wrappers(monkeypatch):
    """The bonus / DAC kernels (k_step_wrap) know the ring like every step_body instance: StateBonus(DACWrapper(env)) on a ring handle
    equals the same on a one-buffer handle, through in-kernel resets at the wrapper's time-outs."""
    env_id, N, T = "MiniGrid-LavaCrossingS9N1-v0", 1100, 700
    a, b = pair(env_id, N, monkeypatch)
    for e in (a, b):
        e.reset()
        e.set_dac(True)
        e.add_bonus("state")
    assert a.step_kernel_name() == "k_step_wrap<0,7>"
    acts = to_np(a.fill_actions(13, 0, T))
    acts = np.where(acts > 2, 2, acts).astype(np.uint8)
    for t in range(T):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        assert np.array_equal(to_np(da), to_np(db)) and np.array_equal(to_np(oa), to_np(ob)) and np.array_equal(to_np(ra), to_np(rb)), t
    assert same_state(a, b) and np.array_equal(a.bonus_counts("state"), b.bonus_counts("state"))
    assert a.stats()["episodes"] == 2 * N        # every episode is max_steps (324) long under the DAC wrapper
    a.close(); b.close()
