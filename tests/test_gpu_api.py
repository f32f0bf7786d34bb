"""C-ABI behaviours that the Python mirror does not exercise by itself: optional outputs, streams, graphs, stats."""
import ctypes

import numpy as np
import pytest
import torch

import gym_minigrid_amd as mg
from gym_minigrid_amd import _lib
from helpers import make_oracle, random_states, to_np

pytestmark = pytest.mark.gpu


def test_optional_outputs_and_raw_calls():
    """obs / reward / done may each be NULL; state still advances identically."""
    N, W, H = 200, 8, 8
    grid, aux, agent, carry, steps = random_states(N, W, H, seed=4)
    acts = np.random.RandomState(0).randint(0, 7, size=(30, N)).astype(np.uint8)
    ref = make_oracle(W, H, 15, False, False, grid, aux, agent, carry, steps)
    for mode in ("partial", "full"):
        c = mg.Config()
        c.width, c.height, c.max_steps = W, H, 15
        env = mg.VecMiniGrid(config=c, num_envs=N, obs_mode=mode, auto_reset=True, backend="numpy")
        env.set_state(grid, agent, aux=aux, carry=carry, steps=steps)
        orc = make_oracle(W, H, 15, False, False, grid, aux, agent, carry, steps)
        L = _lib.lib()
        done = np.zeros(N, np.uint8)
        for t in range(30):
            a = acts[t]
            if t % 3 == 0:    # nothing but the transition
                _lib.check(L.mgx_step(env._h, a.ctypes.data, None, None, None))
            elif t % 3 == 1:  # done only
                _lib.check(L.mgx_step(env._h, a.ctypes.data, None, None, done.ctypes.data))
            else:
                env.step(a)
            oo, orew, odone = orc.step(a)
            orc.reset_where(odone)
            if t % 3 == 1:
                assert np.array_equal(done, odone)
        st = env.get_state()
        assert np.array_equal(st["grid"], orc.grid) and np.array_equal(st["agent"], orc.agent) and np.array_equal(st["steps"], orc.steps)
        assert env.stats()["steps"] == 30 * N
        env.close()
    del ref


def test_hipgraph_replay_equals_eager():
    """mgx_step does no allocation / synchronisation with device pointers, so it can be captured in a hipGraph."""
    N, T = 4096, 12
    env = mg.VecMiniGrid("MiniGrid-DoorKey-8x8-v0", num_envs=N, seeds=0)
    ref = mg.VecMiniGrid("MiniGrid-DoorKey-8x8-v0", num_envs=N, seeds=0)
    env.reset(); ref.reset()
    acts = env.fill_actions(1, 0, T)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        env.step(acts[0])
    ref.step(acts[0])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for t in range(1, T):
            env.step(acts[t])
    g.replay()
    torch.cuda.synchronize()
    for t in range(1, T):
        ref.step(acts[t])
    torch.cuda.synchronize()
    assert torch.equal(env._obs, ref._obs) and torch.equal(env._done, ref._done) and torch.equal(env._reward, ref._reward)
    a, b = env.get_state(), ref.get_state()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    env.close(); ref.close()


def test_two_handles_two_streams_and_stats():
    N = 3000
    e1 = mg.VecMiniGrid("MiniGrid-LavaCrossingS9N1-v0", num_envs=N, seeds=0)
    e2 = mg.VecMiniGrid("MiniGrid-LavaCrossingS9N1-v0", num_envs=N, seeds=0)
    e1.reset(); e2.reset()
    acts = e1.fill_actions(3, 0, 50)
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    dones = 0
    for t in range(50):
        o1, r1, d1, _ = e1.step(acts[t])
        with torch.cuda.stream(s2):
            o2, r2, d2, _ = e2.step(acts[t])
        dones += int(d1.sum().item())
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(d1, d2)
    st = e1.stats()
    assert st["episodes"] == dones == e2.stats()["episodes"] and st["steps"] == 50 * N
    out2 = torch.zeros(2, dtype=torch.float64, device=o1.device)
    e1.read_stats_async(out2)
    torch.cuda.synchronize()
    assert out2.tolist() == [float(dones), st["reward_sum"]]
    e1.close(); e2.close()


@pytest.mark.parametrize("env_id,N,T", [("MiniGrid-DoorKey-8x8-v0", 4096, 16), ("MiniGrid-LavaCrossingS9N1-v0", 1024, 24),
                                        ("MiniGrid-Dynamic-Obstacles-6x6-v0", 512, 12)])
def test_rollout_graph_equals_stepping(env_id, N, T):
    """mgx_rollout (T steps captured into one hipGraph, replayed) == T mgx_step calls, on every output byte; the replay
    of the cached graph continues the episodes.  (DoorKey takes the fused one-launch form here: see the next test.)"""
    seeds = np.arange(N, dtype=np.uint64)
    a_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch", new_level_each_episode=("Lava" in env_id))
    b_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch", new_level_each_episode=("Lava" in env_id))
    a_env.reset(); b_env.reset()
    for rep in range(3):                                     # rep 0 captures, 1 and 2 replay the same graph
        acts = a_env.fill_actions(11, rep * T, T)            # (T, N) uint8 on the device
        acts_keep = acts if rep == 0 else acts_keep
        if rep:                                              # same buffer as the captured one: copy the new actions in
            acts_keep.copy_(acts)
        obs, rew, done = a_env.rollout(acts_keep)
        torch.cuda.synchronize()
        for t in range(T):
            o, r, d, _ = b_env.step(acts_keep[t])
            assert torch.equal(obs[t], o), (rep, t)
            assert torch.equal(rew[t], r) and torch.equal(done[t], d), (rep, t)
    sa, sb = a_env.stats(), b_env.stats()
    assert sa == sb and sa["steps"] == 3 * T * N
    with pytest.raises(mg.MgxError):
        host = np.zeros((T, N), np.uint8)
        _lib.check(_lib.lib().mgx_rollout(a_env._h, T, host.ctypes.data, None, None, None))
    a_env.close(); b_env.close()


@pytest.mark.parametrize("form", ["fused", "graph"])
@pytest.mark.parametrize("env_id,N,T,auto,view", [
    ("MiniGrid-DoorKey-8x8-v0", 1008, 700, True, 7),        # time-outs at 640: resets of dirty envs (N: multiples of 16, the API's obs alignment)
    ("MiniGrid-LavaCrossingS9N1-v0", 1552, 90, True, 7),     # resets every step, nothing dirty
    ("MiniGrid-Fetch-8x8-N3-v0", 784, 60, True, 7),          # task rule; the terminal pickup is undone
    ("MiniGrid-Empty-16x16-v0", 208, 40, True, 7),           # 3 waves per block; the single step takes the gather form, the rollout the staged tile
    ("MiniGrid-KeyCorridorS3R3-v0", 336, 80, False, 7),      # caller resets: cells keep changing
    ("MiniGrid-Unlock-v0", 64, 50, True, 7),
    # round 3: the run-time-size instance (grids without a sized one) and the other view sizes
    ("MiniGrid-FourRooms-v0", 400, 260, True, 7),            # 19x19 (time-outs at 500): the gather form's graph either way
    ("MiniGrid-MultiRoom-N6-v0", 144, 130, True, 7),         # 25x25
    ("MiniGrid-KeyCorridorS3R2-v0", 336, 300, True, 7),      # 7x5: no sized instance -> k_rollout<0,0,7> (time-outs at 270)
    ("MiniGrid-DistShift1-v0", 336, 300, True, 7),           # (lava ends episodes, time-outs)
    ("MiniGrid-DoorKey-8x8-v0", 1008, 90, True, 3),
    ("MiniGrid-LavaCrossingS9N1-v0", 1552, 90, True, 5),
    ("MiniGrid-DoorKey-16x16-v0", 208, 60, True, 9),
    ("MiniGrid-Fetch-8x8-N3-v0", 784, 60, True, 11),
    ("MiniGrid-LockedRoom-v0", 144, 60, False, 9),
    ("MiniGrid-RedBlueDoors-6x6-v0", 336, 90, True, 5)])     # 12x6
def test_rollout_fused_equals_stepping(env_id, N, T, auto, view, form, monkeypatch):
    """k_rollout (all T steps in ONE launch, the tile resident in LDS) against T mgx_step calls: observations, rewards, dones of
    every step, then state, task words and counters; a second rollout continues from the written-back state.  MGX_ROLLOUT=graph
    runs the same check on the captured-graph form."""
    if form == "graph":
        monkeypatch.setenv("MGX_ROLLOUT", "graph")
    seeds = np.arange(N, dtype=np.uint64) * 3 + 1
    a_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch", auto_reset=auto, agent_view_size=view)
    b_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch", auto_reset=auto, agent_view_size=view)
    a_env.reset(); b_env.reset()
    chunk = 64 if T > 200 else T
    done_total = 0
    for rep, t0 in enumerate(range(0, 2 * T, chunk)):
        n = min(chunk, 2 * T - t0)
        acts = a_env.fill_actions(21, t0, n)
        obs, rew, done = a_env.rollout(acts)
        for t in range(n):
            o, r, d, _ = b_env.step(acts[t])
            assert torch.equal(obs[t], o), (rep, t)
            assert torch.equal(rew[t], r) and torch.equal(done[t], d), (rep, t)
        done_total += int(done.sum())
        if rep % 4 == 0:
            sa, sb = a_env.get_state(), b_env.get_state()
            for k in sa:
                assert np.array_equal(sa[k], sb[k]), (k, rep)
    if 2 * T >= a_env.max_steps or any(k in env_id for k in ("Lava", "Fetch", "DistShift")):   # (the others are too short for an episode to end)
        assert done_total > 0
    assert a_env.stats() == b_env.stats()
    if a_env.cfg.task_kind:
        assert np.array_equal(a_env.get_task(), b_env.get_task())
    o2, _, _ = a_env.rollout(acts[:1], with_obs=False)      # reward / done only: no observation stream at all
    assert o2 is None
    a_env.close(); b_env.close()


@pytest.mark.parametrize("form", ["fused", "graph"])
@pytest.mark.parametrize("env_id,N,T,auto", [
    ("MiniGrid-DoorKey-8x8-v0", 1008, 700, True),           # 8x8: 16-cell units (the transposed store path); time-outs at 640
    ("MiniGrid-Empty-16x16-v0", 208, 40, True),             # configs[4]'s grid
    ("MiniGrid-LavaCrossingS9N1-v0", 1552, 90, True),        # 9x9: 81 cells, the 12-byte-record path; resets every step
    ("MiniGrid-Fetch-8x8-N3-v0", 784, 60, True),             # task rule, the terminal pickup undone under the agent's marker
    ("MiniGrid-KeyCorridorS3R3-v0", 336, 80, False),         # 7x7, caller resets
    ("MiniGrid-KeyCorridorS3R2-v0", 336, 300, True),         # 7x5: the run-time-size instance
    ("MiniGrid-RedBlueDoors-6x6-v0", 336, 90, True),         # 12x6: the agent stands in open doors
    ("MiniGrid-FourRooms-v0", 400, 60, True)])               # 19x19: past 16x16 the captured graph of direct-form steps either way
def test_rollout_fully_observable_fused_equals_stepping(env_id, N, T, auto, form, monkeypatch):
    """k_rollout with the FullyObsWrapper observation (emit_full_obs on the resident tile, the agent's marker put into the LDS image and
    taken out again every step) against T mgx_step calls of the direct form: every output byte, then state and counters."""
    if form == "graph":
        monkeypatch.setenv("MGX_ROLLOUT", "graph")
    seeds = np.arange(N, dtype=np.uint64) * 3 + 1
    a_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch", auto_reset=auto, obs_mode="full")
    b_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch", auto_reset=auto, obs_mode="full")
    a_env.reset(); b_env.reset()
    chunk = 64 if T > 200 else T
    done_total = 0
    for rep, t0 in enumerate(range(0, 2 * T, chunk)):
        n = min(chunk, 2 * T - t0)
        acts = a_env.fill_actions(22, t0, n)
        obs, rew, done = a_env.rollout(acts)
        for t in range(n):
            o, r, d, _ = b_env.step(acts[t])
            assert torch.equal(obs[t], o), (rep, t)
            assert torch.equal(rew[t], r) and torch.equal(done[t], d), (rep, t)
        done_total += int(done.sum())
        if rep % 4 == 0:
            sa, sb = a_env.get_state(), b_env.get_state()
            for k in sa:
                assert np.array_equal(sa[k], sb[k]), (k, rep)
    if 2 * T >= a_env.max_steps or any(k in env_id for k in ("Lava", "Fetch")):
        assert done_total > 0
    assert a_env.stats() == b_env.stats()
    a_env.close(); b_env.close()


@pytest.mark.parametrize("N", [300, 1500])                 # (one partial 512-env span / whole spans + a partial one)
@pytest.mark.parametrize("env_id", ["MiniGrid-DoorKey-8x8-v0", "MiniGrid-LavaCrossingS9N1-v0", "MiniGrid-Fetch-8x8-N3-v0"])
def test_masked_reset_same_and_changed_seeds(env_id, N):
    """reset(mask): an env that keeps its seed is restored from the episode-start snapshot (no re-seeding), one whose
    seed changed is re-seeded and regenerated; either way the state equals `seed(s); reset()` of the reference."""
    seeds = np.arange(N, dtype=np.uint64) + 7
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=False, backend="numpy")
    env.reset()
    rs = np.random.RandomState(0)
    for t in range(30):
        env.step(rs.randint(0, 7, size=N).astype(np.uint8))
    mask = (rs.uniform(size=N) < 0.5).astype(np.uint8)
    changed = (rs.uniform(size=N) < 0.5)
    new_seeds = np.where(changed, seeds + 1000, seeds).astype(np.uint64)
    before = env.get_state()
    env.seed(new_seeds)
    env.reset(mask=mask)
    st = env.get_state()
    grid, agent, task = mg.generate_levels(env_id, new_seeds, with_task=True)
    m = mask.astype(bool)
    assert np.array_equal(st["grid"][m], grid[m]) and np.array_equal(st["agent"][m], agent[m])
    assert (st["steps"][m] == 0).all() and (st["carry"][m] == (1, 0, 0)).all()
    for k in ("grid", "agent", "steps", "carry"):
        assert np.array_equal(st[k][~m], before[k][~m]), k
    if env.cfg.task_kind:
        assert np.array_equal(env.get_task()[m], task[m])
    # and once more with everything unchanged (pure snapshot restore), after more steps
    for t in range(10):
        env.step(rs.randint(0, 7, size=N).astype(np.uint8))
    env.reset(mask=mask)
    st = env.get_state()
    assert np.array_equal(st["grid"][m], grid[m]) and np.array_equal(st["agent"][m], agent[m])
    env.close()
    # device buffers: the masked reset rewrites only the tiles that hold a reset env; together with what the last
    # step left in the buffer that is the observation of every env
    tenv = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=False, backend="torch")
    tenv.reset()
    for t in range(12):
        tenv.step(torch.from_numpy(rs.randint(0, 7, size=N).astype(np.uint8)).cuda())
    sparse = np.zeros(N, np.uint8)
    sparse[[3, 200]] = 1                                     # tiles 0 and 3 of 5
    got = tenv.reset(mask=sparse).clone()
    assert torch.equal(got, tenv.observe())
    tenv.close()


def test_reset_restores_default_object_state():
    """object_state handles of a built-in family: mgx_reset puts the hidden Goal/Box state back to the defaults."""
    N = 130
    env = mg.VecMiniGrid("MiniGrid-Empty-6x6-v0", num_envs=N, seeds=0, auto_reset=False, backend="numpy", object_state=True)
    env.reset()
    a = np.tile(np.array([2, 2, 2, 1, 2, 2, 5, 5], np.uint8), 4)   # walk to the goal (6x6: (4,4)) and toggle it away
    for t in range(len(a)):
        env.step(np.full(N, a[t], np.uint8))
    env.reset()
    st, os_ = env.get_state(), env.get_object_state()
    grid, agent = mg.generate_levels("MiniGrid-Empty-6x6-v0", np.arange(N, dtype=np.uint64))
    assert np.array_equal(st["grid"], grid) and (st["aux"] == 0).all()
    assert (os_["contains"] == np.array([1, 0, 0], np.uint8)).all() and (os_["carry_aux"] == 0).all()
    env.close()


@pytest.mark.parametrize("case,mission", [("DoorKey-8x8", "use the key to open the door and then get to the goal"),
                                          ("Empty-5x5", "get to the green goal square"),
                                          ("LavaCrossingS9N3", "avoid the lava and get to the green goal square"), ("MultiRoom-N2-S4", None)])
def test_single_env_adapter_replays_reference_loop(case, mission):
    """gym_minigrid_amd.make(id): the reference's own caller loop (seed, reset, step, `if done: seed; reset`) against a
    recorded trace -- obs dict, float reward, bool done, the attributes callers read."""
    from conftest import load_case
    meta, z = load_case(case)
    assert meta["reseed"] is True
    env = mg.make("MiniGrid-%s-v0" % case)
    assert env.actions.forward == 2 and env.action_space.n == 7
    for k in range(2):
        seed = int(z["seed"][k])
        env.seed(seed)
        obs = env.reset()
        assert set(obs) == {"image", "direction", "mission"}
        assert np.array_equal(obs["image"], z["init_obs"][k])
        if mission is not None:
            assert obs["mission"] == mission == env.mission
        assert np.array_equal(env.encode_grid(), z["init_grid"][k]) and env.step_count == 0 and env.carrying is None
        for t in range(min(z["actions"].shape[1], 300)):
            obs, reward, done, info = env.step(int(z["actions"][k, t]))
            assert isinstance(reward, float) and isinstance(done, bool) and info == {}
            assert np.array_equal(obs["image"], z["obs"][k, t]), (case, k, t)
            assert obs["direction"] == z["direction"][k, t] == env.agent_dir
            assert reward == float(np.float32(z["reward"][k, t])) and done == bool(z["done"][k, t])
            assert env.agent_pos == tuple(z["agent"][k, t, :2]) and env.step_count == z["steps"][k, t]
            if done:
                env.seed(seed)
                obs = env.reset()
                assert np.array_equal(obs["image"], z["init_obs"][k])
    env.close()


def _load_tool(name):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("mgx_tool_" + name, os.path.join(os.path.dirname(__file__), "..", "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_masked_reset_keeps_running_task_word_of_other_envs():
    """TwoGoals' task word is the running goal count: a caller-side reset(mask) of the finished envs must not touch the
    count of the envs still in their episode (fixed-layout ids take mgx_reset's host path: found by tools/fuzz_ids.py)."""
    from oracle.minigrid_oracle import OracleEnvs
    env_id, N = "MiniGrid-TwoGoals-6x6-v0", 700
    cfg = mg.env_config(env_id)
    seeds = np.arange(N, dtype=np.uint64)
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=False, backend="torch")
    env.reset()
    grid, agent, task = mg.generate_levels(env_id, seeds, with_task=True)
    orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, task=cfg.task_kind)
    orc.set_state(grid, agent)
    orc.task = task.copy()
    rs = np.random.RandomState(3)
    resets = 0
    for t in range(150):
        a = rs.choice([0, 1, 2, 2, 2, 5, 6], size=N).astype(np.uint8)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        oo, orew, odone = orc.step(a)
        assert np.array_equal(to_np(done), odone), t
        assert np.array_equal(to_np(rew), orew.astype(np.float32)), t
        assert np.array_equal(to_np(obs), oo), t
        if odone.any():
            env.reset(mask=done)
            orc.reset_where(odone)
            resets += int(odone.sum())
        assert np.array_equal(env.get_task(), orc.task), t
    assert resets > 50
    env.close()


def test_every_env_id_fuzz_round():
    """tools/fuzz_ids.py, one seeded round: all built-in ids, random 64-bit seeds / batch size / obs mode, in-kernel
    auto-reset or the caller's reset(mask=done) loop, against the oracle on host-generated levels (25 more rounds were
    run once: profiles/r01_fuzz.log)."""
    fz = _load_tool("fuzz_ids")
    rs = np.random.RandomState(2024)
    for env_id in mg.env_ids():
        fz.one(env_id, rs)
        fz.one_stream(env_id, rs)   # new_level_each_episode against generate_level_stream
        fz.one_epilogue(env_id, rs)  # one-hot / flat modes against the wrappers' formulas on a twin env
        fz.one_rollout(env_id, rs)   # hipGraph rollout (capture + replays) against single steps on a twin env
        fz.one_options(env_id, rs)   # view size / extended actions / default_vis=False / object_state on the built-in ids


@pytest.mark.parametrize("env_id", ["MiniGrid-Empty-8x8-v0", "MiniGrid-DistShift2-v0", "MiniGrid-TwoGoals-8x8-v0"])
def test_one_level_families_reset_on_the_device(env_id):
    """Families whose level does not depend on the seed: the first full reset hands one host-generated level to every env,
    later resets (full or masked) restore the snapshot on the device; a state injection in between takes the masked reset
    back to the per-env host path (the unmasked envs must keep what was injected)."""
    N = 300
    cfg = mg.env_config(env_id)
    grid, agent = mg.generate_levels(env_id, np.zeros(N, np.uint64))
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=np.arange(N, dtype=np.uint64) * 977, auto_reset=False, backend="torch")
    env.reset()
    st = env.get_state()
    assert np.array_equal(st["grid"], grid) and np.array_equal(st["agent"], agent)
    rs = np.random.RandomState(0)
    for _ in range(12):
        env.step(rs.randint(0, 3, size=N).astype(np.uint8))
    moved = env.get_state()
    mask = (np.arange(N) % 3 == 0).astype(np.uint8)
    env.reset(mask=mask)                                     # device restore of the masked envs only
    st = env.get_state()
    m = mask.astype(bool)
    assert np.array_equal(st["agent"][m], agent[m]) and (st["steps"][m] == 0).all() and np.array_equal(st["grid"][m], grid[m])
    assert np.array_equal(st["agent"][~m], moved["agent"][~m]) and np.array_equal(st["steps"][~m], moved["steps"][~m])
    # inject a state, then a masked reset: masked envs get the level, the others keep the injected one
    g2 = grid.copy()
    g2[:, 1, 2] = (6, 2, 0) if tuple(agent[0, :2]) != (1, 2) else (1, 0, 0)
    a2 = agent.copy()
    env.set_state(g2, a2)
    env.reset(mask=torch.from_numpy(mask).cuda())
    st = env.get_state()
    assert np.array_equal(st["grid"][m], grid[m]) and np.array_equal(st["grid"][~m], g2[~m])
    env.reset()                                              # full reset: everyone back on the level
    st = env.get_state()
    assert np.array_equal(st["grid"], grid) and np.array_equal(st["agent"], agent) and (st["steps"] == 0).all()
    if cfg.task_kind:
        assert (env.get_task() == 0).all()
    env.close()


def test_reset_of_a_random_family_beyond_64x64_runs_on_the_host_generator():
    """W*H > 4096 with a family that draws random numbers: per-env host generation inside mgx_reset, with the seeds and
    the mask handed over as device tensors."""
    cfg = mg.env_config("MiniGrid-Empty-Random-6x6-v0")
    cfg.width, cfg.height, cfg.max_steps = 70, 66, 50
    N = 130
    seeds = np.arange(N, dtype=np.uint64) * 104729 + 11
    env = mg.VecMiniGrid(config=cfg, num_envs=N, seeds=seeds, auto_reset=False, backend="torch")
    obs = env.reset()
    grid, agent = mg.generate_levels(cfg, seeds)
    st = env.get_state()
    assert np.array_equal(st["grid"], grid) and np.array_equal(st["agent"], agent)
    assert len({tuple(a) for a in agent}) > 20               # the starts really are random
    orc = make_oracle(70, 66, 50, cfg.see_through_walls, cfg.lava_v1, grid, np.zeros((N, 70, 66), np.uint8), agent)
    assert np.array_equal(to_np(obs), orc.observe())
    for _ in range(5):
        env.step(np.full(N, 2, np.uint8))
    mask = torch.from_numpy((np.arange(N) % 2).astype(np.uint8)).cuda()
    env.reset(mask=mask)
    st2 = env.get_state()
    m = (np.arange(N) % 2).astype(bool)
    assert np.array_equal(st2["agent"][m], agent[m]) and (st2["steps"][m] == 0).all() and (st2["steps"][~m] == 5).all()
    env.close()


def test_profile_span_and_per_launch_samples():
    """mgx_profile_*: one event pair around the span + one around every n-th launch of the step kernel alone."""
    env = mg.VecMiniGrid("MiniGrid-DoorKey-8x8-v0", num_envs=65536, seeds=0, backend="torch")
    env.reset()
    acts = env.fill_actions(1, 0, 12)
    env.profile_begin(stride=3)
    for t in range(12):
        env.step(acts[t])
    launches, span_ms = env.profile_end()
    samples, kernel_ms = env.profile_kernel()
    assert launches == 12 and samples == 4                       # launches 0, 3, 6, 9
    assert 0 < kernel_ms / samples < 1.0 and span_ms > 0         # a 65,536-env step is ~10 us
    assert kernel_ms / samples < 3 * span_ms / launches + 0.05   # same order: both bound the kernel from above
    with pytest.raises(mg.MgxError):
        env.profile_end()                                        # not running any more
    env.profile_begin()                                          # default stride 8; rollouts count launches but are not sampled
    env.rollout(acts)
    launches, _ = env.profile_end()
    assert launches == 12 and env.profile_kernel()[0] == 0
    # profile_stop: the end marker goes out without a wait; work enqueued after it is outside the span
    env.profile_begin(stride=100)
    for t in range(4):
        env.step(acts[t])
    env.profile_stop()
    with pytest.raises(mg.MgxError):
        env.profile_stop()                                       # one end marker per span
    for t in range(4, 12):
        env.step(acts[t])                                        # (counted as launches, not timed)
    torch.cuda.synchronize()
    launches, span4 = env.profile_end()
    assert launches == 12 and env.profile_kernel()[0] == 1      # stride > launches: the first launch only
    env.profile_begin(stride=100)
    for t in range(12):
        env.step(acts[t])
    _, span12 = env.profile_end()
    assert 0 < span4 < span12
    env.close()


def test_step_kernel_name_follows_the_selector():
    """mgx_step_kernel_name: the instantiation the handle launches (what bench.py prints as roofline.kernel and rocprofv3 as the kernel)."""
    want = {("MiniGrid-Empty-8x8-v0", "partial", 7): "k_step<8,8,0,7>",
            ("MiniGrid-LavaCrossingS9N1-v0", "partial", 7): "k_step<9,9,0,7>",
            ("MiniGrid-Empty-16x16-v0", "partial", 7): "k_step<16,16,3,7>",          # from 13x13 up the view is gathered, not staged
            ("MiniGrid-MemoryS13-v0", "partial", 7): "k_step<13,13,3,7>",
            ("MiniGrid-FourRooms-v0", "partial", 7): "k_step<19,19,3,7>",
            ("MiniGrid-ObstructedMaze-2Dlhb-v0", "partial", 7): "k_step<16,16,3,7,obj>",
            ("MiniGrid-ObstructedMaze-1Dlhb-v0", "partial", 7): "k_step<11,6,0,7,obj>",
            ("MiniGrid-DoorKey-8x8-v0", "partial", 5): "k_step<0,0,0,5>",
            ("MiniGrid-FourRooms-v0", "partial", 5): "k_step<0,0,3,5,obj>",     # (the run-time-size instance carries the object-plane code)
            ("MiniGrid-Empty-16x16-v0", "full", 7): "k_step_fulldirect<16,16>",
            ("MiniGrid-FourRooms-v0", "full", 7): "k_step_fulldirect<19,19,ragged>",
            ("MiniGrid-LavaCrossingS9N1-v0", "full", 7): "k_step<9,9,1,7>"}
    for (env_id, mode, view), name in want.items():
        env = mg.VecMiniGrid(env_id, num_envs=64, obs_mode=mode, agent_view_size=view, backend="numpy")
        assert env.step_kernel_name() == name, (env_id, mode, view, env.step_kernel_name())
        env.close()


@pytest.mark.parametrize("env_id", ["MiniGrid-DoorKey-16x16-v0", "MiniGrid-MemoryS13Random-v0", "MiniGrid-DoorKey-8x8-v0"])
def test_step_after_fused_rollout_and_after_set_state(env_id):
    """Handles whose single step takes the gather form keep a per-env "cell in front" byte between steps; the fused rollout and
    set_state change poses and cells behind its back and must leave it unknown: step() right after either equals the reference env."""
    N, T = 208, 40
    seeds = np.arange(N, dtype=np.uint64) + 5
    a_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch")
    b_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch")
    a_env.reset(); b_env.reset()
    acts = a_env.fill_actions(5, 0, 3 * T)
    for t in range(T):                              # (fills the cache)
        oa, ra, da, _ = a_env.step(acts[t]); ob, rb, db, _ = b_env.step(acts[t])
        assert torch.equal(oa, ob)
    a_env.rollout(acts[T:2 * T].contiguous())
    for t in range(T, 2 * T):
        b_env.step(acts[t])
    for t in range(2 * T, 2 * T + 10):
        oa, ra, da, _ = a_env.step(acts[t]); ob, rb, db, _ = b_env.step(acts[t])
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), t
    st = b_env.get_state()
    if a_env.cfg.task_kind == 0:                    # set_state: rotate every agent, keep the rest
        ag = st["agent"].copy(); ag[:, 2] = (ag[:, 2] + 1) % 4
        for e in (a_env, b_env):
            e.set_state(st["grid"], ag, carry=st["carry"], steps=st["steps"])
        for t in range(2 * T + 10, 2 * T + 20):
            oa, _, _, _ = a_env.step(acts[t]); ob, _, _, _ = b_env.step(acts[t])
            assert torch.equal(oa, ob), t
        cfg = mg.env_config(env_id)
        from helpers import make_oracle
        orc = make_oracle(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, a_env.get_state()["grid"], a_env.get_state()["aux"],
                          a_env.get_state()["agent"], carry=a_env.get_state()["carry"], steps=a_env.get_state()["steps"])
        assert np.array_equal(a_env.observe().cpu().numpy(), orc.observe())
    a_env.close(); b_env.close()
