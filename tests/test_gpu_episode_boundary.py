"""The episode boundaries of the reference on the HIP path, through the C ABI (include/mgx.h):

  * mgx_reset(h, seeds = NULL, mask, obs) without a schedule = the plain `reset()` of the caller loop `if done: env.reset()`
    (minigrid.py:831-858, run_tests.py:64-66): the env's own MT19937 stream continues on the GPU and the next level is drawn from it
    (Dynamic-Obstacles: from where the obstacle walks left it);
  * mgx_set_seed_schedule = ReseedWrapper(env, seeds=[s0..sK-1], seed_idx) (wrappers.py:12-28): in-kernel (auto_reset) and caller-side.

Against the traces recorded from the reference (the seed-list ones through the wrapper class itself), and on larger batches against the
host generator + CPU oracle / the Dynamic-Obstacles restatement."""
import numpy as np
import pytest

import gym_minigrid_amd as mg
from conftest import load_case
from helpers import to_np
from oracle.dynobs_oracle import DynObsOracle
from oracle.minigrid_oracle import OracleEnvs
from test_episode_boundary import DYN, cases, gym_id_of

pytestmark = pytest.mark.gpu


def task_ok(task, got, want):
    if task == 1:                         # Fetch: the high byte is the mission template
        return np.array_equal(got & 0xFF, want)
    return task in (0, 11) or np.array_equal(got, want)   # (TwoGoals' word is a running count)


def replay(name, env, z, meta, sel, caller_reset):
    """Steps trace sel[i] on env i.  caller_reset: auto_reset=0 and `reset(mask=done, reseed=False)` after every done (the observation
    of the terminal step is then the terminal one, and reset() returns the new episode's); else the in-kernel reset."""
    K, T = z["actions"].shape
    task, objstate = meta.get("task", 0), meta.get("objstate", False)
    dyn = name.startswith("DynObs-")
    rmap = {(int(k), int(t)): r for r, (k, t) in enumerate(zip(z["reset_k"], z["reset_t"]))}
    n_resets = 0
    for t in range(T):
        obs, rew, done, _ = env.step(z["actions"][sel, t])
        obs, rew, done = to_np(obs).copy(), to_np(rew), to_np(done)
        d = z["done"][sel, t].astype(bool)
        assert np.array_equal(done, z["done"][sel, t]), (name, t)
        assert np.array_equal(rew, z["reward"][sel, t].astype(np.float32)), (name, t)
        rr = np.array([rmap[(int(sel[i]), t)] for i in np.flatnonzero(d)], np.int64)
        want = z["obs"][sel, t].copy()
        if not caller_reset:
            want[d] = z["reset_obs"][rr]
        assert np.array_equal(obs, want), (name, t)
        if not d.any():
            if dyn and t % 5 == 0:
                assert np.array_equal(env.get_state()["grid"], z["grid"][sel, t]), (name, t)
            continue
        if caller_reset:
            robs = to_np(env.reset(mask=done, reseed=False))
            assert np.array_equal(robs[d], z["reset_obs"][rr]), (name, t)
            assert np.array_equal(robs[~d], obs[~d]), (name, t)            # the others keep their last observation
        n_resets += int(d.sum())
        st = env.get_state()
        assert np.array_equal(st["grid"][d], z["reset_grid"][rr]) and np.array_equal(st["agent"][d], z["reset_agent"][rr]), (name, t)
        assert (st["steps"][d] == 0).all() and (st["carry"][d] == (1, 0, 0)).all()
        if task:
            assert task_ok(task, env.get_task()[d], z["reset_task"][rr]), (name, t)
        if objstate:
            assert np.array_equal(env.get_object_state()["contains"][d], z["reset_contains"][rr]), (name, t)
    assert n_resets == int(z["done"][sel].sum())
    try:
        env.sync()
    except (mg.InvalidAction, mg.OutOfBounds):
        assert task == 11                    # TwoGoals: the recorder kept pickup / drop out, the reference's other exceptions not


def check_start(name, env, obs, z, meta, sel):
    assert np.array_equal(to_np(obs), z["init_obs"][sel]), name
    st = env.get_state()
    assert np.array_equal(st["grid"], z["init_grid"][sel]) and np.array_equal(st["agent"], z["init_agent"][sel]), name
    if meta.get("task", 0):
        assert task_ok(meta["task"], env.get_task(), z["init_task"][sel])
    if meta.get("objstate", False):
        assert np.array_equal(env.get_object_state()["contains"], z["init_contains"][sel])


STREAM = [c for c in cases(False) if load_case(c)[0]["W"] * load_case(c)[0]["H"] <= 4096 and not load_case(c)[0]["full_obs"]]


@pytest.mark.parametrize("name", STREAM)
def test_plain_caller_reset_against_reference_traces(name):
    """`seed(s); reset()` once, then the reference's own loop: step, `if done: reset()` -- no seed().  Every level after the first is
    drawn on the GPU from the env's continuing stream (task words and Box.contains planes included)."""
    meta, z = load_case(name)
    K = z["actions"].shape[0]
    N = 64 + K
    sel = np.arange(N) % K
    env = mg.VecMiniGrid(gym_id_of(name, meta), num_envs=N, seeds=z["seed"][sel].astype(np.uint64), auto_reset=False, backend="torch")
    assert bool(env.cfg.object_state) == bool(meta.get("objstate", False)) and (name.startswith("DynObs-") or env.cfg.task_kind == meta.get("task", 0))
    check_start(name, env, env.reset(), z, meta, sel)
    replay(name, env, z, meta, sel, caller_reset=True)
    env.close()


@pytest.mark.parametrize("caller_reset", [False, True])
@pytest.mark.parametrize("name", cases("list"))
def test_seed_schedule_against_reference_wrapper_traces(name, caller_reset):
    """ReseedWrapper(env, seeds=list, seed_idx) recorded through the reference's wrapper class: the in-kernel reset of an auto_reset
    handle, and the caller's `if done: wrapped.reset()` loop, both take the next seed of each env's list."""
    meta, z = load_case(name)
    K = z["actions"].shape[0]
    N = 64 + K
    sel = np.arange(N) % K
    env = mg.VecMiniGrid(gym_id_of(name, meta), num_envs=N, seeds=0, auto_reset=not caller_reset, backend="torch")
    env.set_seed_schedule(z["seed_list"][sel], seed_idx=meta["seed_idx0"])
    with pytest.raises(mg.MgxError):
        env.step(np.zeros(N, np.uint8))                                  # like the wrapper, the schedule does not reset by itself
    check_start(name, env, env.reset(reseed=False), z, meta, sel)
    replay(name, env, z, meta, sel, caller_reset=caller_reset)
    env.close()


def level_tables(env_id, lists):
    """levels[i][j] of env i under list entry j, from the host generator (pinned to the reference by tests/test_levelgen.py)."""
    N, K = lists.shape
    g, a, tk, ct = mg.generate_levels(env_id, lists.reshape(-1).astype(np.uint64), with_task=True, with_contains=True)
    return g.reshape((N, K) + g.shape[1:]), a.reshape(N, K, 3), tk.reshape(N, K), ct.reshape((N, K) + ct.shape[1:])


@pytest.mark.parametrize("env_id,mode,N,K,T", [
    ("MiniGrid-DoorKey-5x5-v0", "partial", 64 * 5 + 3, 3, 520), ("MiniGrid-LavaCrossingS9N1-v0", "partial", 64 * 9 + 17, 4, 200),
    ("MiniGrid-LavaCrossingS9N1-v0", "full", 64 * 3 + 1, 2, 150),            # FullyObs, LDS form
    ("MiniGrid-Empty-Random-6x6-v0", "full", 64 * 4 + 5, 5, 330),            # FullyObs, direct form
    ("MiniGrid-LockedRoom-v0", "partial", 64 * 2 + 9, 3, 400),               # gather form, 19x19
    ("MiniGrid-LockedRoom-v0", "full", 64 * 2 + 9, 2, 390),                  # FullyObs, ragged direct form
    ("MiniGrid-LavaCrossingS11N5-v0", "full", 64 + 3, 3, 150),
    ("MiniGrid-Fetch-8x8-N3-v0", "partial", 64 * 3 + 2, 3, 200), ("MiniGrid-GoToObject-6x6-N2-v0", "partial", 64 * 3, 7, 120),
    ("MiniGrid-TwoGoals-Random-16x16-v0", "partial", 64 * 2 + 2, 2, 150),    # gather form with object state
    ("MiniGrid-ObstructedMaze-1Dlhb-v0", "partial", 64 + 7, 3, 600), ("MiniGrid-ObstructedMaze-2Dlhb-v0", "full", 64 + 2, 2, 590)])
def test_seed_schedule_vs_host_generator_and_oracle(env_id, mode, N, K, T):
    """In-kernel ReseedWrapper on every kernel form: episode k of env i runs on the level of seeds[i][(idx0 + k) % K]."""
    rs = np.random.RandomState(hash(env_id) % 1000 + K)
    lists = rs.randint(0, 2 ** 31, size=(N, K)).astype(np.uint64)
    lists[0] = lists[0, 0]                                               # one env whose list repeats a single seed
    idx0 = K - 1
    cfg = mg.env_config(env_id)
    G, A, TK, CT = level_tables(env_id, lists)
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=0, auto_reset=True, backend="torch", obs_mode=mode)
    env.set_seed_schedule(lists, seed_idx=idx0)
    obs = to_np(env.reset(reseed=False))
    full = mode == "full"
    pick = (lambda o: o[1]) if full else (lambda o: o)
    orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, task=cfg.task_kind)
    ep = np.full(N, idx0, np.int64)
    idx = np.arange(N)
    orc.set_state(G[idx, ep], A[idx, ep])
    orc.task = TK[idx, ep].copy()
    if cfg.object_state:
        orc.set_contains(CT[idx, ep])
    assert np.array_equal(obs, pick(orc.observe(full=full)))
    acts = to_np(env.fill_actions(3, 0, T))
    if cfg.task_kind == 11:                                              # TwoGoals: pickup / drop are the reference's `assert False`
        acts = np.where((acts == 3) | (acts == 4), 6, acts).astype(np.uint8)
    n_done = 0
    for t in range(T):
        obs, rew, done, _ = env.step(acts[t])
        out = orc.step(acts[t], full=full)
        oo, orew, odone = (out[1], out[2], out[3]) if full else out
        d = odone.astype(bool)
        n_done += int(d.sum())
        ep[d] = (ep[d] + 1) % K
        orc.grid0[d], orc.agent0[d] = G[d, ep[d]], A[d, ep[d]]
        if cfg.object_state:
            orc.contains0[d] = CT[d, ep[d]]
        orc.reset_where(odone)
        if cfg.task_kind and cfg.task_kind != 11:                         # (TwoGoals' word is the running goal count: reset_where zeroes it)
            orc.task[d] = TK[d, ep[d]]
        want = np.where(d[:, None, None, None], pick(orc.observe(full=full)), oo)
        assert np.array_equal(to_np(done), odone), t
        assert np.array_equal(to_np(rew), orew.astype(np.float32)), t
        assert np.array_equal(to_np(obs), want), t
    st = env.get_state()
    assert np.array_equal(st["grid"], orc.grid) and np.array_equal(st["agent"], orc.agent) and np.array_equal(st["steps"], orc.steps)
    assert n_done > N // 2 and env.stats()["episodes"] == n_done
    env.close()


@pytest.mark.parametrize("env_id,size,n_obst", [("MiniGrid-Dynamic-Obstacles-Random-5x5-v0", 5, 2), ("MiniGrid-Dynamic-Obstacles-8x8-v0", 8, 4),
                                                 ("MiniGrid-Dynamic-Obstacles-16x16-v0", 16, 8)])
@pytest.mark.parametrize("boundary", ["plain", "schedule-auto", "schedule-caller"])
def test_dynobs_boundaries_vs_restatement(env_id, size, n_obst, boundary):
    """Dynamic-Obstacles on a batch of several tiles: the plain reset() continues each env's stream where ITS walks left it; a seed
    schedule restores obstacle order, RNG block, draw tape and position of the list entry's episode start."""
    N, T, K = 64 * 3 + 9, 90, 3
    seeds = (np.arange(N, dtype=np.uint64) * 7919 + 5) % 100003
    lists = (seeds[:, None] * 3 + np.arange(K, dtype=np.uint64)[None, :]) % 100003
    sched = boundary != "plain"
    orc = DynObsOracle(size, n_obst, "Random" in env_id, seeds, seed_lists=lists if sched else None, seed_idx=1)
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=boundary == "schedule-auto", backend="torch")
    if sched:
        env.set_seed_schedule(lists, seed_idx=1)
        obs = env.reset(reseed=False)
    else:
        obs = env.reset()
    assert np.array_equal(to_np(obs), orc.observe())
    rs = np.random.RandomState(3)
    n_done = 0
    for t in range(T):
        a = rs.randint(0, 3, size=N).astype(np.uint8)
        turn = rs.uniform(size=N) < 0.4
        a[turn] = rs.choice([0, 1, 4, 6, 255], size=N)[turn]
        obs, rew, done, _ = env.step(a)
        oo, orew, odone = orc.step(a)
        d = odone.astype(bool)
        n_done += int(d.sum())
        assert np.array_equal(to_np(done), odone), t
        assert np.array_equal(to_np(rew), orew.astype(np.float32)), t
        if boundary == "schedule-auto":
            orc.reset_where(odone)
            oo = np.where(d[:, None, None, None], orc.observe(), oo)
        assert np.array_equal(to_np(obs), oo), t
        if boundary != "schedule-auto" and d.any():
            # (sometimes one step late, sometimes with an env that is not done: a reset between two walks, a reset right after a reset)
            m = d.copy()
            if t % 3 == 0:
                m |= rs.uniform(size=N) < 0.05
            robs = to_np(env.reset(mask=m.astype(np.uint8), reseed=False))
            orc.reset_where(m, reseed=sched)
            assert np.array_equal(robs[m], orc.observe()[m]), t
            if t % 4 == 1:                                               # twice in a row: nothing drawn in between
                robs = to_np(env.reset(mask=m.astype(np.uint8), reseed=False))
                orc.reset_where(m, reseed=sched)
                assert np.array_equal(robs[m], orc.observe()[m]), t
        assert np.array_equal(env.get_state()["grid"], orc.base.grid), t
    assert n_done > N // 4
    env.close()


def test_plain_reset_crosses_rng_blocks_with_long_dynobs_episodes():
    """Turn-only agents never crash: every episode runs to max_steps and draws several MT19937 blocks; the level the plain reset()
    then draws starts wherever the last walk stopped (a rank in the tape's look-ahead, a finished block, a block regenerated in place)."""
    env_id, size, n_obst, N = "MiniGrid-Dynamic-Obstacles-6x6-v0", 6, 3, 64 + 9
    seeds = np.arange(N, dtype=np.uint64) + 4000
    orc = DynObsOracle(size, n_obst, False, seeds)
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=False, backend="numpy")
    assert np.array_equal(env.reset(), orc.observe())
    rs = np.random.RandomState(1)
    for t in range(144 * 3 + 10):
        a = rs.randint(0, 2, size=N).astype(np.uint8)
        obs, rew, done, _ = env.step(a)
        oo, orew, odone = orc.step(a)
        assert np.array_equal(done, odone) and np.array_equal(obs, oo), t
        # a third of the envs is also reset early, at a step of their own choosing, so that the resets land on every stream position
        m = odone.astype(bool) | ((np.arange(N) % 3 == 0) & ((t + np.arange(N)) % 37 == 0))
        if m.any():
            robs = env.reset(mask=m.astype(np.uint8), reseed=False)
            orc.reset_where(m, reseed=False)
            assert np.array_equal(robs[m], orc.observe()[m]), t
            assert np.array_equal(env.get_state()["grid"], orc.base.grid), t
    env.close()


def test_plain_reset_of_new_level_handles_and_one_level_families():
    """On a new_level_each_episode handle the caller's plain reset() consumes the waiting next level like the in-kernel one; on a family
    without randomness it is a restore."""
    env_id, N, L = "MiniGrid-LavaCrossingS9N1-v0", 64 * 2 + 5, 12
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=50, auto_reset=True, new_level_each_episode=True, backend="numpy")
    env.reset()
    levels = [mg.generate_level_stream(env_id, 50 + i, L) for i in range(N)]
    ep = np.zeros(N, np.int64)
    rs = np.random.RandomState(2)
    for r in range(L - 2):
        m = rs.uniform(size=N) < 0.5
        env.reset(mask=m.astype(np.uint8), reseed=False)
        ep[m] += 1
        st = env.get_state()
        assert np.array_equal(st["grid"], np.stack([levels[i][0][ep[i]] for i in range(N)])), r
        assert np.array_equal(st["agent"], np.stack([levels[i][1][ep[i]] for i in range(N)])), r
    env.close()
    env = mg.VecMiniGrid("MiniGrid-Empty-8x8-v0", num_envs=70, auto_reset=False, backend="numpy")
    o0 = env.reset().copy()
    env.step(np.full(70, 2, np.uint8))
    assert np.array_equal(env.reset(mask=(np.arange(70) % 2).astype(np.uint8), reseed=False)[1::2], o0[1::2])
    env.set_seed_schedule(np.arange(140, dtype=np.uint64).reshape(70, 2))     # accepted, changes nothing: one level for every seed
    assert np.array_equal(env.reset(reseed=False), o0)
    env.close()


def test_boundary_argument_checks():
    env = mg.VecMiniGrid("MiniGrid-DoorKey-5x5-v0", num_envs=10, seeds=3, auto_reset=False, backend="numpy")
    with pytest.raises(mg.MgxError, match="RNG stream"):
        env.reset(reseed=False)                                           # nothing to continue yet
    env.reset()
    with pytest.raises(mg.MgxError):
        env.set_seed_schedule(np.zeros((10, 2), np.uint64), seed_idx=2)   # the wrapper would raise IndexError
    with pytest.raises(mg.MgxError):
        env.set_seed_schedule(np.zeros((10, 300), np.uint64))
    env.set_seed_schedule(np.arange(20, dtype=np.uint64).reshape(10, 2))
    with pytest.raises(mg.MgxError, match="every env"):
        env.reset(mask=np.ones(10, np.uint8), reseed=False)
    env.reset(reseed=False)
    g0 = env.get_state()["grid"].copy()
    assert np.array_equal(g0, mg.generate_levels("MiniGrid-DoorKey-5x5-v0", np.arange(0, 20, 2, dtype=np.uint64))[0])
    env.reset()                                                           # explicit seeds end the schedule ...
    assert np.array_equal(env.get_state()["grid"], mg.generate_levels("MiniGrid-DoorKey-5x5-v0", np.arange(3, 13, dtype=np.uint64))[0])
    env.reset(reseed=False)                                               # ... and the plain reset() continues THAT stream
    assert np.array_equal(env.get_state()["grid"], np.stack([mg.generate_level_stream("MiniGrid-DoorKey-5x5-v0", 3 + i, 2)[0][1] for i in range(10)]))
    env.set_seed_schedule(np.arange(20, dtype=np.uint64).reshape(10, 2))
    env.set_seed_schedule(None)                                           # removed again
    env.step(np.zeros(10, np.uint8))
    env.close()
    env = mg.VecMiniGrid("MiniGrid-DoorKey-5x5-v0", num_envs=10, auto_reset=True, new_level_each_episode=True, backend="numpy")
    with pytest.raises(mg.MgxError):
        env.set_seed_schedule(np.zeros((10, 2), np.uint64))
    env.close()


@pytest.mark.parametrize("name", ["LavaCrossingS9N1-stream", "DoorKey-8x8-stream", "Fetch-8x8-N3", "DynObs-8x8-stream"])
def test_single_env_adapter_follows_the_reference_at_the_boundary(name):
    """gym_minigrid_amd.make(id): `env.seed(s); env.reset()` then the caller loop of run_tests.py:41-68 with its bare `env.reset()`."""
    meta, z = load_case(name)
    gid = gym_id_of(name, meta)
    env = mg.make(gid)
    # MiniGridEnv.__init__ seeds with 1337 and resets once: the first reset() a caller issues draws the SECOND level of that stream
    if not name.startswith("DynObs-"):
        g2, a2 = mg.generate_level_stream(gid, 1337, 2)
        env.reset()
        assert np.array_equal(env.encode_grid(), g2[1]) and env.agent_pos == tuple(a2[1][:2])
    k = 1
    env.seed(int(z["seed"][k]))
    obs = env.reset()
    assert np.array_equal(obs["image"], z["init_obs"][k])
    rmap = {int(t): r for r, (kk, t) in enumerate(zip(z["reset_k"], z["reset_t"])) if int(kk) == k}
    for t in range(min(z["actions"].shape[1], 400)):
        obs, reward, done, info = env.step(int(z["actions"][k, t]))
        assert np.array_equal(obs["image"], z["obs"][k, t]) and done == bool(z["done"][k, t]), (name, t)
        if done:
            obs = env.reset()
            assert np.array_equal(obs["image"], z["reset_obs"][rmap[t]]), (name, t)
            assert np.array_equal(env.encode_grid(), z["reset_grid"][rmap[t]])
    env.close()


def test_single_env_reseed_wrapper():
    from gym_minigrid_amd.compat import ReseedWrapper
    name = "LavaCrossingS9N1-seedlist"
    meta, z = load_case(name)
    k = 2
    env = ReseedWrapper(mg.make(meta["gym_id"]), seeds=z["seed_list"][k], seed_idx=meta["seed_idx0"])
    obs = env.reset()
    assert np.array_equal(obs["image"], z["init_obs"][k])
    rmap = {int(t): r for r, (kk, t) in enumerate(zip(z["reset_k"], z["reset_t"])) if int(kk) == k}
    for t in range(z["actions"].shape[1]):
        obs, reward, done, info = env.step(int(z["actions"][k, t]))
        assert np.array_equal(obs["image"], z["obs"][k, t]) and done == bool(z["done"][k, t])
        if done:
            assert np.array_equal(env.reset()["image"], z["reset_obs"][rmap[t]])
    assert len(rmap) >= 2
    env.close()
