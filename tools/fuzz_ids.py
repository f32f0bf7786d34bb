#!/usr/bin/env python3
"""Every built-in env id on the GPU box (not part of the test suite): levels generated on the device from random 64-bit
seeds, a random batch size and obs mode, stepped with (a) in-kernel auto-reset or (b) the caller's `reset(mask=done)`
loop, every byte against the CPU oracle on the host-generated levels of the same seeds.

    python tools/fuzz_ids.py [rounds] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import gym_minigrid_amd as mg  # noqa: E402
from gym_minigrid_amd import _lib  # noqa: E402
from oracle.minigrid_oracle import OracleEnvs  # noqa: E402  (checker only)
from oracle.dynobs_oracle import DynObsOracle  # noqa: E402


def np_(x):
    return x if isinstance(x, np.ndarray) else x.cpu().numpy()


def one(env_id, rs):
    cfg = mg.env_config(env_id)
    dyn = cfg.task_kind == _lib.TASK_DYNOBS
    cells = cfg.width * cfg.height
    N = int(rs.choice([1, 63, 65, 300, 1500])) if cells <= 400 else int(rs.choice([1, 65, 200]))
    if dyn:
        N = min(N, 300)  # the restatement of the obstacle walk is a Python loop per env and obstacle
    T = 120
    full = bool(rs.randint(2))
    caller_reset = bool(rs.randint(2))
    small = rs.uniform() < 0.5   # small seeds collide across envs (same level in several envs), large ones use all 64 bits
    seeds = rs.randint(0, 50, size=N).astype(np.uint64) if small else rs.randint(0, 2 ** 63, size=N, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(rs.randint(2))
    print("      %s N=%d full=%d caller_reset=%d small=%d" % (env_id, N, full, caller_reset, small), flush=True) if os.environ.get("FUZZ_VERBOSE") else None
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=not caller_reset, backend="torch", obs_mode="full" if full else "partial")
    obs = np_(env.reset())
    if dyn:
        orc = DynObsOracle(cfg.width, cfg.level_arg0, "Random" in env_id, seeds)
        observe = lambda: orc.base.observe(True)[int(full)]  # noqa: E731
    else:
        grid, agent, task, contains = mg.generate_levels(env_id, seeds, with_task=True, with_contains=True)
        orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, task=cfg.task_kind)
        orc.set_state(grid, agent)
        orc.task = task.copy()
        if cfg.object_state:           # ObstructedMaze: keys hidden in boxes
            orc.set_contains(contains)
        observe = lambda: orc.observe(True)[int(full)]  # noqa: E731
    assert np.array_equal(obs, observe()), (env_id, "reset")
    nact = 3 if dyn else 7
    faults = 0
    for t in range(T):
        a = rs.randint(0, nact, size=N).astype(np.uint8)
        if rs.uniform() < 0.5:
            a[rs.uniform(size=N) < 0.5] = 2          # forward-heavy: reach things
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        if dyn:
            oo, orew, odone = orc.step(a)
            if full:
                oo = observe()   # the observation is a function of the state the step left
        else:
            o1, o2, orew, odone = orc.step(a, True)
            faults += int((orc.err != 0).sum())  # TwoGoals: pickup / drop are 'unknown action' in the fork's step (envs/twogoals.py)
            oo = o2 if full else o1
        d = odone.astype(bool)
        assert np.array_equal(np_(done), odone), (env_id, t)
        assert np.array_equal(np_(rew), orew.astype(np.float32)), (env_id, t)
        if caller_reset:
            assert np.array_equal(np_(obs), oo), (env_id, t)           # the terminal observation, as the reference returns it
            if d.any():
                obs = env.reset(mask=done)
                orc.reset_where(odone)
                assert np.array_equal(np_(obs)[d], observe()[d]), (env_id, t, "masked reset")
        else:
            orc.reset_where(odone)
            if d.any():
                oo = oo.copy()
                oo[d] = observe()[d]
            assert np.array_equal(np_(obs), oo), (env_id, t)
    st = env.get_state()
    base = orc.base if dyn else orc
    assert np.array_equal(st["grid"], base.grid) and np.array_equal(st["agent"], base.agent), env_id
    try:
        env.sync()
        assert faults == 0, (env_id, faults)
    except (mg.InvalidAction, mg.OutOfBounds):
        assert faults > 0, env_id
        env.clear_faults()
    env.close()
    return N * T, dict(N=N, full=full, caller_reset=caller_reset, small_seeds=bool(small))


def one_stream(env_id, rs):
    """new_level_each_episode: every env follows its own level stream; level k of env i == host generate_level_stream(seed_i)[k]."""
    cfg = mg.env_config(env_id)
    N = int(rs.choice([1, 63, 65, 200]))
    T, L = 100, 60
    full = bool(rs.randint(2))
    seeds = rs.randint(0, 2 ** 63, size=N, dtype=np.int64).astype(np.uint64) if rs.uniform() < 0.5 else rs.randint(0, 30, size=N).astype(np.uint64)
    try:
        env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=True, new_level_each_episode=True, backend="torch", obs_mode="full" if full else "partial")
    except mg.MgxError:
        return 0, None          # Dynamic-Obstacles: the flag is refused (the obstacle walk shares the stream)
    obs = np_(env.reset())
    levels = [mg.generate_level_stream(env_id, int(sd), L, with_task=True, with_contains=True) for sd in seeds]
    G, A, K, C = (np.stack([lv[j] for lv in levels]) for j in range(4))
    orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, task=cfg.task_kind)
    orc.set_state(G[:, 0], A[:, 0])
    orc.task = K[:, 0].copy()
    if cfg.object_state:
        orc.set_contains(C[:, 0])
    observe = lambda: orc.observe(True)[int(full)]  # noqa: E731
    assert np.array_equal(obs, observe()), (env_id, "stream reset")
    ep = np.zeros(N, np.int64)
    for t in range(T):
        a = rs.choice([0, 1, 2, 2, 2, 3, 4, 5, 6], size=N).astype(np.uint8)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        o1, o2, orew, odone = orc.step(a, True)
        oo = (o2 if full else o1).copy()
        d = odone.astype(bool)
        ep[d] += 1
        if ep.max() >= L:
            break
        orc.grid0[d], orc.agent0[d] = G[d, ep[d]], A[d, ep[d]]
        if cfg.object_state:
            orc.contains0[d] = C[d, ep[d]]
        orc.reset_where(odone)
        orc.task[d] = K[d, ep[d]]
        oo[d] = observe()[d]
        assert np.array_equal(np_(done), odone), (env_id, t)
        assert np.array_equal(np_(rew), orew.astype(np.float32)), (env_id, t)
        assert np.array_equal(np_(obs), oo), (env_id, t, "stream")
        if t % 25 == 24:
            assert not cfg.task_kind or np.array_equal(env.get_task(), orc.task), (env_id, t, "task words")
    env.clear_faults()
    env.close()
    return N * T, dict(N=N, full=full, stream=True)


def one_epilogue(env_id, rs):
    """The one-hot / flat observation modes against the wrappers' formulas applied to a twin env's (type, color, state)
    images (same seeds, same actions), with in-kernel auto-reset or the caller's reset(mask=done)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import onehot
    from oracle.minigrid_oracle import flat_obs
    mode = str(rs.choice(["partial_onehot", "full_onehot", "full_onehot_nocolor", "flat", "full_flat"]))
    base = "partial" if mode in ("partial_onehot", "flat") else "full"
    N = int(rs.choice([1, 63, 65, 200]))
    caller_reset = bool(rs.randint(2))
    seeds = rs.randint(0, 40, size=N).astype(np.uint64)
    a_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=not caller_reset, backend="torch", obs_mode=base)
    b_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=not caller_reset, backend="torch", obs_mode=mode)

    def expand(img):
        if mode == "partial_onehot":
            return onehot(img, 7, 3)
        if mode == "full_onehot":
            return onehot(img, 7, 4)
        if mode == "full_onehot_nocolor":
            return onehot(img, 0, 4)
        ms = a_env.missions()
        return np.stack([flat_obs(img[i], ms[i]) for i in range(N)])

    oa, ob = np_(a_env.reset()), np_(b_env.reset())
    assert np.array_equal(expand(oa), ob), (env_id, mode, "reset")
    dyn = mg.env_config(env_id).task_kind == _lib.TASK_DYNOBS
    for t in range(40):
        a = torch.from_numpy(rs.choice([0, 1, 2, 2, 2, 5, 6] if not dyn else [0, 1, 2, 2], size=N).astype(np.uint8)).cuda()
        oa, ra, da, _ = a_env.step(a)
        ob, rb, db, _ = b_env.step(a)
        assert np.array_equal(np_(da), np_(db)) and np.array_equal(np_(ra), np_(rb)), (env_id, mode, t)
        assert np.array_equal(expand(np_(oa)), np_(ob)), (env_id, mode, t)
        if caller_reset and np_(da).any():
            oa, ob = a_env.reset(mask=da), b_env.reset(mask=db)
            assert np.array_equal(expand(np_(oa)), np_(ob)), (env_id, mode, t, "masked reset")
    for e in (a_env, b_env):
        e.clear_faults()
        e.close()
    return N * 40, dict(N=N, mode=mode, caller_reset=caller_reset)


def one_options(env_id, rs):
    """ViewSizeWrapper / extended_actions / default_vis=False / object_state on the built-in ids: the task rules under
    the run-time-size kernels."""
    cfg = mg.env_config(env_id)
    if cfg.task_kind == _lib.TASK_DYNOBS:
        return 0, None
    view = int(rs.choice([3, 5, 7, 9, 11]))
    ext, alt = bool(rs.randint(2)), bool(rs.randint(2))
    objstate = bool(cfg.object_state) or rs.uniform() < 0.25   # (ObstructedMaze's boxed keys need the plane)
    full = rs.uniform() < 0.3
    N = int(rs.choice([1, 65, 400]))
    seeds = rs.randint(0, 2 ** 40, size=N).astype(np.uint64)
    if os.environ.get("FUZZ_VERBOSE"):
        print("      options %s N=%d view=%d ext=%d alt=%d objstate=%d full=%d" % (env_id, N, view, ext, alt, objstate, full), flush=True)
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=True, backend="torch", obs_mode="full" if full else "partial",
                         agent_view_size=view, extended_actions=ext, default_vis=not alt, object_state=objstate)
    obs = np_(env.reset())
    grid, agent, task, contains = mg.generate_levels(env_id, seeds, with_task=True, with_contains=True)
    orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, view=view, extended=ext, alt_vis=alt, task=cfg.task_kind)
    orc.set_state(grid, agent)
    orc.task = task.copy()
    if objstate:
        orc.set_contains(contains)    # empty everywhere but in ObstructedMaze's boxes
    observe = lambda: orc.observe(True)[int(full)]  # noqa: E731
    assert np.array_equal(obs, observe()), (env_id, "options reset")
    nact = 9 if ext else 7
    for t in range(80):
        a = rs.randint(0, nact, size=N).astype(np.uint8)
        a[rs.uniform(size=N) < 0.3] = 2
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        o1, o2, orew, odone = orc.step(a, True)
        oo = (o2 if full else o1).copy()
        d = odone.astype(bool)
        orc.reset_where(odone)
        oo[d] = observe()[d]
        assert np.array_equal(np_(done), odone), (env_id, t, view, ext, alt, objstate)
        assert np.array_equal(np_(rew), orew.astype(np.float32)), (env_id, t, view, ext, alt, objstate)
        assert np.array_equal(np_(obs), oo), (env_id, t, view, ext, alt, objstate)
    st = env.get_state()
    assert np.array_equal(st["grid"], orc.grid) and np.array_equal(st["agent"], orc.agent) and np.array_equal(st["aux"], orc.aux), env_id
    env.clear_faults()
    env.close()
    return N * 80, dict(N=N, view=view, ext=ext, alt=alt, objstate=bool(objstate), full=bool(full))


def one_rollout(env_id, rs):
    """rollout(T) (one hipGraph launch, captured once and replayed) against the same steps taken one mgx_step at a time on a twin env."""
    N = int(rs.choice([64, 192, 1024]))
    T = int(rs.choice([1, 7, 32]))
    mode = str(rs.choice(["partial", "full", "partial_onehot", "flat"]))
    stream = bool(rs.randint(2))
    seeds = rs.randint(0, 1000, size=N).astype(np.uint64)
    kw = dict(num_envs=N, seeds=seeds, auto_reset=True, backend="torch", obs_mode=mode)
    try:
        a_env = mg.VecMiniGrid(env_id, new_level_each_episode=stream, **kw)
    except mg.MgxError:
        stream = False
        a_env = mg.VecMiniGrid(env_id, **kw)
    b_env = mg.VecMiniGrid(env_id, new_level_each_episode=stream, **kw)
    assert torch.equal(a_env.reset(), b_env.reset())
    dyn = mg.env_config(env_id).task_kind == _lib.TASK_DYNOBS
    acts = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    for rep in range(3):          # capture, then two replays
        acts.copy_(torch.from_numpy(rs.choice([0, 1, 2, 2, 2, 5, 6] if not dyn else [0, 1, 2, 2], size=(T, N)).astype(np.uint8)))
        obs, rew, done = a_env.rollout(acts)
        for t in range(T):
            o, r, d, _ = b_env.step(acts[t])
            assert torch.equal(obs[t], o) and torch.equal(rew[t], r) and torch.equal(done[t], d), (env_id, mode, stream, rep, t)
    sa, sb = a_env.stats(), b_env.stats()
    assert sa == sb, (env_id, sa, sb)
    for e in (a_env, b_env):
        e.clear_faults()
        e.close()
    return N * T * 3, dict(N=N, T=T, mode=mode, stream=stream, rollout=True)


def one_boundary(env_id, rs):
    """The other two episode boundaries of the reference (round 4): the plain caller-side reset() -- the env's RNG stream continues, level k of env i
    is host generate_level_stream(seed_i)[k] -- and ReseedWrapper with a list of K seeds per env (mgx_set_seed_schedule), in-kernel or caller-side."""
    cfg = mg.env_config(env_id)
    dyn = cfg.task_kind == _lib.TASK_DYNOBS
    kind = str(rs.choice(["plain", "schedule-auto", "schedule-caller"]))
    full = bool(rs.randint(2))
    N = int(rs.choice([1, 63, 65, 200])) if not dyn else int(rs.choice([1, 65, 130]))
    T, L = 100, 50
    seeds = rs.randint(0, 2 ** 63, size=N, dtype=np.int64).astype(np.uint64) if rs.uniform() < 0.5 else rs.randint(0, 30, size=N).astype(np.uint64)
    K = int(rs.randint(1, 6))
    idx0 = int(rs.randint(K))
    lists = rs.randint(0, 40, size=(N, K)).astype(np.uint64) if rs.uniform() < 0.5 else rs.randint(0, 2 ** 62, size=(N, K), dtype=np.int64).astype(np.uint64)
    sched = kind != "plain"
    try:
        env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=kind == "schedule-auto", backend="torch", obs_mode="full" if full else "partial")
        if sched:
            env.set_seed_schedule(lists, seed_idx=idx0)
            obs = np_(env.reset(reseed=False))
        else:
            obs = np_(env.reset())
    except mg.MgxError as e:
        assert cfg.width * cfg.height > 4096, (env_id, e)     # host-generated grids keep no RNG stream on the device
        return 0, None
    idx = np.arange(N)
    ep = np.full(N, idx0 if sched else 0, np.int64)
    if dyn:
        orc = DynObsOracle(cfg.width, cfg.level_arg0, "Random" in env_id, seeds, seed_lists=lists if sched else None, seed_idx=idx0)
        observe = lambda: orc.base.observe(True)[int(full)]  # noqa: E731
    else:
        if sched:
            g, a, tk, ct = mg.generate_levels(env_id, lists.reshape(-1), with_task=True, with_contains=True)
            G, A, TK, C = g.reshape((N, K) + g.shape[1:]), a.reshape(N, K, 3), tk.reshape(N, K), ct.reshape((N, K) + ct.shape[1:])
        else:
            levels = [mg.generate_level_stream(env_id, int(sd), L, with_task=True, with_contains=True) for sd in seeds]
            G, A, TK, C = (np.stack([lv[j] for lv in levels]) for j in range(4))
        orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, task=cfg.task_kind)
        orc.set_state(G[idx, ep], A[idx, ep])
        orc.task = TK[idx, ep].copy()
        if cfg.object_state:
            orc.set_contains(C[idx, ep])
        observe = lambda: orc.observe(True)[int(full)]  # noqa: E731
    assert np.array_equal(obs, observe()), (env_id, kind, "start")
    nact = 3 if dyn else 7
    for t in range(T):
        a = rs.randint(0, nact, size=N).astype(np.uint8)
        if rs.uniform() < 0.5:
            a[rs.uniform(size=N) < 0.5] = 2
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        if dyn:
            oo, orew, odone = orc.step(a)
            if full:
                oo = observe()
        else:
            o1, o2, orew, odone = orc.step(a, True)
            oo = (o2 if full else o1).copy()
        d = odone.astype(bool)
        assert np.array_equal(np_(done), odone), (env_id, kind, t)
        assert np.array_equal(np_(rew), orew.astype(np.float32)), (env_id, kind, t)
        if kind != "schedule-auto":
            assert np.array_equal(np_(obs), oo), (env_id, kind, t)       # the terminal observation, as the reference returns it
        if d.any():
            ep[d] = (ep[d] + 1) % K if sched else ep[d] + 1
            if not sched and ep.max() >= L:
                break
            if dyn:
                orc.reset_where(odone, reseed=sched)
            else:
                orc.grid0[d], orc.agent0[d] = G[d, ep[d]], A[d, ep[d]]
                if cfg.object_state:
                    orc.contains0[d] = C[d, ep[d]]
                orc.reset_where(odone)
                if cfg.task_kind and cfg.task_kind != _lib.TASK_TWOGOALS:
                    orc.task[d] = TK[d, ep[d]]
            if kind == "schedule-auto":
                oo = oo.copy()
                oo[d] = observe()[d]
            else:
                robs = np_(env.reset(mask=done, reseed=False))
                assert np.array_equal(robs[d], observe()[d]), (env_id, kind, t, "reset")
        if kind == "schedule-auto":
            assert np.array_equal(np_(obs), oo), (env_id, kind, t)
    st = env.get_state()
    base = orc.base if dyn else orc
    assert np.array_equal(st["grid"], base.grid) and np.array_equal(st["agent"], base.agent), (env_id, kind)
    env.clear_faults()
    env.close()
    return N * T, dict(N=N, boundary=kind, K=K if sched else 0, full=full)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    total, t0 = 0, time.perf_counter()
    ids = mg.env_ids()
    for r in range(rounds):
        for env_id in ids:
            n, desc = one(env_id, rs)
            total += n
            print("round %d %-46s ok  %s  (%.0f s)" % (r, env_id, desc, time.perf_counter() - t0), flush=True)
            n, desc = one_stream(env_id, rs)
            total += n
            if desc:
                print("round %d %-46s ok  %s  (%.0f s)" % (r, env_id, desc, time.perf_counter() - t0), flush=True)
            n, desc = one_epilogue(env_id, rs)
            total += n
            print("round %d %-46s ok  %s  (%.0f s)" % (r, env_id, desc, time.perf_counter() - t0), flush=True)
            n, desc = one_rollout(env_id, rs)
            total += n
            print("round %d %-46s ok  %s  (%.0f s)" % (r, env_id, desc, time.perf_counter() - t0), flush=True)
            n, desc = one_options(env_id, rs)
            total += n
            if desc:
                print("round %d %-46s ok  %s  (%.0f s)" % (r, env_id, desc, time.perf_counter() - t0), flush=True)
            n, desc = one_boundary(env_id, rs)
            total += n
            if desc:
                print("round %d %-46s ok  %s  (%.0f s)" % (r, env_id, desc, time.perf_counter() - t0), flush=True)
    print("fuzz_ids ok: %d ids x %d rounds, %d env-steps, every byte equal" % (len(ids), rounds, total), flush=True)


if __name__ == "__main__":
    main()
