"""Sharding on the HIP path (BASELINE config 4: LavaCrossingS9N1 sharded over GPUs by global env index).
One GPU, three handles: A = envs [0, N/2) (env_offset 0), B = envs [N/2, N) (env_offset N/2), C = all N envs.
Integer seeds (env i gets seed + GLOBAL index), the synthetic action stream keyed by the GLOBAL index (k_fill_actions),
in-kernel auto-reset: every observation, reward, done flag, the final state and the counters of A || B must equal C's,
byte for byte -- "results do not depend on the number of GPUs" pinned on k_seed / k_levelgen / k_fill_actions / k_step
themselves, not on the CPU oracle standing in for them (tests/test_dist_gloo.py).  The concatenation A || B is also
exactly what dist.gather_done_reward / GatherLogger deliver (rank-major = global env order)."""
import numpy as np
import pytest
import torch

import gym_minigrid_amd as mg
from gym_minigrid_amd import dist as mdist
from helpers import make_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,mode,stream,n_a,n_b,T", [
    ("MiniGrid-LavaCrossingS9N1-v0", "partial", False, 3000, 3000, 400),   # config 4's family, shards not tile-aligned
    ("MiniGrid-LavaCrossingS9N1-v0", "partial", True, 1500, 1501, 400),    # a new level per episode: per-env RNG streams
    ("MiniGrid-DoorKey-8x8-v0", "full", False, 1000, 1090, 660),        # every env times out at 640
    ("MiniGrid-Dynamic-Obstacles-8x8-v0", "partial", False, 700, 640, 120),  # RNG inside step()
])
def test_two_shards_equal_one_handle(env_id, mode, stream, n_a, n_b, T):
    seed, aseed = 7, 3
    N = n_a + n_b
    kw = dict(seeds=seed, obs_mode=mode, auto_reset=True, backend="torch", new_level_each_episode=stream)
    A = mg.VecMiniGrid(env_id, num_envs=n_a, env_offset=0, **kw)
    B = mg.VecMiniGrid(env_id, num_envs=n_b, env_offset=n_a, **kw)
    C = mg.VecMiniGrid(env_id, num_envs=N, env_offset=0, **kw)
    oa, ob, oc = A.reset(), B.reset(), C.reset()
    assert torch.equal(torch.cat([oa, ob]), oc)
    aa, ab, ac = A.fill_actions(aseed, 0, T), B.fill_actions(aseed, 0, T), C.fill_actions(aseed, 0, T)
    assert torch.equal(torch.cat([aa, ab], dim=1), ac)
    assert np.array_equal(ac[:, n_a - 2:n_a + 2].cpu().numpy(),
                          mg.action_stream(aseed, np.arange(n_a - 2, n_a + 2)[None, :], np.arange(T)[:, None]))
    episodes = 0
    for t in range(T):
        oa, ra, da, _ = A.step(aa[t])
        ob, rb, db, _ = B.step(ab[t])
        oc, rc, dc, _ = C.step(ac[t])
        assert torch.equal(torch.cat([oa, ob]), oc), t
        assert torch.equal(torch.cat([ra, rb]), rc) and torch.equal(torch.cat([da, db]), dc), t
        episodes += int(dc.sum())
    assert episodes > N // 4          # the comparison crossed many in-kernel resets
    if "Dynamic" not in env_id:
        sa, sb, sc = A.get_state(), B.get_state(), C.get_state()
        for k in sc:
            assert np.array_equal(np.concatenate([sa[k], sb[k]]), sc[k]), k
    ta, tb, tc = A.stats(), B.stats(), C.stats()
    assert ta["episodes"] + tb["episodes"] == tc["episodes"] == episodes
    assert abs(ta["reward_sum"] + tb["reward_sum"] - tc["reward_sum"]) < 1e-9
    for e in (A, B, C):
        e.sync()
        e.close()


def test_shard_b_matches_the_oracle_directly():
    """A handle with env_offset != 0 against the CPU oracle fed the same global seeds / actions (no handle C involved)."""
    env_id, off, n, T, seed = "MiniGrid-LavaCrossingS9N1-v0", 524288 * 3 + 17, 2000, 200, 5
    cfg = mg.env_config(env_id)
    B = mg.VecMiniGrid(env_id, num_envs=n, env_offset=off, seeds=seed, auto_reset=True, backend="torch")
    obs = B.reset()
    gidx = np.arange(off, off + n)
    grid, agent = mg.generate_levels(env_id, (seed + gidx).astype(np.uint64))
    orc = make_oracle(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, grid, np.zeros(grid.shape[:3], np.uint8), agent)
    assert np.array_equal(obs.cpu().numpy(), orc.observe())
    acts = B.fill_actions(9, 0, T)
    for t in range(T):
        obs, rew, done, _ = B.step(acts[t])
        a = mg.action_stream(9, gidx, t)
        assert np.array_equal(acts[t].cpu().numpy(), a)
        o_obs, o_rew, o_done = orc.step(a)
        orc.reset_where(o_done)
        want = np.where(o_done.astype(bool)[:, None, None, None], orc.observe(), o_obs)
        assert np.array_equal(obs.cpu().numpy(), want), t
        assert np.array_equal(done.cpu().numpy(), o_done) and np.array_equal(rew.cpu().numpy(), o_rew.astype(np.float32))
    B.close()


def test_config4_eight_shards_equal_one_handle():
    """BASELINE configs[3] at its stated size: MiniGrid-LavaCrossingS9N1-v0, 4,194,304 envs as 8 shards of 524,288 (here all on
    one GPU, one handle per shard with its global env_offset) against ONE handle of 4,194,304 envs: the concatenated per-env
    done / reward vectors -- what the RCCL all-gather of the 8-GPU run delivers, rank-major -- and the observations equal the
    single handle's on every step checked, and the counters add up.  (The collective itself needs 8 processes: tests/test_dist_gloo.py
    and tests/test_bench_launcher.py cover its order on CPU.)"""
    env_id, G, n, T = "MiniGrid-LavaCrossingS9N1-v0", 8, 524288, 48
    kw = dict(seeds=0, auto_reset=True, backend="torch")
    shards = [mg.VecMiniGrid(env_id, num_envs=n, env_offset=r * n, **kw) for r in range(G)]
    assert [mdist.shard(G * n, r, G) for r in range(G)] == [(r * n, n) for r in range(G)]
    C = mg.VecMiniGrid(env_id, num_envs=G * n, env_offset=0, **kw)
    oc = C.reset()
    for r, e in enumerate(shards):
        assert torch.equal(e.reset(), oc[r * n:(r + 1) * n])
    # ... and the single 4,194,304-env handle itself against the CPU oracle, so that the comparison above is not HIP against HIP
    # only: 1,600 envs of it (every shard's first and last tile, tile boundaries, a random spread) followed on every step from
    # host-generated levels of their GLOBAL seeds under the GLOBAL action stream -- observation, reward, done, across the resets.
    rs = np.random.RandomState(11)
    edges = np.concatenate([[r * n + k for k in (0, 1, 63, 64, n - 65, n - 64, n - 1)] for r in range(G)])
    sample = np.unique(np.concatenate([edges, rs.randint(0, G * n, size=1600 - len(edges))])).astype(np.int64)
    cfg = mg.env_config(env_id)
    grid, agent = mg.generate_levels(env_id, sample.astype(np.uint64))          # seeds=0: env i is seeded with i
    orc = make_oracle(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, grid, np.zeros(grid.shape[:3], np.uint8), agent)
    sample_d = torch.from_numpy(sample).cuda()
    assert np.array_equal(oc[sample_d].cpu().numpy(), orc.observe())
    acts_c = C.fill_actions(2, 0, T)
    acts = [e.fill_actions(2, 0, T) for e in shards]
    episodes = sampled_episodes = 0
    for t in range(T):
        oc, rc, dc, _ = C.step(acts_c[t])
        outs = [e.step(a[t]) for e, a in zip(shards, acts)]
        assert torch.equal(torch.cat([o[2] for o in outs]), dc) and torch.equal(torch.cat([o[1] for o in outs]), rc), t
        if t % 8 == 7 or t < 2:
            for r, o in enumerate(outs):
                assert torch.equal(o[0], oc[r * n:(r + 1) * n]), (t, r)
        a = mg.action_stream(2, sample, t)
        assert np.array_equal(acts_c[t][sample_d].cpu().numpy(), a)
        o_obs, o_rew, o_done = orc.step(a)
        orc.reset_where(o_done)
        want = np.where(o_done.astype(bool)[:, None, None, None], orc.observe(), o_obs)
        assert np.array_equal(oc[sample_d].cpu().numpy(), want), t
        assert np.array_equal(dc[sample_d].cpu().numpy(), o_done) and np.array_equal(rc[sample_d].cpu().numpy(), o_rew.astype(np.float32)), t
        episodes += int(dc.sum())
        sampled_episodes += int(o_done.sum())
    assert episodes > 100000 and sampled_episodes > 100 and len(sample) >= 1500
    st = [e.stats() for e in shards]
    assert sum(s["episodes"] for s in st) == C.stats()["episodes"] == episodes
    assert abs(sum(s["reward_sum"] for s in st) - C.stats()["reward_sum"]) < 1e-6
    for e in shards + [C]:
        e.close()
