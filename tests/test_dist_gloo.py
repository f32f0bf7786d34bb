"""N>1 path on CPU: world_size 2 over gloo.  Sharding by GLOBAL env index (levels, action stream) makes the
result independent of the number of ranks; the only collectives are the logging gather / all-reduce.
The per-rank simulator here is the CPU oracle standing in for the GPU kernels (no GPU in this container); the
host-side logic under test -- dist.shard, seeds/actions keyed by global index, gather_done_reward,
allreduce_log -- is exactly what bench.py and VecMiniGrid(env_offset=...) use."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gym_minigrid_amd as mg
from gym_minigrid_amd import dist as mdist
from oracle.minigrid_oracle import OracleEnvs

ENV_ID = "MiniGrid-LavaCrossingS9N1-v0"
N, T, SEED = 101, 60, 3


def simulate(offset, count):
    cfg = mg.env_config(ENV_ID)
    idx = np.arange(offset, offset + count)
    grid, agent = mg.generate_levels(ENV_ID, (SEED + idx).astype(np.uint64))
    orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1)
    orc.set_state(grid, agent)
    dones, rewards, episodes, rsum = [], [], 0, 0.0
    for t in range(T):
        a = mg.action_stream(SEED, idx, t)
        obs, rew, done = orc.step(a)
        orc.reset_where(done)
        dones.append(done.copy())
        rewards.append(rew.astype(np.float32))
        episodes += int(done.sum())
        rsum += float(rew.sum())
    return np.stack(dones), np.stack(rewards), episodes, rsum


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = mdist.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    off, cnt = mdist.shard(N, rank, world)
    dones, rewards, episodes, rsum = simulate(off, cnt)
    gd, gr = mdist.gather_done_reward(torch.from_numpy(dones[-1]), torch.from_numpy(rewards[-1]))
    stats = mdist.allreduce_log(torch.tensor([float(episodes), rsum], dtype=torch.float64))
    if rank == 0:
        q.put((gd.numpy(), gr.numpy(), stats.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_covers_range():
    for n in (1, 7, 64, 101, 1 << 20):
        for w in (1, 2, 3, 8):
            parts = [mdist.shard(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (o1, c1), (o2, _) in zip(parts, parts[1:]):
                assert o1 + c1 == o2
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gd, gr, stats = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    dones, rewards, episodes, rsum = simulate(0, N)
    assert np.array_equal(gd, dones[-1]) and np.array_equal(gr, rewards[-1])
    assert stats[0] == episodes and abs(stats[1] - rsum) < 1e-9
    assert episodes > 0
