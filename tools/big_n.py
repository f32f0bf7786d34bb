#!/usr/bin/env python3
"""Index-width check on the GPU box: batches whose buffers pass 2^31 / 2^32 bytes (16 Mi envs: 2.4 GB of observations;
one-hot at 5 Mi envs: 5.4 GB), compared against the CPU oracle / the wrapper formula on the LAST envs of the batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gym_minigrid_amd as mg  # noqa: E402
from oracle.minigrid_oracle import OracleEnvs  # noqa: E402
from helpers import onehot  # noqa: E402

TAIL = 4096
for env_id, N, mode in (("MiniGrid-DoorKey-8x8-v0", 1 << 24, "partial"), ("MiniGrid-LavaCrossingS9N1-v0", 5 << 20, "partial_onehot"),
                        ("MiniGrid-Empty-16x16-v0", 6 << 20, "full")):
    cfg = mg.env_config(env_id)
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=7, auto_reset=True, backend="torch", obs_mode=mode)
    obs = env.reset()
    print(env_id, N, mode, "obs bytes %.2f GB" % (obs.numel() * obs.element_size() / 1e9), flush=True)
    seeds = (np.uint64(7) + np.arange(N - TAIL, N, dtype=np.uint64))
    grid, agent = mg.generate_levels(env_id, seeds)
    orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1)
    orc.set_state(grid, agent)
    full = mode == "full"
    expand = (lambda x: onehot(x, 7, 3)) if mode == "partial_onehot" else (lambda x: x)
    assert np.array_equal(obs[N - TAIL:].cpu().numpy(), expand(orc.observe(True)[int(full)]))
    for t in range(24):
        a = env.fill_actions(3, t, 1)[0]
        obs, rew, done, _ = env.step(a)
        at = a[N - TAIL:].cpu().numpy()
        o1, o2, orew, odone = orc.step(at, True)
        oo = (o2 if full else o1).copy()
        orc.reset_where(odone)
        d = odone.astype(bool)
        oo[d] = orc.observe(True)[int(full)][d]
        assert np.array_equal(done[N - TAIL:].cpu().numpy(), odone), t
        assert np.array_equal(rew[N - TAIL:].cpu().numpy(), orew.astype(np.float32)), t
        assert np.array_equal(obs[N - TAIL:].cpu().numpy(), expand(oo)), t
    s = env.stats()
    assert s["steps"] == N * 24, s
    env.close()
    del env, obs
    torch.cuda.empty_cache()
    print("   last %d envs equal over 24 steps; stats %s" % (TAIL, s), flush=True)
print("big_n ok")
