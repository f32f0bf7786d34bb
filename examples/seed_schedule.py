#!/usr/bin/env python3
"""The reference's ReseedWrapper with a LIST of seeds (wrappers.py:12-28), batched: every env cycles through its own K seeds, one per
episode.  The K levels of every env are generated once on the GPU and stay resident in HBM (K episode-start snapshots per env), so the
reset inside the step kernel is a copy whatever K is.

    python examples/seed_schedule.py [env_id] [num_envs] [K] [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: libmgx shares torch's HIP runtime)
import gym_minigrid_amd as mg  # noqa: E402


def main():
    env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniGrid-LavaCrossingS9N1-v0"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    env = mg.VecMiniGrid(env_id, num_envs=n, auto_reset=True)
    seeds = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(K) + np.arange(K, dtype=np.uint64)[None, :])  # env i: seeds i*K .. i*K + K-1
    t0 = time.perf_counter()
    env.set_seed_schedule(seeds, seed_idx=0)        # ReseedWrapper(env_i, seeds=seeds[i], seed_idx=0): K x (seed + generate) on the GPU
    obs = env.reset(reseed=False)                   # the wrapper's first reset(): every env starts on seeds[i][0]
    env.sync()
    print("schedule of %d seeds per env installed in %.1f ms" % (K, (time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter()
    for t in range(steps):
        actions = torch.randint(0, env.action_space.n, (n,), device=obs.device, dtype=torch.uint8)  # your policy here
        obs, reward, done, info = env.step(actions)  # a finished env moves on to the next seed of its list inside the kernel
    env.sync()
    dt = time.perf_counter() - t0
    print("%s: %d env-steps in %.3f s = %.2f G steps/s, %d episodes" % (env_id, n * steps, dt, n * steps / dt / 1e9, env.stats()["episodes"]))
    env.close()


if __name__ == "__main__":
    main()
