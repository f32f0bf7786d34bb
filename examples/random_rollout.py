#!/usr/bin/env python3
"""Minimal use of the batched env: the reference's caller loop (run_tests.py:41-68, benchmark.py:45-46) for N envs.

    python examples/random_rollout.py [env_id] [num_envs] [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))

import torch  # noqa: E402  (first: libmgx shares torch's HIP runtime)
import gym_minigrid_amd as mg  # noqa: E402


def main():
    env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniGrid-DoorKey-8x8-v0"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    env = mg.VecMiniGrid(env_id, num_envs=n, seeds=0, new_level_each_episode=True)   # plain gym semantics: new level per episode
    obs = env.reset()                                        # uint8 (N, 7, 7, 3) on cuda:0
    print(env_id, "obs", tuple(obs.shape), obs.dtype, "| mission:", env.mission)
    t0 = time.perf_counter()
    for _ in range(steps):
        actions = torch.randint(0, env.action_space.n, (n,), device=obs.device, dtype=torch.uint8)  # your policy here
        obs, reward, done, info = env.step(actions)           # auto-reset: obs of a finished env is its next episode's first
    env.sync()
    dt = time.perf_counter() - t0
    s = env.stats()
    print("%d env-steps in %.3f s = %.2f G steps/s; %d episodes, reward sum %.1f" % (n * steps, dt, n * steps / dt / 1e9, s["episodes"], s["reward_sum"]))
    env.close()


if __name__ == "__main__":
    main()
