#!/usr/bin/env python3
"""The reference's own episode handling, vectorised: `if done: obs = env.reset()` (run_tests.py:64-66) becomes
`env.reset(mask=done)`.  With auto_reset=False nothing is reset inside the step kernel; an env that keeps its seed is
restored from its episode-start snapshot, one whose seed changes is re-seeded and regenerated on the GPU.

    python examples/caller_side_reset.py [env_id] [num_envs] [steps]
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
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    env = mg.VecMiniGrid(env_id, num_envs=n, seeds=np.arange(n, dtype=np.uint64), auto_reset=False)
    obs = env.reset()
    episodes = 0
    t0 = time.perf_counter()
    for t in range(steps):
        actions = torch.randint(0, env.action_space.n, (n,), device=obs.device, dtype=torch.uint8)  # your policy here
        obs, reward, done, info = env.step(actions)           # the terminal observation is still in obs here
        if t % 50 == 49:
            episodes += int(done.sum())                       # (a host sync: only every 50 steps in this example)
        obs = env.reset(mask=done)                            # same buffer: only the tiles with a finished env change
    env.sync()
    dt = time.perf_counter() - t0
    print("%s: %d env-steps in %.3f s = %.2f G steps/s (sampled dones: %d)" % (env_id, n * steps, dt, n * steps / dt / 1e9, episodes))
    env.close()


if __name__ == "__main__":
    main()
