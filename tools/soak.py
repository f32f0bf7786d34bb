#!/usr/bin/env python3
"""Long parity soak on the GPU box (not part of the test suite): every family, N envs x T steps with in-kernel auto-reset,
every observation / reward / done byte against the CPU oracle running the same rule on host-generated levels.

    python tools/soak.py [N] [T] [partial|full]
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
from oracle.minigrid_oracle import OracleEnvs  # noqa: E402  (checker only)

FAMILIES = ["MiniGrid-Empty-8x8-v0", "MiniGrid-DoorKey-8x8-v0", "MiniGrid-LavaCrossingS9N1-v0", "MiniGrid-SimpleCrossingS11N5-v0",
            "MiniGrid-LavaGapS7-v1", "MiniGrid-MultiRoom-N4-S5-v0", "MiniGrid-FourRooms-v0", "MiniGrid-Fetch-8x8-N3-v0",
            "MiniGrid-GoToDoor-8x8-v0", "MiniGrid-GoToObject-8x8-N2-v0", "MiniGrid-PutNear-8x8-N3-v0", "MiniGrid-RedBlueDoors-8x8-v0",
            "MiniGrid-MemoryS13Random-v0", "MiniGrid-UnlockPickup-v0", "MiniGrid-BlockedUnlockPickup-v0", "MiniGrid-KeyCorridorS4R3-v0",
            "MiniGrid-LockedRoom-v0", "MiniGrid-Playground-v0", "MiniGrid-TwoGoals-8x8-v0", "MiniGrid-TwoGoals-Random-16x16-v0",
            "MiniGrid-ObstructedMaze-1Dlhb-v0", "MiniGrid-ObstructedMaze-2Dlh-v0", "MiniGrid-ObstructedMaze-Full-v0"]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    full = len(sys.argv) > 3 and sys.argv[3] == "full"  # FullyObsWrapper kernels (direct / ragged / LDS form by grid size)
    total = 0
    for env_id in FAMILIES:
        t0 = time.perf_counter()
        seeds = np.arange(N, dtype=np.uint64) * 11 + 3
        env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=True, backend="torch", obs_mode="full" if full else "partial")
        obs = env.reset().cpu().numpy()
        grid, agent, task, contains = mg.generate_levels(env_id, seeds, with_task=True, with_contains=True)
        cfg = mg.env_config(env_id)
        orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, task=cfg.task_kind)
        orc.set_state(grid, agent)
        orc.task = task.copy()
        if cfg.object_state:          # ObstructedMaze: keys hidden in boxes
            orc.set_contains(contains)
        assert np.array_equal(obs, orc.observe(True)[int(full)]), env_id
        rs = np.random.RandomState(5)
        episodes = 0
        for t in range(T):
            a = rs.randint(0, 7, size=N).astype(np.uint8)
            obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
            oo, of, orew, odone = orc.step(a, True)
            oo = of if full else oo
            orc.reset_where(odone)
            if odone.any():
                oo[odone.astype(bool)] = orc.observe(True)[int(full)][odone.astype(bool)]
            assert np.array_equal(done.cpu().numpy(), odone), (env_id, t)
            assert np.array_equal(rew.cpu().numpy(), orew.astype(np.float32)), (env_id, t)
            assert np.array_equal(obs.cpu().numpy(), oo), (env_id, t)
            episodes += int(odone.sum())
        st = env.get_state()
        assert np.array_equal(st["grid"], orc.grid) and np.array_equal(st["agent"], orc.agent)
        assert env.stats()["episodes"] == episodes
        env.close()
        total += N * T
        print("%-40s %d envs x %d steps, %8d episodes: every byte equal  (%.1f s)" % (env_id, N, T, episodes, time.perf_counter() - t0), flush=True)
    print("soak ok (%s obs): %d env-steps compared" % ("full" if full else "partial", total), flush=True)


if __name__ == "__main__":
    main()
