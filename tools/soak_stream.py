#!/usr/bin/env python3
"""Parity soak of new_level_each_episode on the GPU box (not part of the test suite): N envs x T steps per family, every env on its own level
stream -- level k of env i == host generate_level_stream(seed_i)[k], task word included -- every observation / reward / done byte against the CPU
oracle, with the level generator in its default form (the ring of next-level buffers beside the steps, MultiRoom: one buffer).

    python tools/soak_stream.py [N] [scale]
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

# env id, steps, levels per env the host generates up front
CASES = [("MiniGrid-LavaCrossingS9N1-v0", 600, 220), ("MiniGrid-LavaCrossingS9N3-v0", 400, 220), ("MiniGrid-LavaGapS7-v1", 900, 20),
         ("MiniGrid-DoorKey-5x5-v0", 1200, 14), ("MiniGrid-DoorKey-8x8-v0", 1400, 6), ("MiniGrid-Empty-Random-6x6-v0", 900, 14),
         ("MiniGrid-SimpleCrossingS11N5-v0", 1500, 8), ("MiniGrid-Fetch-8x8-N3-v0", 600, 60), ("MiniGrid-GoToDoor-8x8-v0", 500, 60),
         ("MiniGrid-GoToObject-6x6-N2-v0", 200, 200), ("MiniGrid-PutNear-6x6-N2-v0", 300, 120), ("MiniGrid-Unlock-v0", 900, 30),
         ("MiniGrid-UnlockPickup-v0", 900, 8), ("MiniGrid-RedBlueDoors-8x8-v0", 1300, 14), ("MiniGrid-MemoryS13Random-v0", 500, 50),
         ("MiniGrid-KeyCorridorS3R2-v0", 900, 10), ("MiniGrid-LockedRoom-v0", 600, 8), ("MiniGrid-Playground-v0", 500, 10),
         ("MiniGrid-FourRooms-v0", 1100, 8), ("MiniGrid-MultiRoom-N4-S5-v0", 500, 12), ("MiniGrid-MultiRoom-N2-S4-v0", 300, 14)]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    total = 0
    for env_id, T, L in CASES:
        T = max(8, int(T * scale))
        L = int(L * max(1.0, scale) * 1.3) + 4
        t0 = time.perf_counter()
        seed = 1234
        cfg = mg.env_config(env_id)
        env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seed, auto_reset=True, new_level_each_episode=True, backend="torch")
        obs = env.reset().cpu().numpy()
        levels = [mg.generate_level_stream(env_id, seed + i, L, with_task=True) for i in range(N)]
        G = np.stack([lv[0] for lv in levels]); A = np.stack([lv[1] for lv in levels]); K = np.stack([lv[2] for lv in levels])
        ep = np.zeros(N, np.int64)
        orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, task=cfg.task_kind)
        orc.set_state(G[:, 0], A[:, 0])
        orc.task = K[:, 0].copy()
        assert np.array_equal(obs, orc.observe()), env_id
        acts = env.fill_actions(21, 0, T).cpu().numpy()
        for t in range(T):
            o, r, d, _ = env.step(acts[t])
            oo, orew, odone = orc.step(acts[t])
            dn = odone.astype(bool)
            ep[dn] += 1
            assert ep.max() < L, (env_id, "raise L", t)
            orc.grid0[dn], orc.agent0[dn] = G[dn, ep[dn]], A[dn, ep[dn]]
            orc.task[dn] = K[dn, ep[dn]]
            orc.reset_where(odone)
            want = np.where(dn[:, None, None, None], orc.observe(), oo)
            assert np.array_equal(d.cpu().numpy(), odone), (env_id, t)
            assert np.array_equal(o.cpu().numpy(), want), (env_id, t)
            assert np.array_equal(r.cpu().numpy(), orew.astype(np.float32)), (env_id, t)
        st = env.get_state()
        assert np.array_equal(st["grid"], orc.grid) and np.array_equal(st["agent"], orc.agent) and np.array_equal(st["steps"], orc.steps), env_id
        assert env.stats()["episodes"] == int(ep.sum())
        env.close()
        total += N * T
        print("%-40s %6d envs x %5d steps, %8d episodes (most per env %3d): every byte equal  (%.1f s)" % (env_id, N, T, int(ep.sum()), int(ep.max()), time.perf_counter() - t0), flush=True)
    print("soak_stream ok: %d env-steps compared, every env on its own level stream" % total)


if __name__ == "__main__":
    main()
