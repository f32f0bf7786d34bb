import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
import numpy as np, torch
import gym_minigrid_amd as mg
from oracle.dynobs_oracle import DynObsOracle
for env_id, size, nob in [("MiniGrid-Dynamic-Obstacles-8x8-v0", 8, 4), ("MiniGrid-Dynamic-Obstacles-16x16-v0", 16, 8), ("MiniGrid-Dynamic-Obstacles-Random-6x6-v0", 6, 3)]:
    N, T = 1500, 1200
    t0 = time.perf_counter()
    seeds = np.arange(N, dtype=np.uint64) * 13 + 1
    orc = DynObsOracle(size, nob, "Random" in env_id, seeds)
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, auto_reset=True, backend="torch")
    assert np.array_equal(env.reset().cpu().numpy(), orc.observe())
    rs = np.random.RandomState(2)
    ep = 0
    for t in range(T):
        a = rs.choice([0, 1, 2, 0, 1, 5, 200], size=N).astype(np.uint8)   # mostly turns: long episodes cross RNG blocks
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        oo, orew, odone = orc.step(a)
        orc.reset_where(odone)
        if odone.any():
            oo[odone.astype(bool)] = orc.observe()[odone.astype(bool)]
        assert np.array_equal(done.cpu().numpy(), odone), (env_id, t)
        assert np.array_equal(rew.cpu().numpy(), orew.astype(np.float32)), (env_id, t)
        assert np.array_equal(obs.cpu().numpy(), oo), (env_id, t)
        ep += int(odone.sum())
    assert np.array_equal(env.get_state()["grid"], orc.base.grid)
    env.close()
    print("%-44s %d envs x %d steps, %7d episodes: every byte equal  (%.1f s)" % (env_id, N, T, ep, time.perf_counter() - t0), flush=True)
