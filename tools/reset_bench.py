import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/gym-minigrid_amd')
import torch, numpy as np, gym_minigrid_amd as mg, ctypes
from gym_minigrid_amd import _lib
IDS = sys.argv[1:] or ("MiniGrid-LavaCrossingS9N1-v0", "MiniGrid-DoorKey-8x8-v0", "MiniGrid-Empty-8x8-v0", "MiniGrid-Dynamic-Obstacles-8x8-v0", "MiniGrid-Fetch-8x8-N3-v0", "MiniGrid-MultiRoom-N4-S5-v0")
for env_id in IDS:
    N = 1 << 20
    env = mg.VecMiniGrid(env_id, num_envs=N, seeds=0, backend="torch", auto_reset=False)
    torch.cuda.synchronize(); t0=time.perf_counter(); env.reset(); torch.cuda.synchronize(); print(env_id, "full reset %.2f ms" % ((time.perf_counter()-t0)*1e3))
    T = 64
    acts = env.fill_actions(1, 0, T)
    seeds_d = torch.from_numpy(env.seeds.astype(np.int64)).cuda()
    def loop(with_reset, new_seeds=False):
        torch.cuda.synchronize(); t0=time.perf_counter()
        for t in range(T):
            obs, rew, done, _ = env.step(acts[t])
            if new_seeds:
                seeds_d.add_(1 << 20)   # every finished env is re-seeded with a seed it has not had: k_seed_masked + k_levelgen really run
            if with_reset:
                _lib.check(_lib.lib().mgx_reset(env._h, ctypes.c_void_p(seeds_d.data_ptr()), ctypes.c_void_p(done.data_ptr()), ctypes.c_void_p(obs.data_ptr())))
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/T*1e6
    loop(True)
    print("   step only %.1f us/step; step + masked reset (device seeds/mask) %.1f us/step; the same with a NEW seed per reset %.1f us/step" % (loop(False), loop(True), min(loop(True, True), loop(True, True))), flush=True)
    env.close()
