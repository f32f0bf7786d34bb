import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/gym-minigrid_amd')
import torch, numpy as np, gym_minigrid_amd as mg
for N in (1024, 4096, 16384, 65536, 262144):
    env = mg.VecMiniGrid("MiniGrid-DoorKey-8x8-v0", num_envs=N, seeds=0, backend="torch")
    env.reset()
    T = 256
    acts = env.fill_actions(1, 0, T)
    for _ in range(2): env.rollout(acts, with_obs=True)
    torch.cuda.synchronize(); t0=time.perf_counter()
    R=8
    for _ in range(R): env.rollout(acts)
    torch.cuda.synchronize(); dt_g=(time.perf_counter()-t0)/(R*T)
    for t in range(32): env.step(acts[t])
    torch.cuda.synchronize(); t0=time.perf_counter()
    for r in range(2):
        for t in range(T): env.step(acts[t])
    torch.cuda.synchronize(); dt_s=(time.perf_counter()-t0)/(2*T)
    print("N=%7d  step(): %6.2f us/step %7.3f G/s   rollout graph: %6.2f us/step %7.3f G/s" % (N, dt_s*1e6, N/dt_s/1e9, dt_g*1e6, N/dt_g/1e9), flush=True)
    env.close()
