import sys, time, numpy as np, torch
sys.path.insert(0, "gym-minigrid_amd")
import gym_minigrid_amd as mg
for N in (4096, 65536):
    env = mg.VecMiniGrid("MiniGrid-DoorKey-8x8-v0", num_envs=N, seeds=0)
    ref = mg.VecMiniGrid("MiniGrid-DoorKey-8x8-v0", num_envs=N, seeds=0)
    env.reset(); ref.reset()
    T = 16
    acts = env.fill_actions(1, 0, T)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        env.step(acts[0])           # binds the stream, warms up
    ref.step(acts[0])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for t in range(1, T):
            env.step(acts[t])
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    for t in range(1, T):
        o, r, d, _ = ref.step(acts[t])
    torch.cuda.synchronize()
    same = torch.equal(env._obs, ref._obs) and torch.equal(env._done, ref._done)
    # timing: graph replay of 15 steps vs eager
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / (200 * (T - 1))
    t0 = time.perf_counter()
    for _ in range(200):
        for t in range(1, T): ref.step(acts[t])
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / (200 * (T - 1))
    print("N=%d graph==eager: %s   per step: graph %.2f us, eager %.2f us" % (N, same, tg * 1e6, te * 1e6))
