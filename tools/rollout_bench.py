#!/usr/bin/env python3
"""env.rollout(actions[T, N]) against T env.step calls (GPU box): python tools/rollout_bench.py [env_id]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
import torch, gym_minigrid_amd as mg
env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniGrid-DoorKey-8x8-v0"
for N in (1024, 16384, 262144, 524288, 1048576):
    T = 64 if N >= 262144 else 256
    res = {}
    for form in ("fused", "graph"):
        os.environ["MGX_ROLLOUT"] = form
        env = mg.VecMiniGrid(env_id, num_envs=N, seeds=0, backend="torch")
        env.reset()
        acts = env.fill_actions(1, 0, T)
        for _ in range(2): env.rollout(acts, with_obs=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R = 8
        for _ in range(R): env.rollout(acts)
        torch.cuda.synchronize(); res[form] = (time.perf_counter() - t0) / (R * T)
        if form == "graph":
            for t in range(32): env.step(acts[t])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for r in range(2):
                for t in range(T): env.step(acts[t])
            torch.cuda.synchronize(); res["step"] = (time.perf_counter() - t0) / (2 * T)
        env.close()
    print("%s N=%8d T=%3d  step(): %7.2f us/step %6.2f G/s | rollout graph: %7.2f us %6.2f G/s | rollout fused: %7.2f us %6.2f G/s" % (
        env_id, N, T, res["step"] * 1e6, N / res["step"] / 1e9, res["graph"] * 1e6, N / res["graph"] / 1e9, res["fused"] * 1e6, N / res["fused"] / 1e9), flush=True)
