#!/usr/bin/env python3
"""Host-side cost of one mgx_step call (small batch: the kernel is ~3 us, the loop is launch-bound)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
import torch
import gym_minigrid_amd as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = 20000
env = mg.VecMiniGrid("MiniGrid-Empty-8x8-v0", num_envs=N, seeds=0)
env.reset()
a = torch.full((N,), 2, dtype=torch.uint8, device="cuda")
for _ in range(2000): env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): env.step(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("N=%d MGX_PTR_CACHE=%s: %.2f us/call issue, %.2f us/step with drain" % (N, os.environ.get("MGX_PTR_CACHE", "1"), (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
