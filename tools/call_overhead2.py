import os, sys, time, ctypes
sys.path.insert(0, "/root/repo/gym-minigrid_amd")
import torch, gym_minigrid_amd as mg
from gym_minigrid_amd import _lib
N=4096; K=20000
env = mg.VecMiniGrid("MiniGrid-Empty-8x8-v0", num_envs=N, seeds=0)
env.reset()
a = torch.full((N,), 2, dtype=torch.uint8, device="cuda")
for _ in range(2000): env.step(a)
torch.cuda.synchronize()
fn = env._step_fn; h = env._h; ap = a.data_ptr(); o = env._out_ptrs
t0=time.perf_counter()
for _ in range(K): fn(h, ap, *o)
t1=time.perf_counter(); torch.cuda.synchronize()
print("raw ctypes mgx_step: %.2f us/call" % ((t1-t0)/K*1e6))
t0=time.perf_counter()
for _ in range(K): env.step(a)
t1=time.perf_counter(); torch.cuda.synchronize()
print("VecMiniGrid.step:    %.2f us/call" % ((t1-t0)/K*1e6))
# an empty kernel launch through torch for comparison
x = torch.zeros(1, device="cuda")
t0=time.perf_counter()
for _ in range(K): x.add_(1)
t1=time.perf_counter(); torch.cuda.synchronize()
print("torch x.add_(1):     %.2f us/call" % ((t1-t0)/K*1e6))
