"""k_seed alone and the full reset around it: `env.seed(s_i); env.reset()` for every env of a handle, with NEW seeds every time
(so that nothing is skipped as "same seed").  Prints host-timed ms per reset; run under rocprofv3 --kernel-trace --stats (or --pmc) for
the per-kernel figures:  python tools/seed_bench.py [env_id] [n_envs] [repeats]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
import ctypes  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import gym_minigrid_amd as mg  # noqa: E402
from gym_minigrid_amd import _lib  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniGrid-DoorKey-8x8-v0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
R = int(sys.argv[3]) if len(sys.argv) > 3 else 8
env = mg.VecMiniGrid(env_id, num_envs=N, seeds=0, backend="torch", auto_reset=False)
seeds = torch.arange(N, dtype=torch.int64, device="cuda")
obs = env._obs
for r in range(R + 2):
    if r == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    seeds += N
    _lib.check(_lib.lib().mgx_reset(env._h, ctypes.c_void_p(seeds.data_ptr()), None, ctypes.c_void_p(obs.data_ptr())))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
st = env.get_state() if N <= 1 << 16 else None
print("%s  %d envs  full reset with new seeds: %.3f ms" % (env_id, N, dt * 1e3), flush=True)
if st is not None:  # small runs double as a check against the host generator
    want = mg.generate_levels(env_id, seeds.cpu().numpy().astype(np.uint64))
    assert np.array_equal(st["grid"], want[0]) and np.array_equal(st["agent"], want[1])
    print("levels == host generator")
env.close()
