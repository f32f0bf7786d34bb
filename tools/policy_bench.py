"""Step time under an action distribution other than bench.py's uniform one: python tools/policy_bench.py <env_id> <n_envs> <p_forward> [p_turn]
(the rest is spread over pickup / drop / toggle / done).  A walking policy is the worst case of the gather form's window records
(StepParams.wcache): every move is a miss.  Compare with MGX_GATHER_CACHE=off."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
import torch  # noqa: E402

import gym_minigrid_amd as mg  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniGrid-FourRooms-v0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
pf = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
pt = float(sys.argv[4]) if len(sys.argv) > 4 else 0.3
T = 64
env = mg.VecMiniGrid(env_id, num_envs=N, seeds=0, backend="torch", auto_reset=True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
u = torch.rand((T, N), device="cuda", generator=g)
rest = (1.0 - pf - pt) / 4
acts = torch.full((T, N), 6, dtype=torch.uint8, device="cuda")
edges = [(pt / 2, 0), (pt, 1), (pt + pf, 2), (pt + pf + rest, 3), (pt + pf + 2 * rest, 4), (pt + pf + 3 * rest, 5)]
lo = 0.0
for hi, a in edges:
    acts[(u >= lo) & (u < hi)] = a
    lo = hi
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(4):
        for t in range(T):
            env.step(acts[t])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (4 * T)
print("%s  %d envs  forward %.2f turn %.2f  cache %s: %.1f us per step (%s)" % (env_id, N, pf, pt, os.environ.get("MGX_GATHER_CACHE", "on"), dt * 1e6, env.step_kernel_name()), flush=True)
env.close()
