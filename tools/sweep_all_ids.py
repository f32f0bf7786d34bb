#!/usr/bin/env python3
"""One short bench per built-in env id (GPU box): us per step at N envs with random actions, to spot ids that are far off their
family's rate.  python tools/sweep_all_ids.py [N] [steps] [partial|full] [newlevel]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
import torch, gym_minigrid_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
T = int(sys.argv[2]) if len(sys.argv) > 2 else 160
MODE = sys.argv[3] if len(sys.argv) > 3 else "partial"
NEWLEVEL = len(sys.argv) > 4 and sys.argv[4] == "newlevel"
for env_id in mg.env_ids():
    cfg = mg.env_config(env_id)
    n = N if cfg.width * cfg.height <= 256 else N // 2
    if MODE == "full":
        n //= 2
    try:
        env = mg.VecMiniGrid(env_id, num_envs=n, seeds=0, backend="torch", obs_mode=MODE, new_level_each_episode=NEWLEVEL)
    except Exception as e:  # (a family without an on-device generator in stream mode)
        print("%-46s skipped: %s" % (env_id, str(e)[:80]), flush=True)
        continue
    env.reset()
    acts = env.fill_actions(1, 0, T)
    for t in range(16): env.step(acts[t])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(16, T): env.step(acts[t])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (T - 16)
    st = env.stats()
    print("%-46s %3dx%-3d n=%7d  %7.1f us/step  %6.2f G env-steps/s  %5.1f ns/kenv  episodes %d" % (env_id, cfg.width, cfg.height, n, dt * 1e6, n / dt / 1e9, dt / n * 1e12, st["episodes"]), flush=True)
    env.close()
