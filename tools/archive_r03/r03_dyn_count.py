#!/usr/bin/env python3
"""How often a lane of k_dynobs leaves the straight-line placements for the loop (needs a build with -DMGX_EXP_DYN=32 as MGX_LIB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
import numpy as np, torch
import gym_minigrid_amd as mg
from gym_minigrid_amd import _lib
for env_id in sys.argv[1:]:
    env = mg.VecMiniGrid(env_id, num_envs=262144, seeds=1, auto_reset=True, backend="torch")
    env.reset()
    lib = ctypes.CDLL(_lib.SO_PATH)
    c0 = (ctypes.c_ulonglong * 4)()
    lib.mgx_debug_dyn_count(c0)
    for t in range(100):
        env.step(env.fill_actions(3, t, 1)[0] % 3)
    c = (ctypes.c_ulonglong * 4)()
    lib.mgx_debug_dyn_count(c)
    d = [c[i] - c0[i] for i in range(3)]
    print(env_id, "lanes into the loop %d of %d lane-steps (%.3f %%), waves with one %d of %d (%.1f %%)" % (d[0], d[2], 100.0 * d[0] / d[2], d[1], d[2] // 64, 100.0 * d[1] / (d[2] / 64)))
    env.close()
