#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python - <<'PY'
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "gym-minigrid_amd")
import torch, gym_minigrid_amd as mg
for env_id, N, view in (("MiniGrid-DoorKey-8x8-v0", 1048576, 7), ("MiniGrid-DoorKey-8x8-v0", 1048576, 3), ("MiniGrid-DoorKey-8x8-v0", 1048576, 5), ("MiniGrid-DoorKey-8x8-v0", 524288, 9),
                        ("MiniGrid-DoorKey-8x8-v0", 262144, 11), ("MiniGrid-FourRooms-v0", 262144, 7), ("MiniGrid-MultiRoom-N6-v0", 262144, 7), ("MiniGrid-DoorKey-8x8-v0", 4096, 5)):
    T = 64
    res = {}
    for form in ("fused", "graph"):
        os.environ["MGX_ROLLOUT"] = form
        env = mg.VecMiniGrid(env_id, num_envs=N, seeds=0, backend="torch", agent_view_size=view)
        env.reset()
        acts = env.fill_actions(1, 0, T)
        for _ in range(2): env.rollout(acts, with_obs=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R = 6
        for _ in range(R): env.rollout(acts)
        torch.cuda.synchronize(); res[form] = (time.perf_counter() - t0) / (R * T)
        env.close()
    print("%-28s view %2d N=%8d T=%3d | rollout graph: %7.2f us %6.2f G/s | rollout fused: %7.2f us %6.2f G/s" % (
        env_id, view, N, T, res["graph"] * 1e6, N / res["graph"] / 1e9, res["fused"] * 1e6, N / res["fused"] / 1e9), flush=True)
PY
