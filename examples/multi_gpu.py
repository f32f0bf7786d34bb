#!/usr/bin/env python3
"""One process per GPU, envs sharded by global index, the per-env done / reward vectors gathered over RCCL for logging
(BASELINE configs[3]: MiniGrid-LavaCrossingS9N1-v0, 524,288 envs per GPU).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        examples/multi_gpu.py [env_id] [envs_per_gpu] [steps]

Results do not depend on the number of GPUs: seeds and (here) the synthetic actions are keyed by the GLOBAL env index."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC between the ranks, before HIP starts

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import gym_minigrid_amd as mg  # noqa: E402
from gym_minigrid_amd import dist as mdist  # noqa: E402


def main():
    env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniGrid-LavaCrossingS9N1-v0"
    n_local = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    rank, local_rank, world = mdist.init_process_group()      # "nccl" = RCCL on ROCm
    torch.cuda.set_device(local_rank)
    offset, count = mdist.shard(n_local * world, rank, world)  # contiguous block of global env indices
    env = mg.VecMiniGrid(env_id, num_envs=count, device=local_rank, seeds=0, env_offset=offset)
    env.reset()
    log = mdist.GatherLogger(count, torch.device("cuda", local_rank), world)
    t0 = time.perf_counter()
    for t in range(steps):
        actions = env.fill_actions(0, t, 1)[0]                 # your policy here (this stream is keyed by the global index)
        obs, reward, done, _ = env.step(actions)               # observations stay on their GPU
        if t % 256 == 255:
            log.submit(done, reward)                           # all-gather on a side stream: the step stream does not wait
    all_done, all_reward = log.wait()                          # (world * count,) in global env order, on every rank
    env.sync()
    dt = time.perf_counter() - t0
    if rank == 0:
        print("%s: %d envs on %d GPUs, %.2f G env-steps/s; last gathered step: %d episodes ended, reward sum %.3f" % (
            env_id, count * world, world, count * world * steps / dt / 1e9, int(all_done.sum()), float(all_reward.sum())))
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
