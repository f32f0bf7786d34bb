#!/usr/bin/env python3
"""A policy in the loop, small batch: `actions = policy(obs); obs = env.step(actions)` captured once into a HIP graph
(torch.cuda.CUDAGraph) and replayed -- below ~200 k envs a step is launch-bound (~8 us of host time per call), and a
graph takes the host out of the loop.  mgx_step makes no allocation and no synchronisation with device pointers, so it can be
captured like any other kernel launch (tests/test_gpu_api.py::test_hipgraph_replay_equals_eager).

    python examples/policy_graph.py [env_id] [num_envs] [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))

import torch  # noqa: E402
import gym_minigrid_amd as mg  # noqa: E402


def main():
    env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniGrid-DoorKey-8x8-v0"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    dev = torch.device("cuda:0")
    env = mg.VecMiniGrid(env_id, num_envs=n, seeds=0, new_level_each_episode=True)
    obs = env.reset()
    torch.manual_seed(0)
    policy = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(147, 64), torch.nn.ReLU(), torch.nn.Linear(64, 7)).to(dev).half()
    actions = torch.zeros(n, dtype=torch.uint8, device=dev)   # static buffers: a graph replays fixed addresses
    returns = torch.zeros(n, dtype=torch.float32, device=dev)

    def one_step():
        with torch.no_grad():
            logits = policy(env._obs.half())                    # env._obs: the handle's observation buffer (what step() returns)
            actions.copy_(torch.argmax(logits + torch.rand_like(logits), dim=1))  # (in-graph RNG: torch registers the generator with the capture)
        _, reward, _, _ = env.step(actions)
        returns.add_(reward)

    def run(fn, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    for _ in range(20):
        one_step()
    eager = run(one_step, steps)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        one_step()                                              # warm the side stream (binds the env to it)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(8):                                      # 8 policy+env steps per replay
            one_step()
    graph = run(g.replay, steps // 8) / 8
    st = env.stats()
    print("%s, %d envs: eager %.1f us per policy+env step (%.2f M env-steps/s) | HIP graph %.1f us (%.2f M env-steps/s) | %d episodes, return sum %.2f" % (
        env_id, n, eager * 1e6, n / eager / 1e6, graph * 1e6, n / graph / 1e6, st["episodes"], float(returns.sum())))
    env.close()


if __name__ == "__main__":
    main()
