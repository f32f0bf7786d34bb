#!/usr/bin/env python3
"""The reference's own caller loop (run_tests.py:41-68) on one env of the GPU library -- a porting aid, not the fast path.

    python examples/single_env.py [env id]
"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gym-minigrid_amd"))
import gym_minigrid_amd as mg  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniGrid-DoorKey-8x8-v0"
env = mg.make(env_id)                      # instead of gym.make(env_id)
env.seed(1337)
obs = env.reset()
print(env_id, "| mission:", obs["mission"], "| image", obs["image"].shape, "| direction", obs["direction"])
random.seed(0)
num_episodes, ret = 0, 0.0
for t in range(2000):
    action = random.randint(0, env.action_space.n - 1)
    obs, reward, done, info = env.step(action)
    ret += reward
    assert obs["image"].shape == env.observation_space.spaces["image"].shape and env.agent_dir == obs["direction"]
    if done:
        num_episodes += 1
        env.seed(1337)
        obs = env.reset()
print("2000 steps, %d episodes, return %.3f, agent at %s facing %d, carrying %s" % (num_episodes, ret, env.agent_pos, env.agent_dir, env.carrying))
env.close()
