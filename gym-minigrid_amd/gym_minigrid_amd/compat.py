"""Single-env adapter with the reference's own calling convention, for code written against `gym.make(id)`:

    env = gym_minigrid_amd.make("MiniGrid-DoorKey-8x8-v0")        # instead of gym.make(...)
    env.seed(1337); obs = env.reset()                              # obs = {'image': (7,7,3) uint8, 'direction': int, 'mission': str}
    obs, reward, done, info = env.step(env.actions.forward)       # reward float, done bool, info {}

(minigrid.py:831-863, 1227-1325, 1359-1381; the caller loop of run_tests.py:41-68 runs unchanged.)  It is a VecMiniGrid
of ONE env on the GPU with auto_reset off -- every call costs a kernel launch and a device sync, so this is for porting
and debugging, not for throughput: batch with VecMiniGrid for that.
"""
import enum

import numpy as np

from .vec_env import VecMiniGrid


class Actions(enum.IntEnum):  # MiniGridEnv.Actions (minigrid.py:731-745)
    left = 0
    right = 1
    forward = 2
    pickup = 3
    drop = 4
    toggle = 5
    done = 6


class SingleEnv:
    actions = Actions

    def __init__(self, env_id, device=0, **kwargs):
        self._vec = VecMiniGrid(env_id, num_envs=1, device=device, seeds=1337, auto_reset=False, backend="numpy", **kwargs)
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self.reward_range = self._vec.reward_range
        self.max_steps = self._vec.max_steps
        self.width, self.height = self._vec.width, self._vec.height
        self._seed = 1337  # MiniGridEnv.__init__(seed=1337)
        self.reset()

    def seed(self, seed=1337):
        self._seed = int(seed)
        self._vec.seed(np.array([self._seed], np.uint64))
        return [seed]

    def _obs(self, image):
        return {"image": image[0].copy(), "direction": int(self._vec.direction()[0]), "mission": self._vec.missions()[0]}

    def reset(self):
        """A reset WITHOUT a preceding seed() continues the env's RNG stream in the reference; here it replays the level of
        the last seed (ReseedWrapper semantics) -- call seed() first for a specific level, as the reference's tests do."""
        return self._obs(self._vec.reset())

    def step(self, action):
        obs, reward, done, info = self._vec.step(np.array([int(action)], np.uint8))
        return self._obs(obs), float(reward[0]), bool(done[0]), info

    # --- the attributes the reference's callers read (minigrid.py:816-823)
    def _state(self):
        return self._vec.get_state()

    @property
    def agent_pos(self):
        return tuple(int(v) for v in self._vec.pose()[0, :2])

    @property
    def agent_dir(self):
        return int(self._vec.pose()[0, 2])

    @property
    def step_count(self):
        return int(self._state()["steps"][0])

    @property
    def carrying(self):
        """encode() of the carried object, or None."""
        c = tuple(int(v) for v in self._state()["carry"][0])
        return None if c == (1, 0, 0) else c

    @property
    def mission(self):
        return self._vec.missions()[0]

    def encode_grid(self):
        """env.grid.encode(): uint8 (W, H, 3)."""
        return self._state()["grid"][0].copy()

    def close(self):
        self._vec.close()


def make(env_id, **kwargs):
    """gym.make(env_id) for the built-in ids (gym_minigrid_amd.env_ids())."""
    return SingleEnv(env_id, **kwargs)
