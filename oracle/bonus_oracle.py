"""CPU restatement of the reference's exploration-bonus wrappers, ActionBonus and StateBonus (gym_minigrid/wrappers.py:87-153).

TEST INFRASTRUCTURE: imported by tests/ only (the product counts on the GPU, inside the step kernels: exploration_bonus, k_step.hip).
Pinned by the `Bonus-*` traces recorded through the reference's own wrapper classes (oracle/gen_golden.py, tests/golden/Bonus-*.npz).

The wrappers keep `self.counts`, a dict keyed by (tuple(agent_pos), agent_dir, action) / tuple(agent_pos) of the state AFTER the step,
never cleared by reset(); every step adds `1 / math.sqrt(new_count)` to the reward in Python doubles.  Here the dicts of n envs are dense
int64 arrays and the arithmetic is numpy float64 (the same IEEE operations)."""
import numpy as np


class BonusOracle:
    def __init__(self, n, W, H, kinds, n_actions=7):
        self.kinds = list(kinds)                      # stacking order, innermost first: the innermost wrapper adds its bonus first
        self.action = np.zeros((n, W, H, 4, n_actions), np.int64)   # ActionBonus.counts  (wrappers.py:94-112)
        self.state = np.zeros((n, W, H), np.int64)                  # StateBonus.counts   (wrappers.py:127-147)

    def step(self, reward, agent, actions, where=None):
        """reward (n,) of the wrapped env's step, agent (n, 3) = (x, y, dir) AFTER it, actions (n,) as the caller gave them
        -> the reward the outermost wrapper returns (float64).  where: bool (n,) of the envs that really stepped (default all)."""
        n = len(reward)
        idx = np.arange(n) if where is None else np.flatnonzero(where)
        r = np.asarray(reward, np.float64).copy()
        x, y, d, a = agent[idx, 0], agent[idx, 1], agent[idx, 2], np.asarray(actions)[idx].astype(np.int64)
        for k in self.kinds:
            if k == "action":
                self.action[idx, x, y, d, a] += 1      # new_count = pre_count + 1   (wrappers.py:108-110)
                c = self.action[idx, x, y, d, a]
            else:
                self.state[idx, x, y] += 1             # (wrappers.py:142-144)
                c = self.state[idx, x, y]
            r[idx] = r[idx] + 1.0 / np.sqrt(c.astype(np.float64))   # bonus = 1 / math.sqrt(new_count); reward += bonus
        return r


class DacOracle:
    """The fork's DACWrapper (wrappers.py:35-84) over n envs: `env_done`, `count` and the all-ones `last_obs` image.  Used like the wrapper:
    reset_where(mask) when the caller resets, then per step `stepping()` = the envs whose env.step() the wrapper still calls, and
    step(obs, reward, done) on what those returned (rows of absorbed envs are ignored) -> what the wrapper returns."""

    def __init__(self, n, max_steps):
        self.env_done = np.zeros(n, bool)
        self.count = np.zeros(n, np.int64)
        self.max_steps = int(max_steps)

    def reset_where(self, mask):
        m = np.asarray(mask).astype(bool)
        self.env_done[m] = False      # (wrappers.py:44-56)
        self.count[m] = 0

    def stepping(self):
        return ~self.env_done         # `if self.env_done: ... return self.last_obs, 0, ...` never touches the env (wrappers.py:60-66)

    def step(self, obs, reward, done):
        self.count += 1
        was = self.env_done.copy()
        d = np.asarray(done).astype(bool) & ~was
        out_obs = np.array(obs, copy=True)
        out_rew = np.where(was, 0.0, np.asarray(reward, np.float64))
        self.env_done = was | d
        out_obs[self.env_done] = 1    # last_obs['image'] = obs['image'] * 0 + 1   (wrappers.py:51)
        out_done = (self.env_done & (self.count >= self.max_steps)).astype(np.uint8)
        return out_obs, out_rew, out_done
