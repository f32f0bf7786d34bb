"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's DynamicObstaclesEnv (envs/dynamicobstacles.py) for a
batch of envs.  Only tests/ may import this.

What is restated here (Python loops: small batches only):
  * legacy gym seeding                  gym.utils.seeding.np_random -> numpy RandomState (the reference's own RNG)
  * DynamicObstaclesEnv._gen_grid       envs/dynamicobstacles.py:35-58 (walls, goal, agent, n blue balls via place_obj)
  * MiniGridEnv.place_obj / place_agent minigrid.py:1003-1090 (rejection sampling, RecursionError after max_tries)
  * DynamicObstaclesEnv.step            envs/dynamicobstacles.py:60-89: fold actions >= 3 to 0, `not_clear` of the front
                                        cell BEFORE the obstacles move, every obstacle re-placed in its 3x3
                                        neighbourhood (kept where it is when 101 tries fail), the base step, then
                                        reward -1 / done when the agent moved forward while not_clear
The base `MiniGridEnv.step` + `gen_obs` run through the C oracle (minigrid_oracle.c), which is pinned separately.
Pinned against tests/golden/DynObs-*.npz (recorded from the reference) by tests/test_dynobs.py.
"""
import hashlib

import numpy as np

from .minigrid_oracle import OracleEnvs

EMPTY, WALL, BALL, GOAL = 1, 2, 6, 8
BLUE, GREEN, GREY = 2, 1, 5


def np_random(seed):
    """gym < 0.22 seeding: sha512(str(seed))[:8] as little-endian uint32 words -> RandomState.seed (init_by_array)."""
    seed = int(seed) % 2 ** 64
    h = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
    v = int.from_bytes(h, "little")
    words = []
    while v > 0:
        v, m = divmod(v, 2 ** 32)
        words.append(m)
    rng = np.random.RandomState()
    rng.seed(words or [0])
    return rng


def n_obstacles_of(size, n_obstacles):
    """envs/dynamicobstacles.py:22-26"""
    return int(n_obstacles) if n_obstacles <= size / 2 + 1 else int(size / 2)


class DynObsOracle:
    def __init__(self, size, n_obstacles, random_start, seeds, view=7, seed_lists=None, seed_idx=0):
        self.size, self.n_obst, self.random_start = int(size), n_obstacles_of(size, n_obstacles), bool(random_start)
        if seed_lists is not None:  # ReseedWrapper(env, seeds=seed_lists[e], seed_idx): `seeds` is ignored
            self.seed_lists = [[int(v) for v in sl] for sl in seed_lists]
            self.seed_idx = [int(seed_idx)] * len(self.seed_lists)
            seeds = [0] * len(self.seed_lists)
        self.seeds = [int(s) for s in seeds]
        self.n = len(self.seeds)
        self.max_steps = 4 * size * size
        self.base = OracleEnvs(size, size, self.max_steps, True, False, view=view)
        self.rng = [None] * self.n
        self.obst = [None] * self.n
        grid = np.zeros((self.n, size, size, 3), np.uint8)
        agent = np.zeros((self.n, 3), np.int32)
        self.base.set_state(grid, agent)
        self.reset_where(np.ones(self.n, bool))

    # ---- minigrid.py:1003-1058
    def _place(self, e, top, size, max_tries, agent_pos):
        g = self.base.grid[e]
        W = H = self.size
        top = (max(top[0], 0), max(top[1], 0))
        tries = 0
        while True:
            if tries > max_tries:
                raise RecursionError
            tries += 1
            x = self.rng[e].randint(top[0], min(top[0] + size[0], W))
            y = self.rng[e].randint(top[1], min(top[1] + size[1], H))
            if g[x, y, 0] != EMPTY:
                continue
            if agent_pos is not None and (x, y) == tuple(agent_pos):
                continue
            return x, y

    def reset_where(self, mask, reseed=True):
        """env.seed(s); env.reset() for the masked envs (ReseedWrapper semantics, wrappers.py:24-28).  reseed=False: the plain reset()
        (minigrid.py:831-858) -- the env's RandomState goes on from where the obstacle walks left it.  With `seed_lists` set (one list
        per env) and reseed=True: ReseedWrapper(seeds=list) -- the next entry of the env's list, cyclically (wrappers.py:19-28)."""
        S = self.size
        for e in np.flatnonzero(np.asarray(mask, bool)):
            if reseed:
                lists = getattr(self, "seed_lists", None)
                if lists is not None:
                    self.seeds[e] = int(lists[e][self.seed_idx[e]])
                    self.seed_idx[e] = (self.seed_idx[e] + 1) % len(lists[e])
                self.rng[e] = np_random(self.seeds[e])
            g = self.base.grid[e]
            g[:] = (EMPTY, 0, 0)
            g[0, :] = g[S - 1, :] = g[:, 0] = g[:, S - 1] = (WALL, GREY, 0)
            g[S - 2, S - 2] = (GOAL, GREEN, 0)
            if self.random_start:                                    # place_agent(): place_obj(None) then _rand_int(0, 4)
                ax, ay = self._place(e, (0, 0), (S, S), np.inf, None)
                ad = self.rng[e].randint(0, 4)
            else:
                ax, ay, ad = 1, 1, 0
            self.base.agent[e] = (ax, ay, ad)
            self.obst[e] = []
            for _ in range(self.n_obst):                             # place_obj(Ball(), max_tries=100)
                x, y = self._place(e, (0, 0), (S, S), 100, (ax, ay))
                g[x, y] = (BALL, BLUE, 0)
                self.obst[e].append((x, y))
            self.base.aux[e] = 0
            self.base.carry[e] = (1, 0, 0)
            self.base.steps[e] = 0

    def observe(self):
        return self.base.observe()

    def step(self, actions):
        a = np.array(actions, np.uint8).copy()
        a[a >= 3] = 0                                               # action >= action_space.n -> 0
        dx, dy = (1, 0, -1, 0), (0, 1, 0, -1)
        crash = np.zeros(self.n, bool)
        for e in range(self.n):
            ax, ay, ad = (int(v) for v in self.base.agent[e])
            g = self.base.grid[e]
            front = g[ax + dx[ad], ay + dy[ad], 0]
            not_clear = front != EMPTY and front != GOAL
            for i, (ox, oy) in enumerate(self.obst[e]):
                try:
                    x, y = self._place(e, (ox - 1, oy - 1), (3, 3), 100, (ax, ay))
                except RecursionError:
                    continue
                g[x, y] = (BALL, BLUE, 0)
                g[ox, oy] = (EMPTY, 0, 0)
                self.obst[e][i] = (x, y)
            crash[e] = a[e] == 2 and not_clear
        obs, reward, done = self.base.step(a)
        reward = np.where(crash, -1.0, reward)
        done = np.where(crash, 1, done).astype(np.uint8)
        return obs, reward, done
