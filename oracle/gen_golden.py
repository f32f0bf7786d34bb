#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE, runs only in the build container).

Imports the read-only reference at /root/reference under the stand-in `gym`
package in oracle/refshim/, drives `MiniGridEnv.step` / `gen_obs` /
`FullyObsWrapper.observation` (reference: gym_minigrid/minigrid.py:1227-1381,
gym_minigrid/wrappers.py:311-338) and records inputs + expected outputs as small
.npz fixtures under tests/golden/.  Nothing of the reference's source travels:
the fixtures are data (states, actions, observations).

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Every trace stores the *explicit* initial state (encoded grid + aux plane +
agent pose), so step/observation parity never depends on how a level was
generated; the seed is stored as well so that level generation itself can be
checked (tests/test_levelgen.py).

File format (one .npz per case), all arrays stacked over traces (leading dim K):
  meta            json: env_id, W, H, max_steps, view (agent_view_size), see_through, lava_v1, full_obs, reseed
  seed      (K,)  int64   seed used for `env.seed(s); env.reset()` (-1: hand-built state)
  init_grid (K,W,H,3) u8  Grid.encode() of the initial grid, index [x][y][c]
  init_aux  (K,W,H)   u8  bit0 = Goal.overlap (terminal goal, toggletimes<=0)
  init_agent(K,3)     i32 x, y, dir
  init_obs  (K,7,7,3) u8  obs['image'] returned by reset()
  actions   (K,T)     u8
  obs       (K,T,7,7,3) u8 ; direction (K,T) u8 ; reward (K,T) f64 ; done (K,T) u8
  agent     (K,T,3)   i32 ; carry (K,T,3) u8 ((1,0,0) = nothing) ; steps (K,T) i32
  grid      (K,T,W,H,3) u8  post-step Grid.encode()
  full      (K,T,W,H,3) u8  FullyObsWrapper image (only when meta.full_obs)
  init_full (K,W,H,3)       FullyObsWrapper image at reset (only when meta.full_obs)
  Episode boundaries (caller-side reset on done, run_tests.py:64-66 convention; meta.reseed=True:
  re-seeding with the SAME seed = ReseedWrapper(seeds=[s]), wrappers.py:24-28; meta.reseed=False: plain
  `reset()`, the env's RNG stream continues and every episode gets a new level):
  reset_k (R,) trace index, reset_t (R,) step index after which reset happened,
  reset_grid (R,W,H,3), reset_aux (R,W,H), reset_agent (R,3), reset_obs (R,7,7,3), reset_task (R,),
  reset_contains (R,W,H,3) (objstate), reset_full (R,W,H,3) (full_obs)
"""
import json
import os
import sys
from collections import deque

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "refshim"))

import numpy as np  # noqa: E402
import gym  # noqa: E402
import gym_minigrid  # noqa: E402,F401
from gym_minigrid import minigrid as M  # noqa: E402
from gym_minigrid.wrappers import ActionBonus, DACWrapper, FullyObsWrapper, ReseedWrapper, StateBonus, ViewSizeWrapper  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


# --------------------------------------------------------------------------- helpers
def obj_aux(o):
    """Hidden Goal/Box state as one byte: bit0 Goal.overlap, bits 3:1 triage_color+1 (0 = None), bits 7:4 (toggletimes-1)&15."""
    if o is None or o.type not in ("goal", "box"):
        return 0
    a = 1 if (o.type == "goal" and o.overlap) else 0
    if o.triage_color is not None:
        a |= (M.COLOR_TO_IDX[o.triage_color] + 1) << 1
    tt = max(int(o.toggletimes), 0)       # Box counts below zero; every value <= 0 behaves the same
    assert tt <= 15
    return a | (((tt - 1) & 15) << 4)


def obj_contains(o):
    c = getattr(o, "contains", None) if o is not None else None
    return c.encode() if c is not None else (1, 0, 0)


def aux_plane(env):
    W, H = env.grid.width, env.grid.height
    a = np.zeros((W, H), np.uint8)
    for x in range(W):
        for y in range(H):
            a[x, y] = obj_aux(env.grid.get(x, y))
    return a


def contains_plane(env):
    W, H = env.grid.width, env.grid.height
    c = np.zeros((W, H, 3), np.uint8)
    for x in range(W):
        for y in range(H):
            c[x, y] = obj_contains(env.grid.get(x, y))
    return c


def task_word(env):
    """Fetch: target object as a cell code (type | color << 4).  GoToObject: tx | ty << 4 | (type - key) << 8 | color << 10."""
    if type(env).__name__.startswith("PutNear"):
        tx, ty = env.target_pos
        return ((M.OBJECT_TO_IDX[env.move_type] - 5) | (M.COLOR_TO_IDX[env.moveColor] << 2) | (int(tx) << 5) | (int(ty) << 8) |
                ((M.OBJECT_TO_IDX[env.target_type] - 5) << 11) | (M.COLOR_TO_IDX[env.target_color] << 13))
    if type(env).__name__ == "LockedRoom":
        locked = [rm for rm in env.rooms if rm.locked][0]
        key = [(x, y) for x in range(env.width) for y in range(env.height) if env.grid.get(x, y) is not None and env.grid.get(x, y).type == "key"][0]
        kroom = [rm for rm in env.rooms if rm.top[0] < key[0] < rm.top[0] + rm.size[0] - 1 and rm.top[1] < key[1] < rm.top[1] + rm.size[1] - 1][0]
        return M.COLOR_TO_IDX[locked.color] | (M.COLOR_TO_IDX[kroom.color] << 3)
    if type(env).__name__ == "Unlock":
        return int(env.door.cur_pos[1])
    if type(env).__name__ in ("UnlockPickup", "BlockedUnlockPickup") or type(env).__name__.startswith(("KeyCorridor", "ObstructedMaze")):
        return M.OBJECT_TO_IDX[env.obj.type] | (M.COLOR_TO_IDX[env.obj.color] << 4)
    if type(env).__name__.startswith("Memory"):
        return int(env.success_pos[0]) | (int(env.success_pos[1] < env.height // 2) << 4)
    if type(env).__name__.startswith("RedBlueDoor"):
        ry = [y for y in range(env.height) if env.grid.get(env.size // 2, y) is env.red_door][0]
        by = [y for y in range(env.height) if env.grid.get(env.size // 2 + env.size - 1, y) is env.blue_door][0]
        return ry | (by << 4)
    if type(env).__name__.startswith(("GoToObject", "GotoEnv")):
        tx, ty = env.target_pos
        return int(tx) | (int(ty) << 4) | ((M.OBJECT_TO_IDX[env.targetType] - 5) << 8) | (M.COLOR_TO_IDX[env.target_color] << 10)
    if hasattr(env, "targetType"):
        return M.OBJECT_TO_IDX[env.targetType] | (M.COLOR_TO_IDX[env.targetColor] << 4)
    return 0


def carry_triple(env):
    return env.carrying.encode() if env.carrying is not None else (1, 0, 0)


def full_image(env):
    return FullyObsWrapper.observation(_FakeWrap(env), {"mission": ""})["image"]


class _FakeWrap:
    """Lets us call FullyObsWrapper.observation on a bare env without re-wrapping."""

    def __init__(self, env):
        self.unwrapped = env


class SoupEnv(M.MiniGridEnv):
    """Our own level: walled room filled at random with every object kind the
    hot path distinguishes (reference classes, reference step/gen_obs)."""

    def __init__(self, width, height, see_through, max_steps, density, v1=False, extended=False, default_vis=True):
        self._density = density
        super().__init__(width=width, height=height, max_steps=max_steps,
                         see_through_walls=see_through, extended_actions=extended, default_vis=default_vis)

    def _gen_grid(self, width, height):
        rs = self.np_random
        self.grid = M.Grid(width, height)
        self.grid.wall_rect(0, 0, width, height)
        colors = list(M.COLOR_TO_IDX.keys())
        for x in range(1, width - 1):
            for y in range(1, height - 1):
                if rs.uniform(0, 1) >= self._density:
                    continue
                k = rs.randint(0, 12)
                c = colors[rs.randint(0, len(colors))]
                if k == 0:
                    o = M.Wall(c)
                elif k == 1:
                    o = M.Floor(c)
                elif k in (2, 3, 4):
                    st = rs.randint(0, 3)
                    o = M.Door(c, is_open=(st == 0), is_locked=(st == 2))
                elif k in (5, 6):
                    o = M.Key(c)
                elif k == 7:
                    o = M.Ball(c)
                elif k == 8:
                    o = M.Box(c)
                elif k == 9:
                    o = M.Goal()
                elif k == 10:
                    o = M.Goal(toggletimes=0)
                else:
                    o = M.Lava()
                self.grid.set(x, y, o)
        # agent on a free interior cell
        while True:
            x, y = rs.randint(1, width - 1), rs.randint(1, height - 1)
            o = self.grid.get(x, y)
            if o is None or o.can_overlap():
                break
        self.agent_pos = np.array((x, y))
        self.agent_dir = rs.randint(0, 4)
        self.mission = "soup"


class SoupAuxEnv(SoupEnv):
    """Soup with the hidden Goal/Box state exercised: toggletimes 0..3, triage colours, boxes with contents."""

    def _gen_grid(self, width, height):
        super()._gen_grid(width, height)
        rs = self.np_random
        colors = list(M.COLOR_TO_IDX.keys())
        for x in range(1, width - 1):
            for y in range(1, height - 1):
                o = self.grid.get(x, y)
                if o is None or (x, y) == tuple(self.agent_pos):
                    continue
                tri = colors[rs.randint(0, 7)] if rs.randint(0, 2) else None
                if o.type == "goal":
                    self.grid.set(x, y, M.Goal(toggletimes=rs.randint(0, 4), triage_color=tri))
                elif o.type == "box":
                    inner = [None, M.Key, M.Ball][rs.randint(0, 3)]
                    inner = inner(colors[rs.randint(0, 7)]) if inner else None
                    self.grid.set(x, y, M.Box(o.color, contains=inner, toggletimes=rs.randint(1, 4), triage_color=tri))


class SoupEnvv1(SoupEnv):
    """Class name contains 'v1' -> lava gives reward -1 and no done (minigrid.py:1262-1268)."""


class PlantedGoalEnv(M.MiniGridEnv):
    """Empty 8x8 with a terminal goal Goal(toggletimes=0) at (3,1): general goal/reward branch."""

    def __init__(self):
        super().__init__(grid_size=8, max_steps=256, see_through_walls=True)

    def _gen_grid(self, width, height):
        self.grid = M.Grid(width, height)
        self.grid.wall_rect(0, 0, width, height)
        self.put_obj(M.Goal(), width - 2, height - 2)
        self.put_obj(M.Goal(toggletimes=0), 3, 1)
        self.agent_pos = np.array((1, 1))
        self.agent_dir = 0
        self.mission = "planted"


# --------------------------------------------------------------------------- planner (own code)
DIRS = [(1, 0), (0, 1), (-1, 0), (0, -1)]


def plan_face(env, target, passable_extra=()):
    """Actions that bring the agent adjacent to `target`, facing it (BFS over (x,y,dir))."""
    W, H = env.grid.width, env.grid.height

    def free(x, y):
        if not (0 <= x < W and 0 <= y < H):
            return False
        if (x, y) in passable_extra:
            return True
        o = env.grid.get(x, y)
        return o is None or (o.can_overlap() and o.type != "lava")

    start = (int(env.agent_pos[0]), int(env.agent_pos[1]), int(env.agent_dir))
    prev = {start: None}
    q = deque([start])
    goal_state = None
    while q:
        s = q.popleft()
        x, y, d = s
        if (x + DIRS[d][0], y + DIRS[d][1]) == tuple(target):
            goal_state = s
            break
        nxt = [((x, y, (d + 3) % 4), 0), ((x, y, (d + 1) % 4), 1)]
        fx, fy = x + DIRS[d][0], y + DIRS[d][1]
        if free(fx, fy):
            nxt.append(((fx, fy, d), 2))
        for n, a in nxt:
            if n not in prev:
                prev[n] = (s, a)
                q.append(n)
    if goal_state is None:
        return None
    acts = []
    s = goal_state
    while prev[s] is not None:
        s, a = prev[s]
        acts.append(a)
    return acts[::-1]


def find(env, typ):
    for x in range(env.grid.width):
        for y in range(env.grid.height):
            o = env.grid.get(x, y)
            if o is not None and o.type == typ:
                return (x, y)
    return None


def doorkey_script(env):
    """pickup key, try to drop onto the wall (no-op), unlock+open door, close, open,
    walk through the doorway (agent stands in the open door carrying the key),
    step on the goal (reward 0, not done in this fork), toggle the goal away."""
    acts = []

    def run(seq):
        for a in seq:
            env.step(a)
            acts.append(a)

    key = find(env, "key")
    run(plan_face(env, key))
    run([2, 3, 3, 4, 4, 3])          # bump key, pickup, pickup again, drop, drop(no-op: occupied), pickup
    door = find(env, "door")
    run(plan_face(env, door))
    run([4, 5, 5, 5, 2, 2])          # drop on door (no-op), unlock/open, close, open, into doorway, beyond
    goal = find(env, "goal")
    p = plan_face(env, goal)
    run(p)
    run([2, 6, 0, 0, 2, 0, 0, 5])    # onto goal, done-action, turn round, step off, turn back, toggle goal away
    run([4, 1, 1])                   # drop key where the goal was, look around
    return acts


# --------------------------------------------------------------------------- recorder
def record_case(name, make_env, seeds, T, scripts=None, full_obs=False, v1=False, reseed=True, n_actions=7, objstate=False, gym_id=None,
                seed_lists=None, seed_idx0=0, bonus=()):
    """seed_lists (round 4): trace k runs under the reference's own ReseedWrapper(env, seeds=seed_lists[k], seed_idx=seed_idx0)
    (wrappers.py:12-28): every reset() -- the first one included -- seeds with the next entry of the list, cyclically.  `seeds` is
    then ignored; z['seed'][k] is the seed of the first episode and z['seed_list'] (K, L) the lists.
    bonus (round 4): the reference's ActionBonus / StateBonus (wrappers.py:87-153) stacked around the env in the given order, innermost
    first ("action", "state"); z['reward'] is then what the outermost wrapper's step() returns; meta['bonus'] keeps the order."""
    if seed_lists is not None:
        seeds = [sl[seed_idx0] for sl in seed_lists]
    K = len(seeds)
    env0 = make_env()
    W, H = env0.width, env0.height
    V = int(env0.agent_view_size)
    meta = dict(env_id=name, W=W, H=H, max_steps=int(env0.max_steps), view=V,
                see_through=bool(env0.see_through_walls), lava_v1=bool(v1), full_obs=bool(full_obs),
                reseed=("list" if seed_lists is not None else bool(reseed)), seed_idx0=int(seed_idx0), extended=bool(n_actions > 7), alt_vis=not bool(env0.default_vis),
                task=11 if type(env0).__name__.startswith("TwoGoals") else 10 if type(env0).__name__.startswith("PutNear") else 9 if type(env0).__name__ == "LockedRoom" else 7 if type(env0).__name__ == "Unlock" else 8 if type(env0).__name__ in ("UnlockPickup", "BlockedUnlockPickup") or type(env0).__name__.startswith(("KeyCorridor", "ObstructedMaze")) else 6 if type(env0).__name__.startswith("Memory") else 5 if type(env0).__name__.startswith("RedBlueDoor") else 4 if type(env0).__name__.startswith(("GoToObject", "GotoEnv")) else (1 if hasattr(env0, "targetType") else (2 if hasattr(env0, "target_pos") else 0)),
                objstate=bool(objstate), dynobs=int(getattr(env0, "n_obstacles", 0)), gym_id=gym_id or "", bonus=list(bonus))
    z = dict(
        seed=np.zeros(K, np.int64), init_grid=np.zeros((K, W, H, 3), np.uint8),
        init_aux=np.zeros((K, W, H), np.uint8), init_agent=np.zeros((K, 3), np.int32),
        init_obs=np.zeros((K, V, V, 3), np.uint8), actions=np.zeros((K, T), np.uint8),
        obs=np.zeros((K, T, V, V, 3), np.uint8), direction=np.zeros((K, T), np.uint8),
        reward=np.zeros((K, T), np.float64), done=np.zeros((K, T), np.uint8),
        agent=np.zeros((K, T, 3), np.int32), carry=np.zeros((K, T, 3), np.uint8),
        steps=np.zeros((K, T), np.int32), grid=np.zeros((K, T, W, H, 3), np.uint8), init_task=np.zeros(K, np.uint32))
    if full_obs:
        z["full"] = np.zeros((K, T, W, H, 3), np.uint8)
        z["init_full"] = np.zeros((K, W, H, 3), np.uint8)
    if objstate:
        z["init_contains"] = np.zeros((K, W, H, 3), np.uint8)
        z["aux"] = np.zeros((K, T, W, H), np.uint8)
        z["contains"] = np.zeros((K, T, W, H, 3), np.uint8)
        z["carry_aux"] = np.zeros((K, T), np.uint8)
        z["carry_contains"] = np.zeros((K, T, 3), np.uint8)
    rk, rt, rg, ra, rag, ro, rtask, rcont, rfull = [], [], [], [], [], [], [], [], []
    if seed_lists is not None:
        z["seed_list"] = np.asarray(seed_lists, np.int64)
    for k, s in enumerate(seeds):
        env = make_env()
        wrapped = None
        if seed_lists is not None:
            wrapped = ReseedWrapper(env, seeds=[int(v) for v in seed_lists[k]], seed_idx=int(seed_idx0))
            o = wrapped.reset()
        else:
            env.seed(int(s))
            o = env.reset()
        stepper = env
        for b in bonus:   # ("dac": the fork's DACWrapper, wrappers.py:35-84 -- obs / reward / done are then the wrapper's and resets go through it;
            stepper = {"action": ActionBonus, "state": StateBonus, "dac": DACWrapper}[b](stepper)   # the recorded state arrays stay the ENV's)
        if "dac" in bonus:
            env.seed(int(s))
            o = stepper.reset()   # (DACWrapper.reset sets last_obs and count; re-seeded: the env draws the same level again)
        z["seed"][k] = s
        z["init_grid"][k] = env.grid.encode()
        z["init_aux"][k] = aux_plane(env)
        z["init_agent"][k] = (env.agent_pos[0], env.agent_pos[1], env.agent_dir)
        z["init_obs"][k] = o["image"]
        z["init_task"][k] = task_word(env)
        if objstate:
            z["init_contains"][k] = contains_plane(env)
        if full_obs:
            z["init_full"][k] = full_image(env)
        # action stream: optional scripted prefix (computed on a scratch copy), then random
        acts = []
        if scripts is not None and scripts[k] is not None:
            scratch = make_env()
            scratch.seed(int(s))
            scratch.reset()
            acts = scripts[k](scratch)
        rnd = np.random.RandomState(1000 + k).randint(0, n_actions, size=T)
        stream = (list(acts) + list(rnd))[:T]
        for t, a in enumerate(stream):
            if a == 8:  # strafe_right onto a goal reads left_cell.overlap (minigrid.py:1310): AttributeError unless
                rc = env.grid.get(*env.right_pos)   # the LEFT cell is a goal too -> keep such steps out of the fixtures
                lc = env.grid.get(*env.left_pos)
                if rc is not None and rc.type == "goal" and not (lc is not None and lc.type == "goal"):
                    a = 6
            if type(env).__name__.startswith("TwoGoals"):   # its own step() raises on pickup / drop and on toggling an empty cell
                if a in (3, 4) or (a == 5 and env.grid.get(*env.front_pos) is None):
                    a = 1
            o, r, d, info = stepper.step(int(a))
            assert info == {}
            z["actions"][k, t] = a
            z["obs"][k, t] = o["image"]
            z["direction"][k, t] = o["direction"]
            z["reward"][k, t] = r
            z["done"][k, t] = d
            z["agent"][k, t] = (env.agent_pos[0], env.agent_pos[1], env.agent_dir)
            z["carry"][k, t] = carry_triple(env)
            z["steps"][k, t] = env.step_count
            z["grid"][k, t] = env.grid.encode()
            if objstate:
                z["aux"][k, t] = aux_plane(env)
                z["contains"][k, t] = contains_plane(env)
                z["carry_aux"][k, t] = obj_aux(env.carrying)
                z["carry_contains"][k, t] = obj_contains(env.carrying)
            if full_obs:
                z["full"][k, t] = full_image(env)
            if d:
                if wrapped is not None:
                    o2 = wrapped.reset()  # ReseedWrapper: seed(next of the list); reset()
                else:
                    if reseed:
                        env.seed(int(s))
                    o2 = stepper.reset() if "dac" in bonus else env.reset()  # reseed=False: the env's own RNG stream continues -> a new level
                rk.append(k)
                rt.append(t)
                rg.append(env.grid.encode())
                ra.append(aux_plane(env))
                rag.append((env.agent_pos[0], env.agent_pos[1], env.agent_dir))
                ro.append(o2["image"])
                rtask.append(task_word(env))
                rcont.append(contains_plane(env))
                if full_obs:
                    rfull.append(full_image(env))
    R = len(rk)
    z["reset_k"] = np.asarray(rk, np.int32)
    z["reset_t"] = np.asarray(rt, np.int32)
    z["reset_grid"] = np.asarray(rg, np.uint8).reshape(R, W, H, 3)
    z["reset_aux"] = np.asarray(ra, np.uint8).reshape(R, W, H)
    z["reset_agent"] = np.asarray(rag, np.int32).reshape(R, 3)
    z["reset_obs"] = np.asarray(ro, np.uint8).reshape(R, V, V, 3)
    z["reset_task"] = np.asarray(rtask, np.uint32)
    if objstate:
        z["reset_contains"] = np.asarray(rcont, np.uint8).reshape(R, W, H, 3)
    if full_obs:
        z["reset_full"] = np.asarray(rfull, np.uint8).reshape(R, W, H, 3)   # FullyObsWrapper image returned by reset()
    z["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **z)
    print("%-34s K=%d T=%d resets=%d dones=%d nonzero-reward=%d  %6.1f KB" % (
        name, K, T, R, int(z["done"].sum()), int((z["reward"] != 0).sum()), os.path.getsize(path) / 1024))


def record_level_streams():
    """Plain reference env (no ReseedWrapper): `env.seed(s)` once, then K consecutive `reset()`s continue the env's
    RNG stream (minigrid.py:836-839) -> a fresh level per episode.  K is large enough to cross several MT19937
    state regenerations (624 outputs each)."""
    out = {}
    for env_id, seeds, K in [("MiniGrid-DoorKey-8x8-v0", [0, 1, 7], 700), ("MiniGrid-DoorKey-5x5-v0", [3], 300),
                             ("MiniGrid-DoorKey-16x16-v0", [2], 200),
                             ("MiniGrid-LavaCrossingS9N1-v0", [0, 5], 1500), ("MiniGrid-LavaCrossingS9N3-v0", [1], 500),
                             ("MiniGrid-LavaCrossingS11N5-v0", [2], 400), ("MiniGrid-SimpleCrossingS9N2-v0", [4], 400),
                             ("MiniGrid-Empty-Random-6x6-v0", [0, 9], 900), ("MiniGrid-Empty-Random-10x10-v0", [1], 400),
                             ("MiniGrid-LavaGapS7-v0", [0, 3], 900), ("MiniGrid-LavaGapS6-v1", [1], 900),
                             ("MiniGrid-NormalGapS6-v0", [2], 300), ("MiniGrid-Empty-8x8-v0", [0], 5),
                             ("MiniGrid-MultiRoom-N2-S4-v0", [0], 300), ("MiniGrid-MultiRoom-N6-v0", [1, 4], 400),
                             ("MiniGrid-Fetch-8x8-N3-v0", [0, 2], 700), ("MiniGrid-GoToDoor-6x6-v0", [1], 700),
                             ("MiniGrid-FourRooms-v0", [0, 3], 600),
                             ("MiniGrid-PutNear-6x6-N2-v0", [0], 400), ("MiniGrid-GoToObject-6x6-N2-v0", [1], 400),
                             ("MiniGrid-KeyCorridorS3R3-v0", [0], 300), ("MiniGrid-KeyCorridorS6R3-v0", [2], 150),
                             ("MiniGrid-Playground-v0", [0], 100), ("MiniGrid-LockedRoom-v0", [1], 150),
                             ("MiniGrid-MemoryS13Random-v0", [0], 300), ("MiniGrid-BlockedUnlockPickup-v0", [0], 300),
                             ("MiniGrid-RedBlueDoors-6x6-v0", [3], 300)]:
        env = gym.make(env_id)
        key = env_id.replace("MiniGrid-", "").replace("-v0", "")
        for s in seeds:
            env.seed(int(s))
            grids, agents = [], []
            for _ in range(K):
                env.reset()
                grids.append(env.grid.encode())
                agents.append((env.agent_pos[0], env.agent_pos[1], env.agent_dir))
            out["%s:%d:grid" % (key, s)] = np.asarray(grids, np.uint8)
            out["%s:%d:agent" % (key, s)] = np.asarray(agents, np.int32)
    path = os.path.join(OUT, "level_streams.npz")
    np.savez_compressed(path, **out)
    print("level_streams.npz %6.1f KB" % (os.path.getsize(path) / 1024))


def record_onehot():
    """Known answers for the one-hot epilogue: FullyObsOneHotWrapper.observation (wrappers.py:340-415, default and
    drop_color, flatten=False) applied to FullyObs images of a DoorKey rollout.  (OneHotPartialObsWrapper,
    wrappers.py:203-243, cannot be run: it allocates `np.zeros(self.observation_space.shape)` on a Dict space, whose
    shape is None, and raises IndexError on the first cell; its formula is restated in tests/helpers.py.)"""
    from gym_minigrid.wrappers import FullyObsOneHotWrapper, ImgObsWrapper
    env = gym.make("MiniGrid-DoorKey-8x8-v0")
    env.seed(3)
    env.reset()
    scr = gym.make("MiniGrid-DoorKey-8x8-v0")
    scr.seed(3)
    scr.reset()
    acts = doorkey_script(scr) + list(np.random.RandomState(0).randint(0, 7, size=40))
    fenv = ImgObsWrapper(FullyObsWrapper(env))
    f1 = FullyObsOneHotWrapper(fenv, flatten=False)
    f2 = FullyObsOneHotWrapper(fenv, drop_color=True, flatten=False)
    part, part_oh, full, full_oh, full_oh_nc = [], [], [], [], []
    for a in acts:
        o, r, d, _ = env.step(int(a))
        part.append(o["image"])
        fo = full_image(env)
        full.append(fo)
        full_oh.append(f1.observation(fo))
        full_oh_nc.append(f2.observation(fo))
    path = os.path.join(OUT, "onehot.npz")
    np.savez_compressed(path, part=np.asarray(part), full=np.asarray(full),
                        full_oh=np.asarray(full_oh), full_oh_nc=np.asarray(full_oh_nc))
    print("onehot.npz %6.1f KB  shapes %s %s" % (os.path.getsize(path) / 1024, np.asarray(full_oh).shape, np.asarray(full_oh_nc).shape))


def record_flat():
    """Known answers for the flat epilogue: FlatObsWrapper.observation (wrappers.py:528-577) on seeded episodes of
    families with constant and per-episode (Fetch) missions, plus one FlatObsWrapper(FullyObsWrapper(env))."""
    from gym_minigrid.wrappers import FlatObsWrapper
    out = {}
    cases = [("MiniGrid-Empty-8x8-v0", [0], False), ("MiniGrid-DoorKey-5x5-v0", [1, 2], False),
             ("MiniGrid-Fetch-5x5-N2-v0", list(range(16)), False), ("MiniGrid-Fetch-8x8-N3-v0", list(range(8)), False),
             ("MiniGrid-GoToDoor-5x5-v0", [0, 1], False), ("MiniGrid-FourRooms-v0", [0], False),
             ("MiniGrid-GoToObject-6x6-N2-v0", list(range(8)), False), ("MiniGrid-LockedRoom-v0", list(range(6)), False),
             ("MiniGrid-KeyCorridorS3R3-v0", [0, 1, 2], False), ("MiniGrid-UnlockPickup-v0", [0, 1, 2], False), ("MiniGrid-MemoryS9-v0", [0], False), ("MiniGrid-PutNear-8x8-N3-v0", list(range(6)), False), ("MiniGrid-Dynamic-Obstacles-6x6-v0", [0, 1], False),
             ("MiniGrid-LavaCrossingS9N1-v0", [0], False), ("MiniGrid-SimpleCrossingS9N1-v0", [0], False),
             ("MiniGrid-MultiRoom-N2-S4-v0", [0], False), ("MiniGrid-Empty-5x5-v0", [0], True), ("MiniGrid-Fetch-5x5-N2-v0", [3, 4], True)]
    ids, seeds, fulls, acts, flats, missions = [], [], [], [], [], []
    T = 4
    for env_id, ss, full in cases:
        base = gym.make(env_id)
        env = FlatObsWrapper(FullyObsWrapper(base) if full else base)
        for sd in ss:
            env.seed(int(sd))
            rows = [env.reset()]
            a = np.random.RandomState(sd + 77).randint(0, 3, size=T)   # turns/forward only: never ends the episode early
            ms = [base.mission]
            for t in range(T):
                o, r, d, _ = env.step(int(a[t]))
                assert not d
                rows.append(o)
                ms.append(base.mission)
            assert all(r.dtype == np.float32 for r in rows)
            ids.append(env_id); seeds.append(sd); fulls.append(full); acts.append(a); missions.append(ms[0])
            assert len(set(ms)) == 1
            flats.append(np.stack(rows))
    for k, f in enumerate(flats):
        out["flat_%d" % k] = f
    path = os.path.join(OUT, "flat.npz")
    np.savez_compressed(path, ids=np.array(ids), seeds=np.array(seeds, np.uint64), full=np.array(fulls), actions=np.array(acts, np.uint8),
                        missions=np.array(missions), **out)
    print("flat.npz %6.1f KB  %d episodes" % (os.path.getsize(path) / 1024, len(ids)))


OBSTRUCTED = ["1Dl", "1Dlh", "1Dlhb", "2Dl", "2Dlh", "2Dlhb", "1Q", "2Q", "Full"]


def record_levels_obstructed():
    """ObstructedMaze (envs/obstructedmaze.py): seed -> initial grid / agent / target / Box.contains plane (keys hidden in boxes),
    and no-reseed level streams for three of the ids.  A file of its own: levels.npz is not re-recorded for it."""
    out = {}
    big = [1337, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 12345, 2 ** 64 - 1]
    for short in OBSTRUCTED:
        env = gym.make("MiniGrid-ObstructedMaze-%s-v0" % short)
        ss = list(range(128 if short.startswith("1D") else 64)) + big
        grids, agents, tasks, conts = [], [], [], []
        for sd in ss:
            env.seed(int(sd))
            env.reset()
            grids.append(env.grid.encode())
            agents.append((env.agent_pos[0], env.agent_pos[1], env.agent_dir))
            tasks.append(task_word(env))
            conts.append(contains_plane(env))
            assert env.mission == "pick up the blue ball"
        key = "ObstructedMaze-" + short
        out[key + ":seeds"] = np.asarray(ss, np.uint64)
        out[key + ":grid"] = np.asarray(grids, np.uint8)
        out[key + ":agent"] = np.asarray(agents, np.int32)
        out[key + ":task"] = np.asarray(tasks, np.uint32)
        out[key + ":contains"] = np.asarray(conts, np.uint8)
        out[key + ":max_steps"] = np.asarray([env.max_steps, int(env.see_through_walls)], np.int32)
    for short, sd, K in [("1Dlhb", 3, 400), ("2Dlh", 1, 150), ("Full", 2, 100)]:
        env = gym.make("MiniGrid-ObstructedMaze-%s-v0" % short)
        env.seed(sd)
        grids, agents, conts = [], [], []
        for _ in range(K):
            env.reset()
            grids.append(env.grid.encode())
            agents.append((env.agent_pos[0], env.agent_pos[1], env.agent_dir))
            conts.append(contains_plane(env))
        key = "stream:ObstructedMaze-%s:%d" % (short, sd)
        out[key + ":grid"] = np.asarray(grids, np.uint8)
        out[key + ":agent"] = np.asarray(agents, np.int32)
        out[key + ":contains"] = np.asarray(conts, np.uint8)
    path = os.path.join(OUT, "levels_obstructed.npz")
    np.savez_compressed(path, **out)
    print("levels_obstructed.npz %6.1f KB" % (os.path.getsize(path) / 1024))


def obstructed_script(env):
    """Open boxes, pick up keys, move blocking balls aside, unlock doors, then pick up the blue ball -- re-planning on the
    scratch env after every sub-goal.  Not every level gets solved within the cap; whatever prefix results is recorded."""
    acts = []

    def do(seq):
        for k in seq:
            if len(acts) < 330:
                env.step(k)
                acts.append(k)

    def find(kind, pred=lambda o: True):
        return [(x, y) for x in range(env.width) for y in range(env.height)
                if env.grid.get(x, y) is not None and env.grid.get(x, y).type == kind and pred(env.grid.get(x, y))]

    def reach(cands):
        best = None
        for c in cands:
            p = plan_face(env, c)
            if p is not None and (best is None or len(p) < len(best)):
                best = p
        return best

    def drop():
        for _ in range(4):
            if env.carrying is None:
                return
            do([0, 4])

    blocked_door, useless = None, set()
    for _ in range(90):
        if env.step_count >= env.max_steps - 2 or len(acts) >= 320:
            break
        blue = reach(find("ball", lambda o: o.color == "blue"))
        if blue is not None:
            if env.carrying is not None:
                drop()
                continue
            do(blue + [3])
            break
        c = env.carrying
        if c is not None and c.type == "key":
            doors = find("door", lambda o: o.is_locked and o.color == c.color)
            p = reach(doors)
            if p is not None:
                do(p + [5])
                continue
            drop()               # its door cannot be faced: a ball stands in front of it, or it is in another room
            blocked_door = doors[0] if doors else None
            if blocked_door is None:
                useless.add(c.color)
            continue
        if c is not None:
            drop()               # a blocking ball: put it down anywhere
            continue
        if blocked_door is not None:
            bx, by = blocked_door
            color = env.grid.get(bx, by).color
            p = reach([q for q in find("ball", lambda o: o.color == "green") if abs(q[0] - bx) + abs(q[1] - by) == 1])
            blocked_door = None
            if p is not None:
                do(p + [3])
                continue
            useless.add(color)
        for kind, then, pred in (("key", [3], lambda o: o.color not in useless), ("box", [5], lambda o: True),
                                 ("door", [5], lambda o: not o.is_open and not o.is_locked)):
            p = reach(find(kind, pred))
            if p is not None:
                do(p + then)
                break
        else:
            break
    return acts


def record_levels():
    """Seeded level generation known answers: seed -> initial grid/agent."""
    out = {}
    for env_id, seeds in [("MiniGrid-Empty-8x8-v0", range(4)), ("MiniGrid-Empty-16x16-v0", range(2)),
                          ("MiniGrid-Empty-5x5-v0", range(2)), ("MiniGrid-Empty-6x6-v0", range(2)),
                          ("MiniGrid-DoorKey-5x5-v0", range(64)), ("MiniGrid-DoorKey-6x6-v0", range(64)),
                          ("MiniGrid-DoorKey-8x8-v0", range(256)), ("MiniGrid-DoorKey-16x16-v0", range(32)),
                          ("MiniGrid-LavaCrossingS9N1-v0", range(256)), ("MiniGrid-LavaCrossingS9N2-v0", range(64)),
                          ("MiniGrid-LavaCrossingS9N0-v0", range(64)), ("MiniGrid-MultiRoom-N2-S4-v0", range(128)),
                          ("MiniGrid-FourRooms-v0", range(128)), ("MiniGrid-Fetch-5x5-N2-v0", range(64)), ("MiniGrid-Fetch-6x6-N2-v0", range(64)),
                          ("MiniGrid-Fetch-8x8-N3-v0", range(128)), ("MiniGrid-GoToDoor-5x5-v0", range(64)),
                          ("MiniGrid-GoToDoor-6x6-v0", range(64)), ("MiniGrid-GoToDoor-8x8-v0", range(128)),
                          ("MiniGrid-GoToObject-6x6-N2-v0", range(128)), ("MiniGrid-GoToObject-8x8-N2-v0", range(128)),
                          ("MiniGrid-RedBlueDoors-6x6-v0", range(128)), ("MiniGrid-RedBlueDoors-8x8-v0", range(128)),
                          ("MiniGrid-KeyCorridorS3R1-v0", range(128)), ("MiniGrid-KeyCorridorS3R2-v0", range(128)), ("MiniGrid-KeyCorridorS3R3-v0", range(128)),
                          ("MiniGrid-KeyCorridorS4R3-v0", range(128)), ("MiniGrid-KeyCorridorS5R3-v0", range(128)), ("MiniGrid-KeyCorridorS6R3-v0", range(128)),
                          ("MiniGrid-TwoGoals-8x8-v0", range(4)), ("MiniGrid-TwoGoals-Random-5x5-v0", range(64)), ("MiniGrid-TwoGoals-Random-6x6-v0", range(64)),
                          ("MiniGrid-TwoGoals-Random-9x9-v0", range(64)), ("MiniGrid-TwoGoals-Random-16x16-v0", range(64)),
                          ("MiniGrid-PutNear-6x6-N2-v0", range(256)), ("MiniGrid-PutNear-8x8-N3-v0", range(256)), ("MiniGrid-LockedRoom-v0", range(256)), ("MiniGrid-Playground-v0", range(256)), ("MiniGrid-Unlock-v0", range(256)), ("MiniGrid-UnlockPickup-v0", range(256)), ("MiniGrid-BlockedUnlockPickup-v0", range(256)),
                          ("MiniGrid-MemoryS7-v0", range(64)), ("MiniGrid-MemoryS9-v0", range(64)), ("MiniGrid-MemoryS11-v0", range(64)),
                          ("MiniGrid-MemoryS13-v0", range(64)), ("MiniGrid-MemoryS13Random-v0", range(128)), ("MiniGrid-MemoryS17Random-v0", range(128)),
                          ("MiniGrid-MultiRoom-N4-S5-v0", range(128)), ("MiniGrid-MultiRoom-N6-v0", range(128)), ("MiniGrid-DistShift1-v0", range(2)),
                          ("MiniGrid-DistShift1-v1", range(2)), ("MiniGrid-DistShift2-v0", range(2)),
                          ("MiniGrid-LavaGapS5-v0", range(32)), ("MiniGrid-LavaGapS7-v0", range(64)),
                          ("MiniGrid-NormalGapS6-v0", range(32)), ("MiniGrid-LavaGapS6-v1", range(32)),
                          ("MiniGrid-Empty-Random-5x5-v0", range(32)), ("MiniGrid-Empty-Random-8x8-v0", range(32)),
                          ("MiniGrid-Empty-Random-10x10-v0", range(32)),
                          ("MiniGrid-LavaCrossingS9N3-v0", range(64)), ("MiniGrid-LavaCrossingS11N5-v0", range(64)),
                          ("MiniGrid-SimpleCrossingS9N1-v0", range(64)), ("MiniGrid-SimpleCrossingS9N2-v0", range(64)),
                          ("MiniGrid-SimpleCrossingS9N3-v0", range(64)), ("MiniGrid-SimpleCrossingS11N5-v0", range(64))]:
        env = gym.make(env_id)
        grids, agents, tasks = [], [], []
        big = [1337, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 12345, 2 ** 64 - 1]
        ss = list(seeds) + big
        for s in ss:
            env.seed(int(s))
            env.reset()
            grids.append(env.grid.encode())
            agents.append((env.agent_pos[0], env.agent_pos[1], env.agent_dir))
            tasks.append(task_word(env))
        key = env_id.replace("MiniGrid-", "")
        key = key[:-3] if key.endswith("-v0") else key
        out[key + ":seeds"] = np.asarray(ss, np.uint64)
        out[key + ":grid"] = np.asarray(grids, np.uint8)
        out[key + ":agent"] = np.asarray(agents, np.int32)
        out[key + ":task"] = np.asarray(tasks, np.uint32)
        out[key + ":max_steps"] = np.asarray([env.max_steps, int(env.see_through_walls)], np.int32)
    path = os.path.join(OUT, "levels.npz")
    np.savez_compressed(path, **out)
    print("levels.npz %6.1f KB" % (os.path.getsize(path) / 1024))


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[2:] if len(sys.argv) > 2 and sys.argv[1] == "--only" else None
    if only and not any(w.startswith("case:") for w in only):   # e.g. --only flat onehot: re-record just those fixture files
        for what in only:
            {"flat": record_flat, "onehot": record_onehot, "levels": record_levels, "level_streams": record_level_streams,
             "levels_obstructed": record_levels_obstructed}[what]()
        return
    if only:                                                     # e.g. --only case:DynObs  (prefix of the case names)
        global record_case
        full_record, prefixes = record_case, tuple(w[5:] for w in only)
        record_case = lambda name, *a, **k: full_record(name, *a, **k) if name.startswith(prefixes) else None  # noqa: E731
    mk = lambda i: (lambda: gym.make(i))  # noqa: E731
    record_case("Empty-8x8", mk("MiniGrid-Empty-8x8-v0"), [0, 1, 2, 3], 300)
    record_case("Empty-5x5", mk("MiniGrid-Empty-5x5-v0"), [0, 1], 120)
    record_case("Empty-Random-6x6", mk("MiniGrid-Empty-Random-6x6-v0"), [0, 1, 2, 3], 160)
    record_case("Empty-16x16-full", mk("MiniGrid-Empty-16x16-v0"), [0, 1], 1100, full_obs=True)
    record_case("DoorKey-8x8", mk("MiniGrid-DoorKey-8x8-v0"), list(range(8)), 700,
                scripts=[doorkey_script] * 4 + [None] * 4)
    record_case("DoorKey-8x8-full", mk("MiniGrid-DoorKey-8x8-v0"), [11, 12], 200,
                scripts=[doorkey_script, None], full_obs=True)
    record_case("DoorKey-5x5", mk("MiniGrid-DoorKey-5x5-v0"), [0, 1, 2, 3], 260, scripts=[doorkey_script] * 4)
    record_case("DoorKey-16x16", mk("MiniGrid-DoorKey-16x16-v0"), [0, 1], 200, scripts=[doorkey_script] * 2)
    record_case("LavaCrossingS9N1", mk("MiniGrid-LavaCrossingS9N1-v0"), list(range(8)), 400)
    record_case("LavaCrossingS9N3", mk("MiniGrid-LavaCrossingS9N3-v0"), [0, 1, 2, 3], 200)
    record_case("LavaCrossingS11N5", mk("MiniGrid-LavaCrossingS11N5-v0"), [0, 1], 200)
    record_case("SimpleCrossingS9N2", mk("MiniGrid-SimpleCrossingS9N2-v0"), [0, 1, 2, 3], 340)
    record_case("LavaGapS7-v1", mk("MiniGrid-LavaGapS7-v1"), [0, 1, 2, 3], 220, v1=True)
    record_case("LavaGapS6", mk("MiniGrid-LavaGapS6-v0"), [0, 1], 160)
    record_case("MultiRoom-N2-S4", mk("MiniGrid-MultiRoom-N2-S4-v0"), [0, 1, 2, 3], 120)
    record_case("MultiRoom-N6", mk("MiniGrid-MultiRoom-N6-v0"), [0, 1], 150)
    record_case("PlantedGoal-8x8", PlantedGoalEnv, [0, 1], 40,
                scripts=[lambda e: [2, 2, 5, 2, 2], lambda e: [6, 2, 0, 1, 2, 2]])
    record_case("Soup-8x8", lambda: SoupEnv(8, 8, False, 96, 0.45), list(range(12)), 200)
    record_case("Soup-8x8-see", lambda: SoupEnv(8, 8, True, 64, 0.45), list(range(4)), 130)
    record_case("Soup-9x9", lambda: SoupEnv(9, 9, False, 100, 0.35), list(range(8)), 200)
    record_case("Soup-7x11", lambda: SoupEnv(7, 11, False, 80, 0.3), list(range(8)), 170)
    record_case("Soup-13x6-full", lambda: SoupEnv(13, 6, False, 80, 0.3), list(range(6)), 170, full_obs=True)
    record_case("Soup-9x9-v1", lambda: SoupEnvv1(9, 9, False, 100, 0.35), list(range(6)), 200, v1=True)
    record_case("Soup-19x19", lambda: SoupEnv(19, 19, False, 150, 0.25), list(range(3)), 300)
    record_case("DistShift1-v1", mk("MiniGrid-DistShift1-v1"), [0, 1], 260, v1=True)
    record_case("DistShift2", mk("MiniGrid-DistShift2-v0"), [0, 1], 260)
    record_case("LavaCrossingS9N0", mk("MiniGrid-LavaCrossingS9N0-v0"), [0, 1, 2, 3], 200)
    record_case("FourRooms", mk("MiniGrid-FourRooms-v0"), [0, 1, 2], 520, reseed=False)
    # other view sizes (ViewSizeWrapper, wrappers.py:579-608: sets env.unwrapped.agent_view_size)
    def vs(make, v):
        def f():
            e = make()
            ViewSizeWrapper(e, v)
            return e
        return f
    record_case("DoorKey-8x8-view5", vs(mk("MiniGrid-DoorKey-8x8-v0"), 5), [0, 1, 2, 3], 300, scripts=[doorkey_script] * 2 + [None] * 2)
    record_case("DoorKey-8x8-view3", vs(mk("MiniGrid-DoorKey-8x8-v0"), 3), [0, 1, 2, 3], 200, scripts=[doorkey_script] * 2 + [None] * 2)
    record_case("Soup-9x9-view9", vs(lambda: SoupEnv(9, 9, False, 100, 0.35), 9), list(range(6)), 200)
    record_case("Soup-7x11-view5", vs(lambda: SoupEnv(7, 11, False, 80, 0.3), 5), list(range(6)), 170)
    record_case("Soup-8x8-see-view3", vs(lambda: SoupEnv(8, 8, True, 64, 0.45), 3), list(range(4)), 130)
    record_case("Soup-19x19-view11", vs(lambda: SoupEnv(19, 19, False, 150, 0.25), 11), list(range(3)), 200)
    record_case("LavaCrossingS9N1-view9", vs(mk("MiniGrid-LavaCrossingS9N1-v0"), 9), [0, 1, 2, 3], 300)
    # ExtendedActions: strafe_left / strafe_right (minigrid.py:747-764,1295-1314)
    record_case("Soup-8x8-strafe", lambda: SoupEnv(8, 8, False, 96, 0.45, extended=True), list(range(12)), 250, n_actions=9)
    record_case("Soup-9x9-v1-strafe", lambda: SoupEnvv1(9, 9, False, 100, 0.35, extended=True), list(range(6)), 250, v1=True, n_actions=9)
    record_case("Soup-13x6-see-strafe", lambda: SoupEnv(13, 6, True, 80, 0.4, extended=True), list(range(6)), 200, n_actions=9)
    # the fork's alternative visibility model, default_vis=False (minigrid.py:649-709)
    record_case("Soup-8x8-altvis", lambda: SoupEnv(8, 8, False, 96, 0.45, default_vis=False), list(range(12)), 200)
    record_case("Soup-9x9-altvis", lambda: SoupEnv(9, 9, False, 100, 0.3, default_vis=False), list(range(8)), 200)
    record_case("Soup-19x19-altvis", lambda: SoupEnv(19, 19, False, 150, 0.2, default_vis=False), list(range(4)), 300)
    record_case("Soup-7x11-altvis-view5", vs(lambda: SoupEnv(7, 11, False, 80, 0.3, default_vis=False), 5), list(range(6)), 170)
    record_case("Soup-9x9-altvis-view9-strafe", vs(lambda: SoupEnv(9, 9, False, 100, 0.3, extended=True, default_vis=False), 9), list(range(6)), 200, n_actions=9)
    # hidden Goal/Box state: toggletimes, triage_color, Box.contains (minigrid.py:156-181,332-364)
    record_case("SoupAux-8x8", lambda: SoupAuxEnv(8, 8, False, 96, 0.55), list(range(16)), 250, objstate=True)
    record_case("SoupAux-9x9-strafe", lambda: SoupAuxEnv(9, 9, True, 100, 0.5, extended=True), list(range(8)), 250, n_actions=9, objstate=True)
    record_case("SoupAux-7x11-full", lambda: SoupAuxEnv(7, 11, False, 80, 0.5), list(range(6)), 200, full_obs=True, objstate=True)
    # moving obstacles drawn from the env's RNG inside step() (envs/dynamicobstacles.py:60-89); actions 3.. fold to 0
    for short, gid, seeds, T in [("DynObs-5x5", "MiniGrid-Dynamic-Obstacles-5x5-v0", range(8), 200),
                                 ("DynObs-Random-6x6", "MiniGrid-Dynamic-Obstacles-Random-6x6-v0", range(12), 300),
                                 ("DynObs-8x8", "MiniGrid-Dynamic-Obstacles-8x8-v0", range(12), 400),
                                 ("DynObs-16x16", "MiniGrid-Dynamic-Obstacles-16x16-v0", range(4), 400)]:
        record_case(short, mk(gid), list(seeds), T, n_actions=4, gym_id=gid)
    # task rules layered on the base step: FetchEnv (envs/fetch.py:74-86), GoToDoorEnv (envs/gotodoor.py:71-93)
    def fetch_script(which):
        def f(env):
            # walk to the target (which=0) or to another object (which=1) and pick it up
            objs = [(x, y) for x in range(env.width) for y in range(env.height)
                    if env.grid.get(x, y) is not None and env.grid.get(x, y).type in ("key", "ball")]
            tgt = [p for p in objs if env.grid.get(*p).type == env.targetType and env.grid.get(*p).color == env.targetColor]
            oth = [p for p in objs if p not in tgt]
            pick = (tgt if which == 0 or not oth else oth)[0]
            acts = plan_face(env, pick) or []
            return acts + [3]
        return f

    def gotodoor_script(which):
        def f(env):
            doors = env.doorPos
            pos = env.target_pos if which == 0 else [d for d in doors if tuple(d) != tuple(env.target_pos)][0]
            acts = plan_face(env, pos) or []
            return acts + [5, 6]   # toggle the locked door (nothing happens), then `done`
        return f
    record_case("Fetch-8x8-N3", mk("MiniGrid-Fetch-8x8-N3-v0"), list(range(8)), 400, scripts=[fetch_script(0), fetch_script(1)] * 2 + [None] * 4, reseed=False)
    record_case("Fetch-5x5-N2", mk("MiniGrid-Fetch-5x5-N2-v0"), list(range(4)), 300, scripts=[fetch_script(1), fetch_script(0), None, None], reseed=False)
    record_case("GoToDoor-8x8", mk("MiniGrid-GoToDoor-8x8-v0"), list(range(8)), 400, scripts=[gotodoor_script(0), gotodoor_script(1)] * 2 + [None] * 4, reseed=False)
    record_case("GoToDoor-5x5", mk("MiniGrid-GoToDoor-5x5-v0"), list(range(4)), 300, scripts=[gotodoor_script(1), gotodoor_script(0), None, None], reseed=False)
    def gotoobject_script(which):
        def f(env):
            # walk next to the target (which=0) or next to another object (which=1) and say `done`
            objs = [(x, y) for x in range(env.width) for y in range(env.height)
                    if env.grid.get(x, y) is not None and env.grid.get(x, y).type in ("key", "ball", "box")]
            oth = [p for p in objs if tuple(p) != tuple(env.target_pos)]
            pick = tuple(env.target_pos) if which == 0 or not oth else oth[0]
            acts = plan_face(env, pick) or []
            return acts + [6]
        return f
    def redblue_script(order):
        def f(env):
            # open the doors in the given order ("rb" pays, "br" does not), closing nothing
            acts = []
            scratch = env
            for ch in order:
                door = scratch.red_door if ch == "r" else scratch.blue_door
                pos = [(x, y) for x in range(scratch.width) for y in range(scratch.height) if scratch.grid.get(x, y) is door][0]
                a = plan_face(scratch, pos) or []
                for k in a + [5]:
                    scratch.step(k)
                acts += a + [5]
            return acts
        return f
    record_case("RedBlueDoors-8x8", mk("MiniGrid-RedBlueDoors-8x8-v0"), list(range(8)), 500, scripts=[redblue_script("rb"), redblue_script("br"), redblue_script("rrb"), redblue_script("b")] + [None] * 4, reseed=False)
    record_case("RedBlueDoors-6x6", mk("MiniGrid-RedBlueDoors-6x6-v0"), list(range(6)), 400, scripts=[redblue_script("br"), redblue_script("rb")] + [None] * 4, reseed=False)
    def unlock_script(pick_box):
        def f(env):
            # (move the blocking ball away,) fetch the key, open the door (, fetch the box): stepping the scratch env as we plan
            acts = []

            def do(seq):
                for k in seq:
                    env.step(k)
                    acts.append(k)

            def find(kind):
                return [(x, y) for x in range(env.width) for y in range(env.height)
                        if env.grid.get(x, y) is not None and env.grid.get(x, y).type == kind]
            door = find("door")[0]
            if env.grid.get(door[0] - 1, door[1]) is not None:          # BlockedUnlockPickup: the ball in front of the door
                do((plan_face(env, (door[0] - 1, door[1])) or []) + [3])
                for _ in range(4):                                     # drop it on the first free side
                    if env.carrying is None:
                        break
                    do([0, 4])
            do((plan_face(env, find("key")[0]) or []) + [3])
            do((plan_face(env, door) or []) + [5])
            if pick_box and find("box"):
                do([4] if False else [])
                # the key is still carried: drop it first (pickup needs empty hands)
                for _ in range(4):
                    if env.carrying is None:
                        break
                    do([0, 4])
                do((plan_face(env, find("box")[0]) or []) + [3, 3])
            return acts
        return f
    record_case("Unlock", mk("MiniGrid-Unlock-v0"), list(range(8)), 400, scripts=[unlock_script(False)] * 3 + [None] * 5, reseed=False)
    record_case("UnlockPickup", mk("MiniGrid-UnlockPickup-v0"), list(range(8)), 400, scripts=[unlock_script(True)] * 3 + [None] * 5, reseed=False)
    record_case("BlockedUnlockPickup", mk("MiniGrid-BlockedUnlockPickup-v0"), list(range(8)), 500, scripts=[unlock_script(True)] * 4 + [None] * 4, reseed=False)

    def keycorridor_script(env):
        # fetch the key, open the locked door, drop the key, pick up the ball -- re-planning on the scratch env; doors on the
        # way are opened as they come (plan_face treats closed doors as walls, so walk door by door)
        acts = []

        def do(seq):
            for k in seq:
                env.step(k)
                acts.append(k)

        def find(kind, pred=lambda o: True):
            return [(x, y) for x in range(env.width) for y in range(env.height)
                    if env.grid.get(x, y) is not None and env.grid.get(x, y).type == kind and pred(env.grid.get(x, y))]

        def go(target, then):
            for _ in range(12):
                p = plan_face(env, target)
                if p is not None:
                    do(p + then)
                    return True
                opened = False
                for d in find("door", lambda o: not o.is_open and not o.is_locked):
                    q = plan_face(env, d)
                    if q is not None:
                        do(q + [5])
                        opened = True
                        break
                if not opened:
                    return False
            return False
        if not go(find("key")[0], [3]):
            return acts
        locked = find("door", lambda o: o.is_locked)
        if not locked or not go(locked[0], [5]):
            return acts
        for _ in range(4):
            if env.carrying is None:
                break
            do([0, 4])
        go(find("ball")[0], [3])
        return acts
    for short, T in [("KeyCorridorS3R1", 200), ("KeyCorridorS3R3", 400), ("KeyCorridorS4R3", 500), ("KeyCorridorS6R3", 600)]:
        record_case(short, mk("MiniGrid-%s-v0" % short), list(range(6)), T, scripts=[keycorridor_script] * 3 + [None] * 3, reseed=False)

    def putnear_script(which):
        def f(env):
            # pick up the object to move (0) or another one (1); with 0 carry it next to the target and drop it
            acts = []

            def do(seq):
                for k in seq:
                    env.step(k)
                    acts.append(k)
            objs = [(x, y) for x in range(env.width) for y in range(env.height)
                    if env.grid.get(x, y) is not None and env.grid.get(x, y).type in ("key", "ball", "box")]
            mv = tuple(env.move_pos)
            oth = [p for p in objs if p != mv]
            do((plan_face(env, mv if which == 0 or not oth else oth[0]) or []) + [3])
            if which == 0 and env.carrying is not None:
                tx, ty = env.target_pos
                for cand in [(tx + 1, ty), (tx - 1, ty), (tx, ty + 1), (tx, ty - 1), (tx + 1, ty + 1), (tx - 1, ty - 1)]:
                    if 0 < cand[0] < env.width - 1 and 0 < cand[1] < env.height - 1 and env.grid.get(*cand) is None and tuple(env.agent_pos) != cand:
                        p = plan_face(env, cand)
                        if p is not None:
                            do(p + [4])
                            break
            return acts
        return f
    record_case("PutNear-8x8-N3", mk("MiniGrid-PutNear-8x8-N3-v0"), list(range(10)), 300, scripts=[putnear_script(0), putnear_script(1)] * 3 + [None] * 4, reseed=False)
    record_case("PutNear-6x6-N2", mk("MiniGrid-PutNear-6x6-N2-v0"), list(range(8)), 240, scripts=[putnear_script(1), putnear_script(0)] * 2 + [None] * 4, reseed=False)
    def twogoals_script(env):
        acts = []
        for color in ("green", "yellow"):
            pos = [(x, y) for x in range(env.width) for y in range(env.height) if env.grid.get(x, y) is not None and env.grid.get(x, y).type == "goal" and env.grid.get(x, y).color == color]
            a = (plan_face(env, pos[0], passable_extra=()) or []) + [5]
            for k in a:
                env.step(k)
            acts += a
        return acts
    import io, contextlib                                 # TwoGoalsEnv.step prints its goal count every step
    for short, gid, seeds in [("TwoGoals-8x8", "MiniGrid-TwoGoals-8x8-v0", range(6)), ("TwoGoals-Random-6x6", "MiniGrid-TwoGoals-Random-6x6-v0", range(8)),
                              ("TwoGoals-Random-16x16", "MiniGrid-TwoGoals-Random-16x16-v0", range(4))]:
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            record_case(short, mk(gid), list(seeds), 300, scripts=[twogoals_script] * 2 + [None] * (len(seeds) - 2), reseed=False)
        print((buf.getvalue().strip().splitlines() or [""])[-1])
    record_case("Playground", mk("MiniGrid-Playground-v0"), list(range(6)), 300, reseed=False)
    record_case("LockedRoom", mk("MiniGrid-LockedRoom-v0"), list(range(6)), 400, reseed=False)

    def memory_script(which):
        def f(env):
            # walk down the hallway to the success (0) or failure (1) cell
            tgt = tuple(env.success_pos if which == 0 else env.failure_pos)
            acts = plan_face(env, tgt) or []
            return acts + [3, 2]   # a pickup (= toggle here) at the wall of objects, then step onto the cell
        return f
    record_case("MemoryS7", mk("MiniGrid-MemoryS7-v0"), list(range(6)), 300, scripts=[memory_script(0), memory_script(1)] + [None] * 4, reseed=False)
    record_case("MemoryS13Random", mk("MiniGrid-MemoryS13Random-v0"), list(range(8)), 500, scripts=[memory_script(0), memory_script(1)] * 2 + [None] * 4, reseed=False)
    record_case("MemoryS17Random", mk("MiniGrid-MemoryS17Random-v0"), list(range(4)), 400, scripts=[memory_script(1), memory_script(0), None, None], reseed=False)
    record_case("GoToObject-8x8-N2", mk("MiniGrid-GoToObject-8x8-N2-v0"), list(range(8)), 400, scripts=[gotoobject_script(0), gotoobject_script(1)] * 2 + [None] * 4, reseed=False)
    record_case("GoToObject-6x6-N2", mk("MiniGrid-GoToObject-6x6-N2-v0"), list(range(6)), 300, scripts=[gotoobject_script(1), gotoobject_script(0)] + [None] * 4, reseed=False)
    # ObstructedMaze (envs/obstructedmaze.py): RoomGrid mazes, keys hidden in boxes (Box.contains), doors blocked by balls
    for short, T, K in [("1Dl", 400, 6), ("1Dlh", 400, 6), ("1Dlhb", 500, 8), ("2Dlhb", 450, 6), ("1Q", 450, 4), ("Full", 400, 4)]:
        box = short not in ("1Dl", "2Dl")
        record_case("ObstructedMaze-" + short, mk("MiniGrid-ObstructedMaze-%s-v0" % short), list(range(K)), T,
                    scripts=[obstructed_script] * (K - 2) + [None] * 2, reseed=False, objstate=box, gym_id="MiniGrid-ObstructedMaze-%s-v0" % short)
    # the same task families with ReseedWrapper semantics at the episode boundary (seed(s); reset() -> the same level again):
    # what the in-kernel auto-reset reproduces; pins its reset observations and task words to the reference
    record_case("ObstructedMaze-1Dlhb-reseed", mk("MiniGrid-ObstructedMaze-1Dlhb-v0"), list(range(6)), 700, scripts=[obstructed_script] * 4 + [None] * 2, objstate=True, gym_id="MiniGrid-ObstructedMaze-1Dlhb-v0")
    record_case("Fetch-8x8-N3-reseed", mk("MiniGrid-Fetch-8x8-N3-v0"), list(range(6)), 700, scripts=[fetch_script(0), fetch_script(1)] * 2 + [None] * 2)
    record_case("GoToDoor-5x5-reseed", mk("MiniGrid-GoToDoor-5x5-v0"), list(range(4)), 300, scripts=[gotodoor_script(1), gotodoor_script(0), None, None])
    record_case("GoToObject-6x6-N2-reseed", mk("MiniGrid-GoToObject-6x6-N2-v0"), list(range(4)), 400, scripts=[gotoobject_script(1), gotoobject_script(0), None, None])
    record_case("RedBlueDoors-6x6-reseed", mk("MiniGrid-RedBlueDoors-6x6-v0"), list(range(4)), 400, scripts=[redblue_script("br"), redblue_script("rb"), None, None])
    record_case("UnlockPickup-reseed", mk("MiniGrid-UnlockPickup-v0"), list(range(4)), 650, scripts=[unlock_script(True)] * 2 + [None] * 2)
    record_case("Unlock-reseed", mk("MiniGrid-Unlock-v0"), list(range(4)), 650, scripts=[unlock_script(False)] * 2 + [None] * 2)
    record_case("KeyCorridorS3R3-reseed", mk("MiniGrid-KeyCorridorS3R3-v0"), list(range(4)), 600, scripts=[keycorridor_script] * 2 + [None] * 2)
    record_case("PutNear-6x6-N2-reseed", mk("MiniGrid-PutNear-6x6-N2-v0"), list(range(6)), 200, scripts=[putnear_script(1), putnear_script(0)] * 2 + [None] * 2)
    record_case("MemoryS7-reseed", mk("MiniGrid-MemoryS7-v0"), list(range(4)), 450, scripts=[memory_script(0), memory_script(1), None, None])
    record_case("LockedRoom-reseed", mk("MiniGrid-LockedRoom-v0"), list(range(3)), 420)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        record_case("TwoGoals-8x8-reseed", mk("MiniGrid-TwoGoals-8x8-v0"), list(range(4)), 300, scripts=[twogoals_script] * 2 + [None] * 2)
    print((buf.getvalue().strip().splitlines() or [""])[-1])
    # plain reference semantics at the episode boundary: reset() WITHOUT re-seeding (a new level every episode)
    record_case("LavaCrossingS9N1-stream", mk("MiniGrid-LavaCrossingS9N1-v0"), list(range(6)), 500, reseed=False)
    record_case("DoorKey-5x5-stream", mk("MiniGrid-DoorKey-5x5-v0"), [0, 1, 2], 800, reseed=False)
    record_case("LavaGapS6-stream", mk("MiniGrid-LavaGapS6-v0"), [0, 1, 2, 3], 400, reseed=False)
    record_case("Empty-Random-6x6-stream", mk("MiniGrid-Empty-Random-6x6-v0"), [0, 1], 450, reseed=False)
    # round 4: the two episode boundaries the C ABI lacked.  (1) plain caller-side reset() without seed() on the BASELINE families and on
    # Dynamic-Obstacles, whose step() itself draws from the stream the next level continues (minigrid.py:836-839, run_tests.py:64-66);
    record_case("DoorKey-8x8-stream", mk("MiniGrid-DoorKey-8x8-v0"), list(range(6)), 1400, scripts=[doorkey_script] * 3 + [None] * 3, reseed=False)
    record_case("DynObs-8x8-stream", mk("MiniGrid-Dynamic-Obstacles-8x8-v0"), list(range(10)), 400, reseed=False, n_actions=4, gym_id="MiniGrid-Dynamic-Obstacles-8x8-v0")
    record_case("DynObs-Random-6x6-stream", mk("MiniGrid-Dynamic-Obstacles-Random-6x6-v0"), list(range(8)), 300, reseed=False, n_actions=4, gym_id="MiniGrid-Dynamic-Obstacles-Random-6x6-v0")
    record_case("DynObs-16x16-stream", mk("MiniGrid-Dynamic-Obstacles-16x16-v0"), list(range(3)), 400, reseed=False, n_actions=4, gym_id="MiniGrid-Dynamic-Obstacles-16x16-v0")
    # (2) the reference's ReseedWrapper with a list of K > 1 seeds (wrappers.py:12-28), recorded through the wrapper class itself
    sl = lambda K, L, base: [[base + 10 * k + j for j in range(L)] for k in range(K)]  # noqa: E731
    record_case("Empty-Random-6x6-seedlist", mk("MiniGrid-Empty-Random-6x6-v0"), None, 450, seed_lists=sl(4, 3, 100), gym_id="MiniGrid-Empty-Random-6x6-v0")
    record_case("DoorKey-8x8-seedlist", mk("MiniGrid-DoorKey-8x8-v0"), None, 1400, scripts=[doorkey_script] * 3 + [None] * 3, seed_lists=sl(6, 3, 200), gym_id="MiniGrid-DoorKey-8x8-v0")
    record_case("LavaCrossingS9N1-seedlist", mk("MiniGrid-LavaCrossingS9N1-v0"), None, 500, seed_lists=sl(6, 4, 300), seed_idx0=2, gym_id="MiniGrid-LavaCrossingS9N1-v0")
    record_case("Fetch-8x8-N3-seedlist", mk("MiniGrid-Fetch-8x8-N3-v0"), None, 500, scripts=[fetch_script(0), fetch_script(1)] * 2 + [None] * 2, seed_lists=sl(6, 3, 400), gym_id="MiniGrid-Fetch-8x8-N3-v0")
    record_case("ObstructedMaze-1Dlhb-seedlist", mk("MiniGrid-ObstructedMaze-1Dlhb-v0"), None, 700, scripts=[obstructed_script] * 4 + [None] * 2, seed_lists=sl(6, 2, 500),
                objstate=True, gym_id="MiniGrid-ObstructedMaze-1Dlhb-v0")
    record_case("DynObs-8x8-seedlist", mk("MiniGrid-Dynamic-Obstacles-8x8-v0"), None, 400, seed_lists=sl(10, 3, 600), seed_idx0=1, n_actions=4, gym_id="MiniGrid-Dynamic-Obstacles-8x8-v0")
    record_case("DynObs-16x16-seedlist", mk("MiniGrid-Dynamic-Obstacles-16x16-v0"), None, 300, seed_lists=sl(3, 2, 700), n_actions=4, gym_id="MiniGrid-Dynamic-Obstacles-16x16-v0")
    # round 4: the exploration-bonus wrappers (wrappers.py:87-153) around the env; names start with "Bonus-" (tests/test_bonus.py, test_gpu_bonus.py)
    record_case("Bonus-DoorKey-8x8-action", mk("MiniGrid-DoorKey-8x8-v0"), list(range(6)), 900, scripts=[doorkey_script] * 3 + [None] * 3,
                bonus=("action",), gym_id="MiniGrid-DoorKey-8x8-v0")
    record_case("Bonus-LavaCrossingS9N1-state", mk("MiniGrid-LavaCrossingS9N1-v0"), list(range(6)), 500, bonus=("state",), gym_id="MiniGrid-LavaCrossingS9N1-v0")
    record_case("Bonus-DoorKey-5x5-state-action", mk("MiniGrid-DoorKey-5x5-v0"), list(range(6)), 600, scripts=[doorkey_script] * 4 + [None] * 2,
                bonus=("state", "action"), gym_id="MiniGrid-DoorKey-5x5-v0")     # ActionBonus(StateBonus(env))
    record_case("Bonus-Empty-Random-6x6-action-state-stream", mk("MiniGrid-Empty-Random-6x6-v0"), list(range(4)), 500, reseed=False,
                bonus=("action", "state"), gym_id="MiniGrid-Empty-Random-6x6-v0")   # StateBonus(ActionBonus(env)), a new level per episode
    record_case("Bonus-Soup-8x8-strafe-action", lambda: SoupEnv(8, 8, False, 96, 0.45, extended=True), list(range(4)), 300, n_actions=9, bonus=("action",))
    record_case("Bonus-MemoryS13Random-action", mk("MiniGrid-MemoryS13Random-v0"), list(range(4)), 300, bonus=("action",), gym_id="MiniGrid-MemoryS13Random-v0")
    record_case("Bonus-TwoGoals-8x8-state", mk("MiniGrid-TwoGoals-8x8-v0"), list(range(4)), 300, bonus=("state",), gym_id="MiniGrid-TwoGoals-8x8-v0")
    record_case("Bonus-Empty-16x16-full-state", mk("MiniGrid-Empty-16x16-v0"), [0, 1], 1100, full_obs=True, bonus=("state",), gym_id="MiniGrid-Empty-16x16-v0")
    # the fork's DACWrapper (names start with "Dac-"; tests/test_oracle_golden.py, tests/test_gpu_bonus.py)
    record_case("Dac-LavaCrossingS9N1", mk("MiniGrid-LavaCrossingS9N1-v0"), list(range(6)), 700, bonus=("dac",), gym_id="MiniGrid-LavaCrossingS9N1-v0")
    record_case("Dac-Fetch-5x5-N2", mk("MiniGrid-Fetch-5x5-N2-v0"), list(range(6)), 300, bonus=("dac",), gym_id="MiniGrid-Fetch-5x5-N2-v0")
    record_case("Dac-LavaGapS7-state", mk("MiniGrid-LavaGapS7-v0"), list(range(4)), 500, bonus=("dac", "state"), gym_id="MiniGrid-LavaGapS7-v0")   # StateBonus(DACWrapper(env))
    if only:
        return
    record_levels()
    record_level_streams()
    record_onehot()
    record_flat()


if __name__ == "__main__":
    main()
