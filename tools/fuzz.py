#!/usr/bin/env python3
"""Configuration fuzz on the GPU box (not part of the test suite): random grid sizes, view sizes, visibility rules, action
sets, obs modes, batch sizes and kernel forms, each stepped against the CPU oracle on random rooms, every byte compared.

    python tools/fuzz.py [trials] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import gym_minigrid_amd as mg  # noqa: E402
from oracle.minigrid_oracle import OracleEnvs  # noqa: E402  (checker only)
from helpers import random_states, random_object_state, to_np  # noqa: E402


def one(rs, trial):
    W = int(rs.choice([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 19, 22, 25, 31, 40, 57, 90, 255]))
    H = int(rs.choice([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 19, 22, 25, 31, 40, 57, 90, 255])) if rs.uniform() < 0.5 else W
    view = int(rs.choice([3, 5, 7, 7, 7, 9, 11]))
    see, v1, ext, alt = (bool(rs.randint(2)) for _ in range(4))
    mode = str(rs.choice(["partial", "partial", "full"]))
    objstate = rs.uniform() < 0.2
    auto = bool(rs.randint(2)) and not objstate
    N = int(rs.choice([1, 63, 64, 65, 200, 777, 2048 + 5])) if W * H <= 700 else int(rs.choice([1, 65, 130]))
    max_steps = int(rs.randint(4, 40))
    T = 40
    forms = {}
    if mode == "partial" and rs.uniform() < 0.4:   # (round 3: every view size / visibility rule / object-state handle has both forms)
        forms["MGX_PARTIAL_KERNEL"] = str(rs.choice(["staged", "gather"]))
    if mode == "full" and rs.uniform() < 0.4:
        forms["MGX_FULL_KERNEL"] = str(rs.choice(["lds", "direct"]))
    if forms.get("MGX_PARTIAL_KERNEL") == "staged" and W * H > 2400:
        forms.pop("MGX_PARTIAL_KERNEL")          # a 64-env tile of that size does not fit the LDS: the rule would refuse it
    if forms.get("MGX_FULL_KERNEL") == "lds" and W * H > 2300:
        forms.pop("MGX_FULL_KERNEL")
    desc = dict(W=W, H=H, view=view, see=see, v1=v1, ext=ext, alt=alt, mode=mode, objstate=objstate, auto=auto, N=N, max_steps=max_steps, **forms)
    for k in ("MGX_PARTIAL_KERNEL", "MGX_FULL_KERNEL"):
        os.environ.pop(k, None)
    os.environ.update(forms)
    if W >= 5 and H >= 5:
        grid, aux, agent, carry, steps = random_states(N, W, H, seed=trial, density=float(rs.choice([0.15, 0.4, 0.6])))
    else:
        grid, aux, agent, carry, steps = random_states(N, W, H, seed=trial, density=0.0)
    contains = None
    if objstate:
        aux, contains = random_object_state(grid, seed=trial)
        carry, steps = None, None
    orc = OracleEnvs(W, H, max_steps, see, v1, view=view, extended=ext, alt_vis=alt)
    orc.set_state(grid, agent, aux=aux, carry=carry, steps=steps)
    if objstate:
        orc.set_contains(contains)
    cfg = mg.Config()
    cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1 = W, H, max_steps, int(see), int(v1)
    env = mg.VecMiniGrid(config=cfg, num_envs=N, obs_mode=mode, auto_reset=auto, backend="torch", agent_view_size=view,
                         extended_actions=ext, default_vis=not alt, object_state=objstate)
    env.set_state(grid, agent, aux=aux, carry=carry, steps=steps)
    if objstate:
        env.set_object_state(contains=contains)
    full = mode == "full"
    assert np.array_equal(to_np(env.observe()), orc.observe(True)[int(full)]), desc
    nact = 9 if ext else 7
    faults = 0
    for t in range(T):
        a = rs.randint(0, nact, size=N).astype(np.uint8)
        obs, rew, done, _ = env.step(a)
        oo, of, orew, odone = orc.step(a, True)
        faults += int((orc.err != 0).sum())   # e.g. the reference's strafe_right-onto-goal AttributeError (minigrid.py:1310)
        want = (of if full else oo).copy()
        d = odone.astype(bool)
        if auto:
            orc.reset_where(odone)
            want[d] = orc.observe(True)[int(full)][d]
        assert np.array_equal(to_np(done), odone), (desc, t)
        assert np.array_equal(to_np(rew), orew.astype(np.float32)), (desc, t)
        assert np.array_equal(to_np(obs), want), (desc, t)
        if not auto and d.any():
            orc.reset_where(odone)
            env.set_state(orc.grid, orc.agent, aux=orc.aux, carry=orc.carry, steps=orc.steps)
            if objstate:
                env.set_object_state(contains=orc.contains, carry_aux=orc.carry_aux, carry_contains=orc.carry_contains)
        if t % 13 == 12 or t == T - 1:
            st = env.get_state()
            assert np.array_equal(st["grid"], orc.grid) and np.array_equal(st["agent"], orc.agent), (desc, t)
            assert np.array_equal(st["carry"], orc.carry) and np.array_equal(st["steps"], orc.steps) and np.array_equal(st["aux"], orc.aux), (desc, t)
            assert np.array_equal(to_np(env.pose()), orc.agent), (desc, t)
    try:
        env.sync()
        assert faults == 0, (desc, faults)
    except mg.OutOfBounds:
        assert faults > 0, desc
        env.clear_faults()
    env.close()
    return desc, N * T


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    total, t0 = 0, time.perf_counter()
    for trial in range(trials):
        desc, n = one(rs, seed * 100000 + trial)
        total += n
        if trial % 10 == 9:
            print("trial %4d ok (%.0f s): last %s" % (trial + 1, time.perf_counter() - t0, desc), flush=True)
    print("fuzz ok: %d configurations, %d env-steps, every byte equal" % (trials, total), flush=True)


if __name__ == "__main__":
    main()
