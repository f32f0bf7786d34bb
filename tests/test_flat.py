"""FlatObsWrapper epilogue (wrappers.py:528-577).  CPU: the numpy restatement and the mission
strings of mgx_mission are pinned to FlatObsWrapper outputs recorded from the reference.  GPU: k_flat against the
recorded outputs on the same seeds and actions, and against the restatement on random batches."""
import ctypes
import os

import numpy as np
import pytest

import gym_minigrid_amd as mg
from gym_minigrid_amd import _lib
from conftest import GOLDEN
from helpers import make_oracle, random_states, to_np
from oracle.minigrid_oracle import flat_obs


def episodes():
    z = np.load(os.path.join(GOLDEN, "flat.npz"))
    for k in range(len(z["ids"])):
        yield str(z["ids"][k]), int(z["seeds"][k]), bool(z["full"][k]), z["actions"][k], str(z["missions"][k]), z["flat_%d" % k]


def mission_of(cfg, task):
    buf = ctypes.create_string_buffer(128)
    n = _lib.lib().mgx_mission(ctypes.byref(cfg), int(task), buf, 128)
    assert n >= 0
    return buf.value.decode()


def test_numpy_restatement_matches_reference_wrapper():
    n = 0
    for env_id, seed, full, acts, mission, flat in episodes():
        img = flat.shape[1] - 27 * 96
        for row in flat:
            assert row.dtype == np.float32
            assert np.array_equal(flat_obs(row[:img].astype(np.uint8), mission), row)
        n += 1
    assert n >= 30


def test_mission_strings_match_reference():
    seen = set()
    for env_id, seed, full, acts, mission, flat in episodes():
        cfg = mg.env_config(env_id)
        _, _, task = mg.generate_levels(env_id, [seed], with_task=True)
        assert mission_of(cfg, task[0]) == mission, (env_id, seed)
        seen.add(mission)
    assert len(seen) >= 12            # Fetch templates x objects + the constant missions
    cfg = mg.env_config("MiniGrid-Fetch-5x5-N2-v0")
    buf = ctypes.create_string_buffer(8)
    assert _lib.lib().mgx_mission(ctypes.byref(cfg), 0x05, buf, 8) < 0      # buffer too small
    buf = ctypes.create_string_buffer(128)
    assert _lib.lib().mgx_mission(ctypes.byref(cfg), 0x08, buf, 128) < 0    # a goal is not a Fetch target


@pytest.mark.gpu
def test_flat_matches_reference_traces():
    groups = {}
    for ep in episodes():
        groups.setdefault((ep[0], ep[2]), []).append(ep)
    for (env_id, full), eps in groups.items():
        seeds = np.array([e[1] for e in eps], np.uint64)
        env = mg.VecMiniGrid(env_id, num_envs=len(eps), seeds=seeds, obs_mode="full_flat" if full else "flat", backend="numpy")
        assert env.observation_space.shape == (1, eps[0][5].shape[1])
        obs = env.reset()
        assert env.missions() == [e[4] for e in eps]
        assert obs.dtype == np.float32
        assert np.array_equal(obs, np.stack([e[5][0] for e in eps])), env_id
        for t in range(len(eps[0][3])):
            obs, _, done, _ = env.step(np.array([e[3][t] for e in eps], np.uint8))
            assert not done.any()
            assert np.array_equal(obs, np.stack([e[5][t + 1] for e in eps])), (env_id, t)
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,W,H,view,N", [("flat", 8, 8, 7, 64 * 3 + 5), ("flat", 9, 7, 5, 70), ("full_flat", 6, 6, 7, 129), ("flat", 5, 5, 7, 1)])
def test_flat_epilogue_random(mode, W, H, view, N):
    T, max_steps = 12, 9
    grid, aux, agent, carry, steps = random_states(N, W, H, seed=W + view, density=0.4)
    orc = make_oracle(W, H, max_steps, False, False, grid, aux, agent, carry, steps)
    orc.cfg.view, orc.V = view, view
    c = mg.Config()
    c.width, c.height, c.max_steps = W, H, max_steps
    c.level_kind = 2                                       # DoorKey's mission string on caller-supplied states
    env = mg.VecMiniGrid(config=c, num_envs=N, obs_mode=mode, auto_reset=True, backend="torch", agent_view_size=view)
    env.set_state(grid, agent, aux=aux, carry=carry, steps=steps)
    full = mode == "full_flat"
    flat = lambda imgs: np.stack([flat_obs(i, env.mission) for i in imgs])  # noqa: E731
    assert env.mission == "use the key to open the door and then get to the goal"
    assert np.array_equal(to_np(env.observe()), flat(orc.observe(full=True)[1 if full else 0]))
    rs = np.random.RandomState(1)
    for t in range(T):
        a = rs.randint(0, 7, size=N).astype(np.uint8)
        obs, rew, done, _ = env.step(a)
        oo, of, orew, odone = orc.step(a, full=True)
        orc.reset_where(odone)
        ro = orc.observe(full=True)
        want = np.where(odone.astype(bool)[:, None, None, None], ro[1 if full else 0], of if full else oo)
        assert np.array_equal(to_np(obs), flat(want)), t
    env.close()
