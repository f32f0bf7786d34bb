"""The two episode boundaries of the reference besides `seed(s); reset()` (CPU side; the HIP path: tests/test_gpu_episode_boundary.py):

  * the plain caller-side `reset()` (minigrid.py:831-858, caller loop run_tests.py:64-66): the env's RNG stream continues;
  * `ReseedWrapper(env, seeds=[s0..sK-1], seed_idx)` (wrappers.py:12-28): every reset() takes the next seed of the list.

The fixtures were recorded from the reference (oracle/gen_golden.py: the second kind through the wrapper class itself).  Here the
levels the reference drew at every recorded boundary are pinned to the host generator (libmgx's levelgen.cpp: no GPU involved), and
the Dynamic-Obstacles restatement -- whose step() draws from the same stream -- replays its stream / seed-list traces."""
import numpy as np
import pytest

import gym_minigrid_amd as mg
from conftest import GOLDEN, golden_cases, load_case
from oracle.dynobs_oracle import DynObsOracle

import os


def cases(kind):
    out = []
    for f in sorted(os.listdir(GOLDEN)):
        if not f.endswith(".npz") or f.startswith(("Bonus-", "Dac-")) or f in ("levels.npz", "level_streams.npz", "onehot.npz", "flat.npz", "levels_obstructed.npz"):
            continue
        meta, _ = load_case(f[:-4])
        if meta.get("reseed", True) == kind:
            out.append(f[:-4])
    return out


def gym_id_of(name, meta):
    if meta.get("gym_id"):
        return meta["gym_id"]
    base = name[:-7] if name.endswith("-stream") else name
    return "MiniGrid-%s-v0" % base


def episode_seeds(meta, z):
    """seed of the episode that starts at recorded reset r (ReseedWrapper: the list index advances once per reset() of its env)."""
    L = z["seed_list"].shape[1]
    nth = {}
    out = []
    for k in z["reset_k"]:
        j = nth.get(int(k), 0) + 1                      # the first reset() of the trace took entry seed_idx0
        nth[int(k)] = j
        out.append(int(z["seed_list"][k, (meta["seed_idx0"] + j) % L]))
    return out


def test_fixture_inventory():
    assert {"DoorKey-8x8-stream", "DynObs-8x8-stream", "LavaCrossingS9N1-stream", "Empty-Random-6x6-stream", "Fetch-8x8-N3",
            "ObstructedMaze-1Dlhb"} <= set(cases(False))
    assert set(cases("list")) == {"Empty-Random-6x6-seedlist", "DoorKey-8x8-seedlist", "LavaCrossingS9N1-seedlist", "Fetch-8x8-N3-seedlist",
                                  "ObstructedMaze-1Dlhb-seedlist", "DynObs-8x8-seedlist", "DynObs-16x16-seedlist"}


@pytest.mark.parametrize("name", cases("list"))
def test_seed_list_levels_match_host_generator(name):
    """Every level the reference's ReseedWrapper produced is `generate_levels(seed of that list entry)`."""
    meta, z = load_case(name)
    gid = gym_id_of(name, meta)
    L = z["seed_list"].shape[1]
    assert L > 1 and np.array_equal(z["seed"], z["seed_list"][:, meta["seed_idx0"]])
    out = mg.generate_levels(gid, z["seed"].astype(np.uint64), with_task=True, with_contains=True)
    assert np.array_equal(out[0], z["init_grid"]) and np.array_equal(out[1], z["init_agent"])
    seeds = episode_seeds(meta, z)
    assert len(set(seeds)) > 1
    grid, agent, task, cont = mg.generate_levels(gid, np.array(seeds, np.uint64), with_task=True, with_contains=True)
    assert np.array_equal(grid, z["reset_grid"]) and np.array_equal(agent, z["reset_agent"])
    if meta.get("task", 0) == 1:   # Fetch: the low byte names the target (the high one the mission template)
        assert np.array_equal(task & 0xFF, z["reset_task"])
    elif meta.get("task", 0) not in (0, 11):
        assert np.array_equal(task, z["reset_task"])
    if meta.get("objstate"):
        assert np.array_equal(cont, z["reset_contains"])


@pytest.mark.parametrize("name", [c for c in cases(False) if not c.startswith("DynObs-")])
def test_stream_levels_match_host_generator(name):
    """Level j + 1 of `generate_level_stream(seed)` is what the reference's j-th plain reset() of that trace drew."""
    meta, z = load_case(name)
    if meta["W"] * meta["H"] > 4096:
        pytest.skip("host-only grid size")
    gid = gym_id_of(name, meta)
    for k, s in enumerate(z["seed"]):
        rs = np.flatnonzero(z["reset_k"] == k)
        grid, agent = mg.generate_level_stream(gid, int(s), len(rs) + 1)
        assert np.array_equal(grid[0], z["init_grid"][k]) and np.array_equal(agent[0], z["init_agent"][k])
        assert np.array_equal(grid[1:], z["reset_grid"][rs]) and np.array_equal(agent[1:], z["reset_agent"][rs]), (name, k)


DYN = {"DynObs-8x8": (8, 4, False), "DynObs-Random-6x6": (6, 3, True), "DynObs-16x16": (16, 8, False)}


@pytest.mark.parametrize("name", [c for c in cases(False) + cases("list") if c.startswith("DynObs-")])
def test_dynobs_restatement_at_both_boundaries(name):
    """Dynamic-Obstacles: step() itself draws from the env's stream, so the level a plain reset() draws depends on every walk before
    it.  The restatement (numpy RandomState kept across the boundary / re-seeded from the list) against the reference's traces."""
    meta, z = load_case(name)
    size, n_obst, rnd = DYN[name.rsplit("-", 1)[0]]
    lst = meta["reseed"] == "list"
    K, T = z["actions"].shape
    o = DynObsOracle(size, n_obst, rnd, z["seed"], seed_lists=z["seed_list"] if lst else None, seed_idx=meta.get("seed_idx0", 0))
    assert np.array_equal(o.base.grid, z["init_grid"]) and np.array_equal(o.base.agent, z["init_agent"])
    assert np.array_equal(o.observe(), z["init_obs"])
    rmap = {(int(k), int(t)): r for r, (k, t) in enumerate(zip(z["reset_k"], z["reset_t"]))}
    for t in range(T):
        obs, r, d = o.step(z["actions"][:, t])
        assert np.array_equal(obs, z["obs"][:, t]), (name, t)
        assert np.array_equal(r, z["reward"][:, t]) and np.array_equal(d, z["done"][:, t]), (name, t)
        assert np.array_equal(o.base.grid, z["grid"][:, t]) and np.array_equal(o.base.agent, z["agent"][:, t]), (name, t)
        o.reset_where(d, reseed=lst)
        if d.any():
            ro = o.observe()
            for k in np.flatnonzero(d):
                rr = rmap[(int(k), t)]
                assert np.array_equal(o.base.grid[k], z["reset_grid"][rr]) and np.array_equal(o.base.agent[k], z["reset_agent"][rr]), (name, t, k)
                assert np.array_equal(ro[k], z["reset_obs"][rr]), (name, t, k)
    assert len(rmap) > 20
