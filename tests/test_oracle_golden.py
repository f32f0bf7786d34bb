"""Pins the CPU oracle (oracle/minigrid_oracle.c) to the reference: every golden
trace recorded from the reference (oracle/gen_golden.py) is replayed through the
oracle and every output byte, reward, done flag and post-step state must match."""
import numpy as np
import pytest

from conftest import bonus_cases, dac_cases, golden_cases, load_case
from oracle.bonus_oracle import BonusOracle, DacOracle
from oracle.minigrid_oracle import OracleEnvs


@pytest.mark.parametrize("name", golden_cases() + bonus_cases())
def test_oracle_replays_reference_trace(name):
    """(`Bonus-*`: the same through the restatement of the ActionBonus / StateBonus wrappers, oracle/bonus_oracle.py, on top.)"""
    meta, z = load_case(name)
    K, T = z["actions"].shape
    bonus = BonusOracle(K, meta["W"], meta["H"], meta["bonus"], 9 if meta.get("extended") else 7) if meta.get("bonus") else None
    full = meta["full_obs"]
    env = OracleEnvs(meta["W"], meta["H"], meta["max_steps"], meta["see_through"], meta["lava_v1"], view=meta.get("view", 7), extended=meta.get("extended", False), alt_vis=meta.get("alt_vis", False), task=meta.get("task", 0))
    env.set_state(z["init_grid"], z["init_agent"], aux=z["init_aux"])
    if meta.get("task", 0):
        env.task = z["init_task"].copy()
    objstate = meta.get("objstate", False)
    if objstate:
        env.set_contains(z["init_contains"])
    if full:
        o, f = env.observe(full=True)
        assert np.array_equal(f, z["init_full"])
    else:
        o = env.observe()
    assert np.array_equal(o, z["init_obs"])
    resets = {}
    for r, (k, t) in enumerate(zip(z["reset_k"], z["reset_t"])):
        resets.setdefault(int(t), []).append((int(k), r))
    for t in range(T):
        out = env.step(z["actions"][:, t], full=full)
        if full:
            obs, fo, rew, done = out
            assert np.array_equal(fo, z["full"][:, t]), (name, t)
        else:
            obs, rew, done = out
        assert (env.err == 0).all()
        if bonus:
            rew = bonus.step(rew, env.agent, z["actions"][:, t])
        assert np.array_equal(obs, z["obs"][:, t]), (name, t)
        assert np.array_equal(rew, z["reward"][:, t]), (name, t)       # float64, exact
        assert np.array_equal(done, z["done"][:, t]), (name, t)
        assert np.array_equal(env.agent, z["agent"][:, t]), (name, t)
        assert np.array_equal(env.agent[:, 2], z["direction"][:, t]), (name, t)
        assert np.array_equal(env.carry, z["carry"][:, t]), (name, t)
        assert np.array_equal(env.steps, z["steps"][:, t]), (name, t)
        assert np.array_equal(env.grid, z["grid"][:, t]), (name, t)
        if objstate:
            assert np.array_equal(env.aux, z["aux"][:, t]), (name, t)
            assert np.array_equal(env.contains, z["contains"][:, t]), (name, t)
            assert np.array_equal(env.carry_aux, z["carry_aux"][:, t]), (name, t)
            assert np.array_equal(env.carry_contains, z["carry_contains"][:, t]), (name, t)
        # caller-side reset on done, same seed -> the recorded reset state is the episode start
        assert sorted(k for k, _ in resets.get(t, [])) == sorted(np.flatnonzero(done).tolist())
        for k, r in resets.get(t, []):
            if meta.get("reseed", True) is True:  # same seed -> the recorded reset state is the episode start
                assert np.array_equal(z["reset_grid"][r], z["init_grid"][k])
                assert np.array_equal(z["reset_aux"][r], z["init_aux"][k])
                assert np.array_equal(z["reset_agent"][r], z["init_agent"][k])
                assert np.array_equal(z["reset_obs"][r], z["init_obs"][k])
                assert z["reset_task"][r] == z["init_task"][k]
                if full:
                    assert np.array_equal(z["reset_full"][r], z["init_full"][k])
                if objstate:
                    assert np.array_equal(z["reset_contains"][r], z["init_contains"][k])
            else:                         # the RNG stream continued / ReseedWrapper moved on to the next seed of its list
                                          # (tests/test_episode_boundary.py pins those levels): injected from the recording
                env.grid0[k], env.aux0[k], env.agent0[k] = z["reset_grid"][r], z["reset_aux"][r], z["reset_agent"][r]
                if meta.get("task", 0):
                    env.task[k] = z["reset_task"][r]
                if objstate:
                    env.contains0[k] = z["reset_contains"][r]
        env.reset_where(done)
        if done.any():
            o = env.observe(full=True) if full else env.observe()
            for k, r in resets.get(t, []):
                assert np.array_equal((o[0] if full else o)[k], z["reset_obs"][r])
                if full:
                    assert np.array_equal(o[1][k], z["reset_full"][r])   # FullyObsWrapper image returned by the reference's reset()


STATE = ("grid", "aux", "agent", "carry", "steps", "task")


@pytest.mark.parametrize("name", dac_cases())
def test_oracle_replays_dac_trace(name):
    """The fork's DACWrapper (wrappers.py:35-84) around the env: the oracle steps only the envs the wrapper still steps (the others keep
    their state), oracle/bonus_oracle.py's DacOracle turns what they return into the wrapper's obs / reward / done; every byte of the
    recorded trace, resets through the wrapper included."""
    meta, z = load_case(name)
    K, T = z["actions"].shape
    full = meta["full_obs"]
    env = OracleEnvs(meta["W"], meta["H"], meta["max_steps"], meta["see_through"], meta["lava_v1"], task=meta.get("task", 0))
    env.set_state(z["init_grid"], z["init_agent"], aux=z["init_aux"])
    if meta.get("task", 0):
        env.task = z["init_task"].copy()
    dac = DacOracle(K, meta["max_steps"])
    outer = [b for b in meta["bonus"][meta["bonus"].index("dac") + 1:]]
    assert meta["bonus"][0] == "dac"
    bonus = BonusOracle(K, meta["W"], meta["H"], outer) if outer else None
    want_obs = z["full"] if full else z["obs"]
    for t in range(T):
        keep = ~dac.stepping()
        state = [k for k in STATE if getattr(env, k) is not None]
        saved = {k: getattr(env, k)[keep].copy() for k in state}
        out = env.step(z["actions"][:, t], full=full)
        for k in state:
            getattr(env, k)[keep] = saved[k]
        obs, rew, done = (out[1], out[2], out[3]) if full else out
        obs, rew, done = dac.step(obs, rew, done)
        if bonus:
            rew = bonus.step(rew, env.agent, z["actions"][:, t])
        assert np.array_equal(obs, want_obs[:, t]), (name, t)
        assert np.array_equal(rew, z["reward"][:, t]), (name, t)
        assert np.array_equal(done, z["done"][:, t]), (name, t)
        assert np.array_equal(env.agent, z["agent"][:, t]) and np.array_equal(env.grid, z["grid"][:, t]) and np.array_equal(env.steps, z["steps"][:, t]), (name, t)
        if done.any():   # the recorder's `env.seed(s); wrapper.reset()`: the episode start again
            env.reset_where(done)
            dac.reset_where(done)
    assert z["done"].sum() >= 2 * K and (want_obs.reshape(K, T, -1) == 1).all(-1).sum() > K * 50
