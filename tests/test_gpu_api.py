"""C-ABI behaviours that the Python mirror does not exercise by itself: optional outputs, streams, graphs, stats."""
import ctypes

import numpy as np
import pytest
import torch

import gym_minigrid_amd as mg
from gym_minigrid_amd import _lib
from helpers import make_oracle, random_states, to_np

pytestmark = pytest.mark.gpu


def test_optional_outputs_and_raw_calls():
    """obs / reward / done may each be NULL; state still advances identically."""
    N, W, H = 200, 8, 8
    grid, aux, agent, carry, steps = random_states(N, W, H, seed=4)
    acts = np.random.RandomState(0).randint(0, 7, size=(30, N)).astype(np.uint8)
    ref = make_oracle(W, H, 15, False, False, grid, aux, agent, carry, steps)
    for mode in ("partial", "full"):
        c = mg.Config()
        c.width, c.height, c.max_steps = W, H, 15
        env = mg.VecMiniGrid(config=c, num_envs=N, obs_mode=mode, auto_reset=True, backend="numpy")
        env.set_state(grid, agent, aux=aux, carry=carry, steps=steps)
        orc = make_oracle(W, H, 15, False, False, grid, aux, agent, carry, steps)
        L = _lib.lib()
        done = np.zeros(N, np.uint8)
        for t in range(30):
            a = acts[t]
            if t % 3 == 0:    # nothing but the transition
                _lib.check(L.mgx_step(env._h, a.ctypes.data, None, None, None))
            elif t % 3 == 1:  # done only
                _lib.check(L.mgx_step(env._h, a.ctypes.data, None, None, done.ctypes.data))
            else:
                env.step(a)
            oo, orew, odone = orc.step(a)
            orc.reset_where(odone)
            if t % 3 == 1:
                assert np.array_equal(done, odone)
        st = env.get_state()
        assert np.array_equal(st["grid"], orc.grid) and np.array_equal(st["agent"], orc.agent) and np.array_equal(st["steps"], orc.steps)
        assert env.stats()["steps"] == 30 * N
        env.close()
    del ref


def test_hipgraph_replay_equals_eager():
    """mgx_step does no allocation / synchronisation with device pointers, so it can be captured in a hipGraph."""
    N, T = 4096, 12
    env = mg.VecMiniGrid("MiniGrid-DoorKey-8x8-v0", num_envs=N, seeds=0)
    ref = mg.VecMiniGrid("MiniGrid-DoorKey-8x8-v0", num_envs=N, seeds=0)
    env.reset(); ref.reset()
    acts = env.fill_actions(1, 0, T)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        env.step(acts[0])
    ref.step(acts[0])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for t in range(1, T):
            env.step(acts[t])
    g.replay()
    torch.cuda.synchronize()
    for t in range(1, T):
        ref.step(acts[t])
    torch.cuda.synchronize()
    assert torch.equal(env._obs, ref._obs) and torch.equal(env._done, ref._done) and torch.equal(env._reward, ref._reward)
    a, b = env.get_state(), ref.get_state()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    env.close(); ref.close()


def test_two_handles_two_streams_and_stats():
    N = 3000
    e1 = mg.VecMiniGrid("MiniGrid-LavaCrossingS9N1-v0", num_envs=N, seeds=0)
    e2 = mg.VecMiniGrid("MiniGrid-LavaCrossingS9N1-v0", num_envs=N, seeds=0)
    e1.reset(); e2.reset()
    acts = e1.fill_actions(3, 0, 50)
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    dones = 0
    for t in range(50):
        o1, r1, d1, _ = e1.step(acts[t])
        with torch.cuda.stream(s2):
            o2, r2, d2, _ = e2.step(acts[t])
        dones += int(d1.sum().item())
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(d1, d2)
    st = e1.stats()
    assert st["episodes"] == dones == e2.stats()["episodes"] and st["steps"] == 50 * N
    out2 = torch.zeros(2, dtype=torch.float64, device=o1.device)
    e1.read_stats_async(out2)
    torch.cuda.synchronize()
    assert out2.tolist() == [float(dones), st["reward_sum"]]
    e1.close(); e2.close()


@pytest.mark.parametrize("env_id,N,T", [("MiniGrid-DoorKey-8x8-v0", 4096, 16), ("MiniGrid-LavaCrossingS9N1-v0", 1024, 24),
                                        ("MiniGrid-Dynamic-Obstacles-6x6-v0", 512, 12)])
def test_rollout_graph_equals_stepping(env_id, N, T):
    """mgx_rollout (T steps captured into one hipGraph, replayed) == T mgx_step calls, on every output byte; the replay
    of the cached graph continues the episodes."""
    seeds = np.arange(N, dtype=np.uint64)
    a_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch", new_level_each_episode=("Lava" in env_id))
    b_env = mg.VecMiniGrid(env_id, num_envs=N, seeds=seeds, backend="torch", new_level_each_episode=("Lava" in env_id))
    a_env.reset(); b_env.reset()
    for rep in range(3):                                     # rep 0 captures, 1 and 2 replay the same graph
        acts = a_env.fill_actions(11, rep * T, T)            # (T, N) uint8 on the device
        acts_keep = acts if rep == 0 else acts_keep
        if rep:                                              # same buffer as the captured one: copy the new actions in
            acts_keep.copy_(acts)
        obs, rew, done = a_env.rollout(acts_keep)
        torch.cuda.synchronize()
        for t in range(T):
            o, r, d, _ = b_env.step(acts_keep[t])
            assert torch.equal(obs[t], o), (rep, t)
            assert torch.equal(rew[t], r) and torch.equal(done[t], d), (rep, t)
    sa, sb = a_env.stats(), b_env.stats()
    assert sa == sb and sa["steps"] == 3 * T * N
    with pytest.raises(mg.MgxError):
        host = np.zeros((T, N), np.uint8)
        _lib.check(_lib.lib().mgx_rollout(a_env._h, T, host.ctypes.data, None, None, None))
    a_env.close(); b_env.close()
