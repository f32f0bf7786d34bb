"""The launch-shaping rules of the step kernel (csrc/k_step.hip: raised wave priority for the grid's last blocks, staggered start of
the first round's waves by SIMD slot, guard bands around the LDS tile images) change WHEN waves run, never what they compute.
Pinned twice against the CPU oracle: at the size where the rules switch on by themselves (more than one round of blocks), and
in a child process that forces them onto a small batch through the tuning overrides (MGX_STAGGER_MIN / MGX_STAGGER / MGX_TAIL_BLOCKS /
MGX_WPB are read once per process; they exist in -DMGX_TUNING builds only, so the second half runs under MGX_LIB=<such a build>)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_against_oracle(env_id, n, T, seed):
    import gym_minigrid_amd as mg
    from helpers import make_oracle
    cfg = mg.env_config(env_id)
    E = mg.VecMiniGrid(env_id, num_envs=n, seeds=seed, auto_reset=True, backend="torch")
    obs = E.reset()
    gidx = np.arange(n)
    grid, agent = mg.generate_levels(env_id, (seed + gidx).astype(np.uint64))
    orc = make_oracle(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1, grid, np.zeros(grid.shape[:3], np.uint8), agent)
    assert np.array_equal(obs.cpu().numpy(), orc.observe())
    acts = E.fill_actions(11, 0, T)
    dones = 0
    for t in range(T):
        obs, rew, done, _ = E.step(acts[t])
        a = mg.action_stream(11, gidx, t)
        o_obs, o_rew, o_done = orc.step(a)
        orc.reset_where(o_done)
        fresh = orc.observe()
        want = np.where(o_done.astype(bool)[:, None, None, None], fresh, o_obs)
        assert np.array_equal(obs.cpu().numpy(), want), t
        assert np.array_equal(done.cpu().numpy(), o_done) and np.array_equal(rew.cpu().numpy(), o_rew.astype(np.float32)), t
        dones += int(o_done.sum())
    st = E.get_state()
    assert np.array_equal(st["grid"], orc.grid) and np.array_equal(st["agent"], orc.agent)
    E.close()
    return dones


def test_rules_switch_on_above_one_round_of_blocks():
    # 2,560 blocks of 4 tiles > one round (1,792 resident blocks of k_step<9,9>): stagger and tail priority are both active
    assert _run_against_oracle("MiniGrid-LavaCrossingS9N1-v0", 655360, 10, 3) > 10000


@pytest.mark.parametrize("overrides", [
    {"MGX_STAGGER_MIN": "0", "MGX_STAGGER": "3", "MGX_TAIL_BLOCKS": "5"},
    {"MGX_STAGGER_MIN": "0", "MGX_STAGGER": "1", "MGX_TAIL_BLOCKS": "2", "MGX_WPB": "1"},
    {"MGX_STAGGER": "0", "MGX_TAIL_BLOCKS": "0"},
])
def test_forced_rules_on_a_small_batch(overrides):
    from gym_minigrid_amd import _lib
    if b"tuning" not in _lib.lib().mgx_version():
        pytest.skip("the default library is built without the tuning knobs (-DMGX_TUNING: tools/build_variant.sh tuning -DMGX_TUNING; MGX_LIB=ab/tuning.so)")
    env = dict(os.environ, **overrides)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "child ok" in r.stdout


if __name__ == "__main__" and sys.argv[1:] == ["child"]:
    _here = os.path.dirname(os.path.abspath(__file__))
    for _p in (_here, os.path.dirname(_here), os.path.join(os.path.dirname(_here), "gym-minigrid_amd")):  # as tests/conftest.py does
        sys.path.insert(0, _p)
    d1 = _run_against_oracle("MiniGrid-DoorKey-8x8-v0", 6000, 150, 5)
    d2 = _run_against_oracle("MiniGrid-LavaCrossingS9N1-v0", 5000, 60, 1)
    d3 = _run_against_oracle("MiniGrid-FourRooms-v0", 3000, 40, 2)  # 19x19: the gather form of k_step
    print("child ok", d1, d2, d3)
