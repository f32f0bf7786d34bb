#!/usr/bin/env python3
"""Timing of the reference's OWN CPU path (TEST / MEASUREMENT INFRASTRUCTURE, runs only in the build container).

BASELINE.json configs[0]: "MiniGrid-Empty-8x8-v0, 1 env, random-action rollout on reference CPU path (benchmark.py
plumbing)".  Imports the read-only reference at /root/reference under the stand-in `gym` of oracle/refshim/ (exactly as
oracle/gen_golden.py does), and for each of the four BASELINE config envs runs the loop SURVEY.md section 8(d) input 1
describes, with /root/reference/benchmark.py:20-53's plumbing (`gym.make`, `time` around a plain Python loop):

    actions = RandomState(0).randint(0, 7, size=T);  for a in actions: obs, r, done, _ = env.step(a);  if done: env.reset()

T = 20,000 steps, `time.perf_counter` around the loop, one process, one core.  Writes profiles/reference_cpu.json with
the CPU model and `nproc`; bench.py's `cpu_baseline.sample` quotes that file next to the C port's own number (the
reference cannot travel to the GPU box, so it is never timed there).

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference.py [--steps 20000] [--repeats 3]
"""
import argparse
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "refshim"))

import numpy as np  # noqa: E402
import gym  # noqa: E402
import gym_minigrid  # noqa: E402,F401
from gym_minigrid.wrappers import FullyObsWrapper  # noqa: E402

# (key in the output file, env id, wrap with FullyObsWrapper)
CASES = [
    ("MiniGrid-Empty-8x8-v0", "MiniGrid-Empty-8x8-v0", False),                      # configs[0] / configs[1]
    ("MiniGrid-DoorKey-8x8-v0", "MiniGrid-DoorKey-8x8-v0", False),                  # configs[2]
    ("MiniGrid-LavaCrossingS9N1-v0", "MiniGrid-LavaCrossingS9N1-v0", False),        # configs[3]
    ("MiniGrid-Empty-16x16-v0+FullyObsWrapper", "MiniGrid-Empty-16x16-v0", True),   # configs[4]
]


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def time_env(env_id, full, T):
    env = gym.make(env_id)
    if full:
        env = FullyObsWrapper(env)
    env.seed(0)
    env.reset()
    actions = np.random.RandomState(0).randint(0, 7, size=T)
    episodes = 0
    t0 = time.perf_counter()
    for a in actions:
        obs, reward, done, info = env.step(a)
        if done:
            env.reset()
            episodes += 1
    dt = time.perf_counter() - t0
    return dt, episodes, tuple(obs["image"].shape)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--repeats", type=int, default=3, help="the loop is repeated and the FASTEST repeat is reported (all are kept in the file)")
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "reference_cpu.json"))
    args = ap.parse_args()
    out = {"what": "reference gym_minigrid (pure Python + NumPy) timed on its own CPU path: gym.make(id) [+ FullyObsWrapper], "
                   "actions RandomState(0).randint(0,7,size=T), env.step(a), env.reset() on done; time.perf_counter around the loop",
           "script": "oracle/time_reference.py", "reference": "/root/reference (rohitrango/gym-minigrid), imported under oracle/refshim (stand-in gym)",
           "cpu_model": cpu_model(), "nproc": os.cpu_count(), "cores_used": 1,
           "python": platform.python_version(), "numpy": np.__version__, "envs": {}}
    for key, env_id, full in CASES:
        runs = [time_env(env_id, full, args.steps) for _ in range(args.repeats)]
        best = min(r[0] for r in runs)
        out["envs"][key] = {"steps": args.steps, "seconds": best, "steps_per_s": args.steps / best,
                            "all_repeats_steps_per_s": [args.steps / r[0] for r in runs], "episodes": runs[0][1],
                            "obs_shape": list(runs[0][2])}
        print("%-45s %8.0f steps/s (%d steps, best of %d; %d episodes)" % (key, args.steps / best, args.steps, args.repeats, runs[0][1]), flush=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
