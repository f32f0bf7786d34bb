#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched MiniGrid hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one lockstep `env.step(actions)` (transition + gen_obs + encode, with in-kernel auto-reset) over
every env of the rank: by default BASELINE.json configs[1], MiniGrid-Empty-8x8-v0 with 1,048,576 envs PER GPU
(weak scaling: per-GPU work is fixed as N grows; envs shard by global index, no data-path collective).
Inputs (synthetic counter-based actions for all K+W steps, env state) are resident in HBM before the timed
region.  The timed region is bracketed by barrier + torch.cuda.synchronize() on both sides; the reported time
is the MAX over ranks; rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     dominant kernel (k_step) against the HBM roof: algorithmic bytes per launch / the kernel's average
               duration measured with HIP events on the launch stream over the timed region.  The bytes are
               those of THIS layout (233 B/env-step for 8x8 + 7x7 view) -- smaller than
               SURVEY.md section 8d's 372 B, which assumed 3-byte cells; the survey-basis rate is given beside it
               (it exceeds the HBM peak precisely because the layout moves fewer bytes).  `traffic` = HBM bytes per
               launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE, KiB).
  cpu_baseline the CPU oracle (C restatement of the reference, oracle/minigrid_oracle.c; kind "port") timed on
               one host core of this box on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gym-minigrid_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured streaming ceiling


OBS_CHANNELS = {"partial": 3, "full": 3, "partial_onehot": 21, "full_onehot": 22, "full_onehot_nocolor": 15, "flat": 3, "full_flat": 3}
FLAT_MISSION = 27 * 96  # FlatObsWrapper: float32 image ++ one-hot mission string


def obs_cells(W, H, obs_mode, view=7):
    return view * view if obs_mode.startswith("partial") or obs_mode == "flat" else W * H


def survey_bytes_per_step(W, H, obs_mode, view=7):
    """SURVEY.md section 8d figure (3-byte cells, 12-byte agent records): action 1 + agent 12 rd + 12 wr + grid W*H*3 rd
    + <=1 cell (3) wr + obs + reward 4 + done 1  ->  372 B for an 8x8 grid with the 7x7 view."""
    grid = W * H * 3
    obs = obs_cells(W, H, obs_mode, view) * OBS_CHANNELS[obs_mode]
    if obs_mode.endswith("flat"):
        obs = (obs + FLAT_MISSION) * 4
    return 1 + 12 + 12 + grid + 3 + obs + 4 + 1


def layout_bytes_per_step(W, H, obs_mode, view=7):
    """Bytes THIS layout has to move per env-step: 1-byte cell codes (W*H rounded up to 4) read,
    8-byte agent record read + written, action 1, obs written, reward 4, done 1  ->  233 B for 8x8 + 7x7 view.
    The roofline is priced on this (smaller, conservative) figure: it is what the kernel really streams."""
    cells = (W * H + 3) // 4 * 4
    n = obs_cells(W, H, obs_mode, view)
    obs = n * OBS_CHANNELS[obs_mode]
    if obs_mode.endswith("onehot") or obs_mode.endswith("nocolor"):
        obs += 2 * 3 * n  # the one-hot epilogue is a second kernel: triples written by k_step and read back
    if obs_mode.endswith("flat"):
        obs = (obs + FLAT_MISSION) * 4 + 2 * 3 * n
    return cells + 8 + 8 + 1 + obs + 4 + 1


def cpu_baseline(env_id, obs_mode, target_seconds=10.0):
    """Oracle (scalar C port of the reference algorithm) on the host: one core, then one thread per core (the C
    rollout holds no Python lock), on a bounded sample of the same workload."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import gym_minigrid_amd as mg
    from oracle.minigrid_oracle import OracleEnvs

    cfg = mg.env_config(env_id)
    n = 4096
    grid, agent = mg.generate_levels(env_id, np.arange(n, dtype=np.uint64))
    full = obs_mode.startswith("full")
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))          # a one-GPU box's CPU share

    def run(T, workers):
        acts = mg.action_stream(0, np.arange(n)[None, :], np.arange(T)[:, None])
        parts = []
        for w in range(workers):
            sl = slice(w * n // workers, (w + 1) * n // workers)
            orc = OracleEnvs(cfg.width, cfg.height, cfg.max_steps, cfg.see_through_walls, cfg.lava_v1)
            orc.set_state(grid[sl], agent[sl])
            parts.append((orc, np.ascontiguousarray(acts[:, sl])))
        t0 = time.perf_counter()
        if workers == 1:
            steps = parts[0][0].rollout(parts[0][1], with_obs=not full, full=full)
        else:
            with ThreadPoolExecutor(workers) as ex:
                steps = sum(ex.map(lambda pa: pa[0].rollout(pa[1], with_obs=not full, full=full), parts))
        return steps, time.perf_counter() - t0

    steps, dt = run(64, 1)                    # calibrate
    T = int(max(64, min(100000, 64 * target_seconds / max(dt, 1e-6))))
    steps1, dt1 = run(T, 1)
    out = {"value": steps1 / dt1, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": "%s, %d envs x %d steps (%.1f s), oracle/minigrid_oracle.c single thread, same action stream; "
                     "reference Python measured at 6.7e3 steps/s/core in the build container (BASELINE.md)" % (env_id, n, T, dt1)}
    if cores > 1:
        stepsC, dtC = run(T * min(cores, 4), cores)
        out["all_cores"] = {"value": stepsC / dtC, "cores": cores, "seconds": dtC}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--env", default="MiniGrid-Empty-8x8-v0")
    ap.add_argument("--envs-per-gpu", type=int, default=1048576)
    ap.add_argument("--obs-mode", default="partial", choices=sorted(OBS_CHANNELS))
    ap.add_argument("--view", type=int, default=7, help="agent_view_size (ViewSizeWrapper)")
    ap.add_argument("--log-every", type=int, default=256, help="all-reduce (episodes, reward_sum) every L steps")
    ap.add_argument("--new-level-each-episode", action="store_true",
                    help="plain reference episode boundary: every reset draws a new level on the GPU (k_levelgen)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="REHEARSAL ONLY: every rank uses cuda:0 (with --backend gloo) to exercise the multi-rank code "
                         "path on a one-GPU box; the numbers it prints are meaningless")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between the ranks (before HIP starts)
    import torch
    import gym_minigrid_amd as mg
    from gym_minigrid_amd import dist as mdist
    import torch.distributed as dist

    rank, local_rank, world = mdist.init_process_group(args.backend)
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n_local = args.envs_per_gpu
    n_total = n_local * world
    offset = rank * n_local
    K, Wm = args.steps, args.warmup
    env = mg.VecMiniGrid(args.env, num_envs=n_local, device=local_rank, seeds=0, obs_mode=args.obs_mode,
                         auto_reset=True, backend="torch", env_offset=offset,
                         new_level_each_episode=args.new_level_each_episode, agent_view_size=args.view)
    env.reset()
    # synthetic inputs for every step, resident in HBM before timing starts
    acts = env.fill_actions(0, 0, K + Wm)
    stats2 = torch.zeros(2, dtype=torch.float64, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for t in range(Wm):
        env.step(acts[t])
    barrier()
    env.profile_begin()
    t0 = time.perf_counter()
    for t in range(Wm, Wm + K):
        env.step(acts[t])
        if world > 1 and args.log_every > 0 and (t - Wm + 1) % args.log_every == 0:
            env.read_stats_async(stats2)
            mdist.allreduce_log(stats2)      # RCCL over xGMI, 16 bytes, logging only
    launches, kernel_ms = env.profile_end()  # HIP events on the launch stream
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    env.read_stats_async(stats2)
    mdist.allreduce_log(stats2)
    env.sync()
    episodes, reward_sum = [float(x) for x in stats2.tolist()]

    if rank == 0:
        cfg = mg.env_config(args.env)
        bps = layout_bytes_per_step(cfg.width, cfg.height, args.obs_mode, args.view)
        sbps = survey_bytes_per_step(cfg.width, cfg.height, args.obs_mode, args.view)
        avg_kernel_s = kernel_ms * 1e-3 / max(launches, 1)
        achieved = bps * n_local / avg_kernel_s / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                key = "%s/%s/%d" % (args.env, args.obs_mode, n_local)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec", "value": n_total * K / dt, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s, %d batched envs per GPU (%d total), obs %s, uniform random actions 0..6 "
                                   "(counter-based), auto-reset on done" % (args.env, n_local, n_total,
                                                                            "float32 (N,%d) FlatObs" % (obs_cells(cfg.width, cfg.height, args.obs_mode, args.view) * 3 + FLAT_MISSION) if args.obs_mode.endswith("flat")
                                                                            else "uint8 (N,%d,%d,%d)" % (args.view, args.view, OBS_CHANNELS[args.obs_mode]) if args.obs_mode.startswith("partial")
                                                                            else "uint8 (N,W,H,%d) FullyObs" % OBS_CHANNELS[args.obs_mode]),
                       "env_id": args.env, "envs_per_gpu": n_local, "obs_mode": args.obs_mode, "parallelism": "env-shard x%d" % world,
                       "new_level_each_episode": bool(args.new_level_each_episode)},
            "episodes": episodes, "reward_sum": reward_sum,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_step" + ("+k_onehot" if "onehot" in args.obs_mode else "") + ("+k_flat" if args.obs_mode.endswith("flat") else "") + ("+k_levelgen" if args.new_level_each_episode else ""), "avg_kernel_us": avg_kernel_s * 1e6, "launches": launches,
                         "algorithmic_bytes_per_env_step": bps, "measured_streaming_ceiling": 6290.0,
                         "survey_bytes_per_env_step": sbps, "achieved_on_survey_bytes": sbps * n_local / avg_kernel_s / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.env, args.obs_mode)
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
