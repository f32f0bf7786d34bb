#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched MiniGrid hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config empty8|doorkey8|lava4m|empty16full]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Started plainly with --gpus N > 1 (WORLD_SIZE unset) this process is only a LAUNCHER: it makes no torch / HIP call,
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (one rank per GPU over RCCL),
relays rank 0's single JSON line and exits with the child's code.  Nothing that has touched the GPU is ever re-exec'ed.

A "step" is one lockstep `env.step(actions)` (transition + gen_obs + encode, with in-kernel auto-reset) over every env
of the rank.  Workloads are BASELINE.json's configs:
  empty8       configs[1]  MiniGrid-Empty-8x8-v0, 1,048,576 envs per GPU, uint8 (N,7,7,3)       (default, the bench line)
  doorkey8     configs[2]  MiniGrid-DoorKey-8x8-v0, 1,048,576 envs per GPU
  lava4m       configs[3]  MiniGrid-LavaCrossingS9N1-v0, 524,288 envs per GPU (4,194,304 over 8 GPUs), RCCL gather of
                           the per-env done/reward vectors every --log-every steps on a side stream
  empty16full  configs[4]  MiniGrid-Empty-16x16-v0 + FullyObsWrapper encode, 262,144 envs per GPU
Weak scaling: per-GPU work is fixed as N grows; envs shard by global index (seeds and synthetic actions are keyed by the
GLOBAL env index), no data-path collective.  For N > 1 the logging exchange (north_star: "RCCL ... to gather the
done/reward scalars") runs every --log-every steps (default min(256, --steps // 2 + 1): a short run exchanges once, half-way through, so
the timed region holds the exchange AND steps enqueued behind it; one more, untimed, runs with the warm-up steps because a
communicator's first collective sets up its channels): an all-gather of done u8 + reward f32 of every rank, issued on a side stream behind a snapshot of
the step's outputs into one of two snapshot buffers; the step stream waits for it only if a gather is still running when
its buffer comes round again, and the line says whether that happened (`rccl.step_stream_wait_us_total`).
Inputs (synthetic counter-based actions for all K+W steps, env state) are resident in HBM before the timed region.
The timed region is bracketed by barrier + torch.cuda.synchronize() on both sides; the reported time is the MAX over
ranks; rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     the step kernel (`kernel` = the instantiation the library's selector launches, mgx_step_kernel_name) against the
               HBM roof: algorithmic bytes per launch / the kernel's average duration, measured with HIP events on the launch
               stream inside the timed region: `span_us_per_step` = the whole stream span / steps (kernel + launch gaps and,
               where a workload has them, k_dynobs / epilogues / k_levelgen) -- an upper bound of the kernel's mean duration,
               and what `avg_kernel_us`, `achieved` and `frac` are derived from; `event_pair_us` = event pairs around every
               16th launch of THAT kernel alone (mgx_profile_kernel; includes the ~2 us the two markers take), reported when
               the run is long enough for >= 8 pairs (K >= 128) and used instead of the span only for workloads whose step is
               several kernels.  rocprofv3's per-dispatch average under profiles/ reads just below both.
               `achieved` uses the bytes the kernel ALGORITHMICALLY touches: the gather form of large grids (S > 256) reads its
               7x7 window, not the grid (`algorithmic_bytes_per_env_step` then says so).
               The bytes are those of THIS layout (233 B/env-step for 8x8 + 7x7 view) -- smaller
               than SURVEY.md section 8d's 372 B, which assumed 3-byte cells; the survey-basis rate is given beside it.
               `traffic` = HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2
               + WRITE_SIZE, KiB), keyed by workload in profiles/traffic.json.  `per_rank` lists every rank's figure;
               the top-level achieved/frac is the SLOWEST rank's.
  cpu_baseline the CPU oracle (C restatement of the reference, oracle/minigrid_oracle.c; kind "port") timed on
               one host core of this box on a bounded sample of the same workload (rank 0, N = 1 only).
  rccl         (N > 1) backend, world size, the device identity (UUID / PCI address) every rank reported -- ranks sharing a GPU
               fail the run unless --rehearse-on-one-gpu -- and the exchange's own timing: exchanges inside the timed region,
               mean / max duration on the side stream, time the step stream waited for it.
  episodes_in_timed_region   episodes that ended (and were auto-reset) inside the K timed steps, over all ranks: a short
               window on a time-out-only workload (Empty-8x8: every 256 steps) honestly shows 0 here.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gym-minigrid_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured streaming ceiling

CONFIGS = {
    "empty8": dict(env="MiniGrid-Empty-8x8-v0", envs_per_gpu=1048576, obs_mode="partial", baseline="configs[1]"),
    "doorkey8": dict(env="MiniGrid-DoorKey-8x8-v0", envs_per_gpu=1048576, obs_mode="partial", baseline="configs[2]"),
    "lava4m": dict(env="MiniGrid-LavaCrossingS9N1-v0", envs_per_gpu=524288, obs_mode="partial", baseline="configs[3]"),
    "empty16full": dict(env="MiniGrid-Empty-16x16-v0", envs_per_gpu=262144, obs_mode="full", baseline="configs[4]"),
}

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


def step_kernel_bytes_per_step(W, H, obs_mode, view=7, kernel=""):
    """What the STEP KERNEL alone streams (the one-hot / flat epilogues are kernels of their own): its output is the triples.
    The gather form (k_step<0,0,3,V>: partial view on grids past 16x16) never reads the grid: per env-step it touches the
    window excerpt (V columns of 4 / 8 / 12 bytes for V = 3 / 5, 7 / 9, 11) and the forward cell, so that is what its roofline is priced on."""
    cells = (W * H + 3) // 4 * 4
    if re.match(r"k_step<\d+,\d+,3,", kernel):
        cells = view * (4 if view <= 3 else 8 if view <= 7 else 12) + 1   # V columns of 4 / 8 / 12 bytes + the forward cell
    if kernel.startswith("k_step_dyn<"):
        # Dynamic-Obstacles, walk fused into the step: the tile read AND written back (the obstacles moved), obstacle order 8 + 8, RNG position
        # 4 + 4, the 24-byte window of the draw tape -- plus what every step moves
        return 2 * cells + 8 + 8 + 8 + 8 + 4 + 4 + 24 + 1 + obs_cells(W, H, obs_mode, view) * 3 + 4 + 1
    if kernel.startswith("k_step_onehot<"):
        return cells + 8 + 8 + 1 + obs_cells(W, H, obs_mode, view) * 21 + 4 + 1   # the one-hot image leaves from the step kernel itself
    return cells + 8 + 8 + 1 + obs_cells(W, H, obs_mode, view) * 3 + 4 + 1


def reference_cpu_note(env_id, obs_mode):
    """The reference's own Python path timed by oracle/time_reference.py in the build container (profiles/reference_cpu.json):
    quoted beside the port's number, never measured here (the reference does not travel to the GPU box)."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "reference_cpu.json")))
        key = env_id + ("+FullyObsWrapper" if obs_mode.startswith("full") else "")
        r = j["envs"][key]
        return "reference Python %.0f steps/s on 1 core of %s (%d logical CPUs; %d steps, profiles/reference_cpu.json, oracle/time_reference.py)" % (
            r["steps_per_s"], j["cpu_model"], j["nproc"], r["steps"])
    except Exception:
        return "reference Python: profiles/reference_cpu.json has no entry for this workload"


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
           "sample": "%s, %d envs x %d steps (%.1f s), oracle/minigrid_oracle.c single thread, same action stream; %s"
                     % (env_id, n, T, dt1, reference_cpu_note(env_id, obs_mode))}
    if cores > 1:
        stepsC, dtC = run(T * min(cores, 4), cores)
        out["all_cores"] = {"value": stepsC / dtC, "cores": cores, "seconds": dtC}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--config", default="empty8", choices=sorted(CONFIGS), help="a BASELINE.json workload (see the module docstring)")
    ap.add_argument("--env", default=None, help="override the config's env id")
    ap.add_argument("--envs-per-gpu", type=int, default=None, help="override the config's batch per GPU")
    ap.add_argument("--obs-mode", default=None, choices=sorted(OBS_CHANNELS))
    ap.add_argument("--view", type=int, default=7, help="agent_view_size (ViewSizeWrapper)")
    ap.add_argument("--log", default="gather", choices=("gather", "allreduce", "none"),
                    help="N > 1 logging exchange every --log-every steps: all-gather of per-env done/reward on a side stream "
                         "(default), all-reduce of (episodes, reward_sum), or none")
    ap.add_argument("--log-every", type=int, default=None,
                    help="steps between two logging exchanges (default min(256, --steps // 2 + 1): a short run exchanges once, half-way through its timed region)")
    ap.add_argument("--new-level-each-episode", action="store_true",
                    help="plain reference episode boundary: every reset draws a new level on the GPU (k_levelgen)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="REHEARSAL ONLY: every rank uses cuda:0 (backend gloo: RCCL refuses two ranks on one GPU) to exercise the multi-rank code "
                         "path on a one-GPU box; the numbers it prints are meaningless")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="PLUMBING TEST ONLY (no GPU, gloo): launcher -> ranks -> rendezvous -> sharding -> logging gather -> "
                         "one JSON line, with a stand-in for the env that derives done/reward from the action stream; "
                         "the numbers it prints are meaningless")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="launcher: give up on the rank processes after this many seconds")
    args = ap.parse_args(argv)
    c = CONFIGS[args.config]
    args.env = args.env or c["env"]
    args.envs_per_gpu = args.envs_per_gpu or c["envs_per_gpu"]
    args.obs_mode = args.obs_mode or c["obs_mode"]
    if args.log_every is None:
        # a short run (the driver's --steps 20) exchanges once, half-way through, so that the timed region holds both the exchange and
        # steps that overlap it; from 512 steps on the interval is the 256 steps a training loop would log at
        args.log_every = max(1, min(256, args.steps // 2 + 1))
    return args


# ---------------------------------------------------------------------------------------------- launcher (no GPU call)
def launch_ranks(args, argv):
    """python bench.py --gpus N with WORLD_SIZE unset: start the N rank processes, relay rank 0's JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC between the ranks (RCCL), before any rank starts HIP
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True, text=True)
    try:
        out, _ = proc.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, 15)  # the process group this launcher created (start_new_session), nothing else
            time.sleep(5)
            os.killpg(proc.pid, 9)
        except ProcessLookupError:
            pass
        proc.wait()
        sys.stderr.write("bench.py launcher: the rank processes did not finish within %.0f s\n" % args.launch_timeout)
        return 124
    line = None
    for ln in out.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            sys.stderr.write(ln + "\n")      # anything else the ranks printed is not the result line
    if proc.returncode != 0:
        sys.stderr.write("bench.py launcher: torch.distributed.run exited with %d\n" % proc.returncode)
        return proc.returncode
    if line is None:
        sys.stderr.write("bench.py launcher: rank 0 printed no result line\n")
        return 1
    print(line, flush=True)
    return 0


# ---------------------------------------------------------------------------------------------- dry-run stand-in
class _DryEnv:
    """--dry-run-ranks: no simulator, no GPU.  done/reward are pure functions of the synthetic action stream and the
    GLOBAL env index, so the gathered vectors can be checked against a formula on every rank."""

    def __init__(self, n_local, offset):
        import torch
        self.n, self.offset, self.torch = n_local, offset, torch
        self.done = torch.zeros(n_local, dtype=torch.uint8)
        self.reward = torch.zeros(n_local, dtype=torch.float32)
        self.episodes = 0.0
        self.reward_sum = 0.0

    @staticmethod
    def outputs(a):
        return (a == 6).astype("uint8"), (a.astype("float32") * 0.125)

    def actions(self, offset, n, t):
        import numpy as np
        import gym_minigrid_amd as mg
        return mg.action_stream(0, np.arange(offset, offset + n), t)

    def step(self, t):
        d, r = self.outputs(self.actions(self.offset, self.n, t))
        self.done.copy_(self.torch.from_numpy(d))
        self.reward.copy_(self.torch.from_numpy(r))
        self.episodes += float(d.sum())
        self.reward_sum += float(r.sum())


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between the ranks (before HIP starts)
    import torch
    import gym_minigrid_amd as mg
    from gym_minigrid_amd import dist as mdist
    import torch.distributed as dist

    dry = args.dry_run_ranks
    rank, local_rank, world = mdist.env_rank_info()
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)          # before the process group: RCCL's communicator is bound to THIS device
        dev = torch.device("cuda", local_rank)
    backend = "gloo" if (dry or (args.rehearse_on_one_gpu and not args.backend)) else args.backend
    mdist.init_process_group(backend, device=None if dry else dev)
    # who is here: every rank's device identity, all-gathered (N ranks must be N distinct GPUs)
    rccl = None
    if world > 1:
        ident = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid()}
        ident.update({"key": "cpu|pid-%d" % os.getpid()} if dry else mdist.device_identity(local_rank))
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        keys = [i["host"] + "|" + i["key"] for i in idents]
        distinct = len(set(keys)) == world
        if not all(i.get("verifiable", True) for i in idents):
            distinct = None   # a runtime that reports neither UUID nor PCI address: device indices under per-rank visibility masks prove nothing either way
        rccl = {"backend": dist.get_backend(), "world": world, "distinct_devices": distinct,
                "devices": [{k: v for k, v in i.items() if k in ("rank", "local_rank", "index", "name", "uuid", "pci", "host")} for i in idents]}
        if distinct is False and not (args.rehearse_on_one_gpu or dry):
            raise SystemExit("bench.py: %d ranks but only %d distinct GPUs (%s); one rank per GPU, or --rehearse-on-one-gpu" % (world, len(set(keys)), sorted(set(keys))))

    n_local = args.envs_per_gpu
    n_total = n_local * world
    offset = rank * n_local
    K, Wm = args.steps, args.warmup
    stats2 = torch.zeros(2, dtype=torch.float64, device=dev)
    stats0 = torch.zeros(2, dtype=torch.float64, device=dev)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
            sync()

    logger = mdist.GatherLogger(n_local, dev, world) if (world > 1 and args.log == "gather") else None
    if dry:
        env = _DryEnv(n_local, offset)
        step = lambda t: env.step(t)
        outputs = lambda: (env.done, env.reward)
        read_stats = lambda out: out.copy_(torch.tensor([env.episodes, env.reward_sum], dtype=torch.float64))
    else:
        env = mg.VecMiniGrid(args.env, num_envs=n_local, device=local_rank, seeds=0, obs_mode=args.obs_mode,
                             auto_reset=True, backend="torch", env_offset=offset,
                             new_level_each_episode=args.new_level_each_episode, agent_view_size=args.view)
        env.reset()
        # synthetic inputs for every step, resident in HBM before timing starts
        acts = env.fill_actions(0, 0, K + Wm)
        step = lambda t: env.step(acts[t])
        outputs = lambda: (env._done, env._reward)
        read_stats = env.read_stats_async

    for t in range(Wm):
        step(t)
    warmup_exchanges = 0
    if world > 1 and (Wm > 0 or dry):
        # one untimed exchange with the warm-up steps: the first collective of a communicator sets up its channels (RCCL: tens of
        # milliseconds), which is start-up cost, not a per-step cost
        if logger is not None:
            logger.submit(*outputs())
            logger.wait()
            warmup_exchanges = 1
        elif args.log == "allreduce":
            read_stats(stats2)
            mdist.allreduce_log(stats2)
    read_stats(stats0)
    barrier()
    if not dry:
        # an event pair around every 16th launch of the step kernel (each pair costs the stream ~2-5 us, so short runs take none
        # that count: below 128 steps only the first launch is bracketed and the figure is left off the line)
        env.profile_begin(stride=16 if K >= 128 else K + 1)
    exchanges_timed = 0
    t0 = time.perf_counter()
    for t in range(Wm, Wm + K):
        step(t)
        if world > 1 and args.log_every > 0 and (t - Wm + 1) % args.log_every == 0:
            exchanges_timed += 1
            if logger is not None:
                logger.submit(*outputs())        # RCCL all-gather of done/reward over xGMI, on the side stream
            elif args.log == "allreduce":
                read_stats(stats2)
                mdist.allreduce_log(stats2)      # 16 bytes
    if not dry:
        env.profile_stop()             # the span's end marker, enqueued behind the K-th step (no wait here)
    if logger is not None:
        logger.wait()
    barrier()
    dt = time.perf_counter() - t0
    launches, span_ms = (K, 0.0) if dry else env.profile_end()  # HIP events on the launch stream (already complete)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- after the timed region: totals, per-rank kernel times, a check of the gathered vectors
    read_stats(stats2)
    local_totals = (stats2 - stats0).clone()
    mdist.allreduce_log(stats2)
    mdist.allreduce_log(stats0)
    sync()
    episodes, reward_sum = [float(x) for x in stats2.tolist()]
    ep0 = float(stats0.tolist()[0])
    gather_checked = None
    if logger is not None:
        if logger.submitted == 0:            # (--log-every larger than --steps was asked for: exercise the exchange once, outside the timed region)
            logger.submit(*outputs())
        gd, gr = logger.wait()
        sync()
        # the gathered vectors hold every rank's shard in global env order: own shard in place, totals equal the sum of the shards'
        own = slice(rank * n_local, (rank + 1) * n_local)
        ok = bool(torch.equal(gd[own], logger.done)) and bool(torch.equal(gr[own], logger.reward))
        sums = torch.tensor([float(logger.done.sum()), float(logger.reward.double().sum())], dtype=torch.float64, device=dev)
        mdist.allreduce_log(sums)
        ok = ok and float(gd.sum()) == float(sums[0]) and abs(float(gr.double().sum()) - float(sums[1])) <= 1e-6 * max(1.0, abs(float(sums[1])))
        if dry:                                # the stand-in's outputs are a formula of the global index: check every element
            d_all, r_all = env.outputs(env.actions(0, n_total, Wm + (K // args.log_every) * args.log_every - 1 if K >= args.log_every else Wm + K - 1))
            ok = ok and bool((gd.numpy() == d_all).all()) and bool((gr.numpy() == r_all).all())
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_checked = bool(flag.item() == 1.0)
        if not gather_checked:
            raise SystemExit("bench.py: the gathered done/reward vectors do not match the shards")
    k_n, k_ms = (0, 0.0) if dry else env.profile_kernel()
    span_s = span_ms * 1e-3 / max(launches, 1)
    sampled_s = (k_ms * 1e-3 / k_n) if k_n >= 8 else 0.0   # fewer than 8 pairs say nothing about the mean (one pair = the first launch after a barrier)
    kernel_name = "dry-run" if dry else env.step_kernel_name()
    # The step kernel's mean duration is bounded from above by the stream span / launches (kernel + inter-kernel gap + whatever
    # else the workload enqueues per step): the roofline figure is derived from THAT, so the rate is a lower bound.  Only a
    # workload whose step is several kernels (k_dynobs in front, an epilogue or k_levelgen behind) uses the per-launch event
    # pairs instead -- kernel + the ~2 us the two markers take -- and only with >= 8 of them.  rocprofv3's per-dispatch
    # average (profiles/) reads slightly below both.
    multi_kernel = (not dry) and (args.new_level_each_episode or args.obs_mode.endswith("onehot") or args.obs_mode.endswith("nocolor")
                                  or args.obs_mode.endswith("flat") or "Dynamic-Obstacles" in args.env)
    avg_kernel_s = sampled_s if (multi_kernel and sampled_s > 0.0) else span_s
    mine = torch.tensor([avg_kernel_s, span_s, float(local_totals[0]), sampled_s], dtype=torch.float64, device=dev)
    per = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(per, mine)
    else:
        per = [mine]
    sync()

    if rank == 0:
        cfg = mg.env_config(args.env)
        bps = step_kernel_bytes_per_step(cfg.width, cfg.height, args.obs_mode, args.view, kernel_name)
        lbps = layout_bytes_per_step(cfg.width, cfg.height, args.obs_mode, args.view)
        sbps = survey_bytes_per_step(cfg.width, cfg.height, args.obs_mode, args.view)
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                key = "%s/%s/%d" % (args.env, args.obs_mode, n_local)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        per_rank = []
        for r, v in enumerate(per):
            ks, ss, ep, es = [float(x) for x in v.tolist()]
            ach = bps * n_local / ks / 1e9 if ks > 0 else 0.0
            per_rank.append({"rank": r, "avg_kernel_us": ks * 1e6, "span_us_per_step": ss * 1e6, "event_pair_us": (es * 1e6) if es > 0 else None, "achieved": ach,
                             "frac": ach / HBM_PEAK_GBS, "episodes": ep})
        slow = min(per_rank, key=lambda x: x["achieved"])
        obs_desc = ("float32 (N,%d) FlatObs" % (obs_cells(cfg.width, cfg.height, args.obs_mode, args.view) * 3 + FLAT_MISSION) if args.obs_mode.endswith("flat")
                    else "uint8 (N,%d,%d,%d)" % (args.view, args.view, OBS_CHANNELS[args.obs_mode]) if args.obs_mode.startswith("partial")
                    else "uint8 (N,W,H,%d) FullyObs" % OBS_CHANNELS[args.obs_mode])
        exchange = "none" if world == 1 else ("%s all-gather of done u8 + reward f32 (%d B per rank) every %d steps on a side stream, %d inside the timed region" % ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend(), 5 * n_local, args.log_every, exchanges_timed)
                                              if logger is not None else ("all-reduce of (episodes, reward_sum) every %d steps" % args.log_every if args.log == "allreduce" else "none"))
        out = {
            "metric": "env-steps/sec", "value": n_total * K / dt, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s, %d batched envs per GPU (%d total), obs %s, uniform random actions 0..6 "
                                   "(counter-based), auto-reset on done" % (args.env, n_local, n_total, obs_desc),
                       "name": args.config, "baseline_config": CONFIGS[args.config]["baseline"] if (args.env == CONFIGS[args.config]["env"] and n_local == CONFIGS[args.config]["envs_per_gpu"] and args.obs_mode == CONFIGS[args.config]["obs_mode"]) else None,
                       "env_id": args.env, "envs_per_gpu": n_local, "obs_mode": args.obs_mode, "parallelism": "env-shard x%d" % world,
                       "logging_exchange": exchange,
                       "new_level_each_episode": bool(args.new_level_each_episode)},
            "episodes": episodes, "reward_sum": reward_sum, "episodes_in_timed_region": episodes - ep0,
            "roofline": {"bound": "hbm", "achieved": slow["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": slow["frac"], "traffic": traffic,
                         "kernel": kernel_name, "avg_kernel_from": "event pairs" if (multi_kernel and sampled_s > 0.0) else "span / launches",
                         "avg_kernel_us": slow["avg_kernel_us"], "kernel_samples": k_n, "launches": launches,
                         "span_us_per_step": slow["span_us_per_step"], "event_pair_us": slow["event_pair_us"],
                         "algorithmic_bytes_per_env_step": bps, "layout_bytes_per_env_step_all_kernels": lbps,
                         "measured_streaming_ceiling": 6290.0,
                         "achieved_on_layout_bytes": lbps * n_local / (slow["avg_kernel_us"] * 1e-6) / 1e9 if (slow["avg_kernel_us"] > 0 and lbps != bps and not multi_kernel) else None,
                         "survey_bytes_per_env_step": sbps, "achieved_on_survey_bytes": sbps * n_local / (slow["avg_kernel_us"] * 1e-6) / 1e9 if slow["avg_kernel_us"] > 0 else 0.0,
                         "per_rank": per_rank},
        }
        if rccl is not None:
            rccl["exchanges_in_timed_region"] = exchanges_timed
            rccl["warmup_exchanges"] = warmup_exchanges
            rccl["steps_enqueued_behind_the_last_exchange"] = (K - (K // args.log_every) * args.log_every) if args.log_every <= K else 0
            rccl["log_every"] = args.log_every
            rccl["bytes_per_rank_per_exchange"] = 5 * n_local if logger is not None else (16 if args.log == "allreduce" else 0)
            if logger is not None:
                rccl.update(logger.stats())   # rank 0's side stream: collective_us_mean / max, step-stream waits
            out["rccl"] = rccl
        if gather_checked is not None:
            out["gather_checked"] = gather_checked
        if dry:
            out["dry_run"] = True
        if args.rehearse_on_one_gpu:
            out["rehearsal_on_one_gpu"] = True
        if world == 1 and not args.no_cpu_baseline and not dry:
            out["cpu_baseline"] = cpu_baseline(args.env, args.obs_mode)
        print(json.dumps(out), flush=True)
    if not dry:
        env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
