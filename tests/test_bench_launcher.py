"""`python bench.py --gpus N` started plainly (WORLD_SIZE unset) is a launcher that makes no GPU call: it starts the N
rank processes, relays rank 0's ONE JSON line and passes failures on.  Driven end to end here on CPU with
--dry-run-ranks: launcher -> torch.distributed.run -> 2 gloo ranks -> sharding by global env index -> the side-stream
logging gather of done/reward (GatherLogger) -> per-rank figures -> one line.  The stand-in env of the dry run derives
done/reward from the synthetic action stream, so every rank checks every gathered element against a formula of the
global env index (bench.py exits non-zero if they differ)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(*args, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, BENCH, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


def keys_of(d, prefix=""):
    out = set()
    for k, v in d.items():
        out.add(prefix + k)
        if isinstance(v, dict):
            out |= keys_of(v, prefix + k + ".")
    return out


@pytest.mark.timeout(600)
def test_launcher_two_gloo_ranks_one_json_line():
    rc, out, err = run_bench("--gpus", "2", "--dry-run-ranks", "--config", "lava4m", "--envs-per-gpu", "1001",
                             "--steps", "40", "--warmup", "3", "--log-every", "16")
    assert rc == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out                      # exactly ONE line on stdout: rank 0's
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 40 and j["warmup"] == 3 and j["scaling"] == "weak"
    assert j["gather_checked"] is True and j["dry_run"] is True
    assert j["config"]["env_id"] == "MiniGrid-LavaCrossingS9N1-v0" and j["config"]["envs_per_gpu"] == 1001
    assert "all-gather" in j["config"]["logging_exchange"] and "gloo" in j["config"]["logging_exchange"]
    # the line proves who took part and what the exchange cost: backend, world, one identity per rank, exchanges inside the timed region
    r = j["rccl"]
    assert r["backend"] == "gloo" and r["world"] == 2 and [d["rank"] for d in r["devices"]] == [0, 1] and r["distinct_devices"] is True
    assert r["exchanges_in_timed_region"] == 2 and r["exchanges"] == 3 and r["warmup_exchanges"] == 1 and r["log_every"] == 16 and r["collective_us_mean"] > 0
    assert r["bytes_per_rank_per_exchange"] == 5 * 1001
    assert [r["rank"] for r in j["roofline"]["per_rank"]] == [0, 1]
    assert j["episodes_in_timed_region"] == sum(r["episodes"] for r in j["roofline"]["per_rank"]) > 0
    # the N = 1 line has the same shape (plus what only one rank can have: no exchange, no gather check)
    rc1, out1, err1 = run_bench("--gpus", "1", "--dry-run-ranks", "--config", "lava4m", "--envs-per-gpu", "1001",
                                "--steps", "40", "--warmup", "3", "--no-cpu-baseline")
    assert rc1 == 0, err1[-2000:]
    j1 = json.loads(out1.strip())
    assert {k for k in keys_of(j) - keys_of(j1) if not k.startswith("rccl")} == {"gather_checked"} and keys_of(j1) <= keys_of(j)
    assert j1["config"]["logging_exchange"] == "none"
    # global indexing: 2 ranks x 1001 envs saw the same (env, t) pairs as 1 rank x 2002 envs
    rc2, out2, err2 = run_bench("--gpus", "1", "--dry-run-ranks", "--config", "lava4m", "--envs-per-gpu", "2002",
                                "--steps", "40", "--warmup", "3", "--no-cpu-baseline")
    assert rc2 == 0, err2[-2000:]
    j2 = json.loads(out2.strip())
    assert j2["episodes"] == j["episodes"] and abs(j2["reward_sum"] - j["reward_sum"]) < 1e-6


@pytest.mark.timeout(600)
def test_drivers_own_command_shape_times_the_exchange():
    """`bench.py --gpus N --steps 20 --warmup W` (what the driver runs): --log-every defaults to min(256, steps // 2 + 1), so the one
    exchange of a 20-step run lands INSIDE the timed region, with nine steps enqueued behind it, and is reported with its own duration."""
    rc, out, err = run_bench("--gpus", "2", "--dry-run-ranks", "--steps", "20", "--warmup", "5", "--envs-per-gpu", "777")
    assert rc == 0, err[-2000:]
    j = json.loads(out.strip())
    r = j["rccl"]
    assert r["log_every"] == 11 and r["exchanges_in_timed_region"] == 1 and r["exchanges"] == 2 and r["warmup_exchanges"] == 1
    assert r["collective_us_mean"] > 0 and r["steps_enqueued_behind_the_last_exchange"] == 9
    assert j["gather_checked"] is True and "1 inside the timed region" in j["config"]["logging_exchange"]
    assert len(r["devices"]) == 2 and r["devices"][0]["host"]


@pytest.mark.timeout(600)
def test_launcher_passes_rank_failures_on():
    rc, out, err = run_bench("--gpus", "2", "--dry-run-ranks", "--env", "MiniGrid-NoSuchEnv-v0", "--steps", "4", "--warmup", "1")
    assert rc != 0 and not out.strip()


def test_plain_multi_gpu_start_makes_no_gpu_call_in_the_parent():
    """The launcher branch must run before torch / the package are imported (a parent that has initialised HIP must never
    spawn-and-wait with the GPU held, let alone re-exec)."""
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args, argv)") < main.index("import torch")
    launcher = src[src.index("def launch_ranks"):src.index("class _DryEnv")]
    assert "import torch" not in launcher and "os.exec" not in src
