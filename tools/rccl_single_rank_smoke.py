#!/usr/bin/env python3
"""RCCL on ONE rank (GPU box): everything bench.py's N > 1 path asks of torch.distributed's nccl (= RCCL) backend, with world_size 1 --
the only RCCL run a one-GPU lease allows (RCCL refuses two ranks on one GPU; the multi-rank order of the same calls is covered on CPU
over gloo by tests/test_bench_launcher.py and tests/test_dist_gloo.py).  Exercises: init_process_group(backend='nccl', device_id=...)
after set_device, all_gather_object (device identities), all_gather_into_tensor of u8 / f32 on a side stream behind an event
(dist.GatherLogger with the collective forced on), all_reduce SUM / MAX / MIN on f64, barrier, destroy.
    python tools/rccl_single_rank_smoke.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gym-minigrid_amd"))
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist
from gym_minigrid_amd import dist as mdist

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
ident = mdist.device_identity(0)
got = [None]
dist.all_gather_object(got, ident)
assert got[0]["key"] == ident["key"]
n = 524288
lg = mdist.GatherLogger(n, dev, world=1)
lg.world = 1
done = (torch.arange(n, device=dev) % 7 == 0).to(torch.uint8)
reward = torch.arange(n, device=dev, dtype=torch.float32) * 0.5
# the collective itself, as GatherLogger.submit issues it for world > 1: on the side stream, behind an event of the current stream
side = torch.cuda.Stream(device=dev)
ev = torch.cuda.Event()
gd = torch.zeros(n, dtype=torch.uint8, device=dev)
gr = torch.zeros(n, dtype=torch.float32, device=dev)
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in range(4):
    ev.record(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        side.wait_event(ev)
        t0.record(side)
        dist.all_gather_into_tensor(gd, done)
        dist.all_gather_into_tensor(gr, reward)
        t1.record(side)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
assert torch.equal(gd, done) and torch.equal(gr, reward)
x = torch.tensor([3.0, 4.5], dtype=torch.float64, device=dev)
for op in (dist.ReduceOp.SUM, dist.ReduceOp.MAX, dist.ReduceOp.MIN):
    y = x.clone()
    dist.all_reduce(y, op=op)
    assert torch.equal(y, x)
parts = [torch.zeros_like(x)]
dist.all_gather(parts, x)
assert torch.equal(parts[0], x)
dist.barrier()
torch.cuda.synchronize()
print("RCCL single-rank smoke OK: backend %s, device %s, all_gather_into_tensor of %d B + %d B on a side stream: %.1f us (last of 4)" % (
    dist.get_backend(), ident.get("pci") or ident.get("uuid"), n, 4 * n, t0.elapsed_time(t1) * 1e3))
dist.destroy_process_group()
