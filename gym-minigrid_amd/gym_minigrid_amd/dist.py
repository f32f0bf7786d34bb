"""Multi-GPU sharding (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Envs are independent: rank g of G owns the contiguous block
[g*N/G, (g+1)*N/G) of GLOBAL env indices, seeds and synthetic action streams are keyed by the global index,
so results do not depend on G.  The data path needs NO collective; the only exchange is for logging:
an all-reduce of (episodes finished, reward sum) or an all-gather of per-env done/reward."""
import os


def shard(n_total, rank, world):
    """Contiguous block of global env indices owned by `rank`: (offset, count)."""
    base, rem = divmod(int(n_total), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def env_rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None, device=None):
    """Rendezvous from RANK/WORLD_SIZE/MASTER_* (torchrun).  backend: 'nccl' (RCCL) on GPUs, 'gloo' on CPU.
    `device` (a cuda torch.device, already made current by the caller) binds the RCCL communicator to that GPU up front
    (init_process_group(device_id=...)): the communicator is then created eagerly on the right device instead of on whatever
    device is current at the first collective."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank_info()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this pool
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def device_identity(index):
    """What distinguishes the physical GPU behind cuda:<index> from every other GPU of the node: its UUID where the
    runtime reports one, else its PCI address.  bench.py all-gathers these to prove that N ranks drove N distinct GPUs."""
    import torch
    p = torch.cuda.get_device_properties(index)
    ident = {"index": int(index), "name": p.name}
    u = getattr(p, "uuid", None)
    if u is not None:
        ident["uuid"] = str(u)
    if hasattr(p, "pci_bus_id"):
        ident["pci"] = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    # (both: a runtime that reports one constant UUID for every GPU must not make distinct GPUs look like one)
    ident["key"] = "%s|%s" % (ident.get("uuid", "-"), ident.get("pci", "index-%d" % index))
    # Neither a UUID nor a PCI address: the device index is all there is, and under per-rank HIP_VISIBLE_DEVICES every rank sees index 0
    # (while a shared mask makes distinct indices look trustworthy without proof).  Such an identity is marked unverifiable and bench.py
    # then reports distinct_devices = null instead of aborting a legitimate N-GPU run.
    ident["verifiable"] = "uuid" in ident or "pci" in ident
    return ident


def allreduce_log(stats2):
    """Sum a 2-element (episodes, reward_sum) tensor over all ranks (RCCL all-reduce over xGMI; 16 bytes)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats2, op=dist.ReduceOp.SUM)
    return stats2


def gather_done_reward(done, reward, sizes=None):
    """All-gather the per-env done (uint8) and reward (f32) vectors of every shard, in global env order.
    Shards may differ in size by one env; they are padded to the largest for the collective.  `sizes` (the shard
    sizes of all ranks, e.g. [shard(N, r, G)[1] for r in range(G)]) saves the size exchange and its host sync."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return done, reward
    world = dist.get_world_size()
    if sizes is None:
        n = torch.tensor([done.shape[0]], device=done.device, dtype=torch.int64)
        got = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(got, n)
        sizes = [int(s.item()) for s in got]
    m = max(sizes)
    if done.shape[0] == m:
        pd, pr = done, reward
    else:
        pd = torch.zeros(m, dtype=done.dtype, device=done.device)
        pr = torch.zeros(m, dtype=reward.dtype, device=reward.device)
        pd[: done.shape[0]] = done
        pr[: reward.shape[0]] = reward
    gd = torch.empty(world * m, dtype=done.dtype, device=done.device)
    gr = torch.empty(world * m, dtype=reward.dtype, device=reward.device)
    dist.all_gather_into_tensor(gd, pd.contiguous())
    dist.all_gather_into_tensor(gr, pr.contiguous())
    if min(sizes) == m:
        return gd, gr
    return (torch.cat([gd[r * m: r * m + s] for r, s in enumerate(sizes)]),
            torch.cat([gr[r * m: r * m + s] for r, s in enumerate(sizes)]))


class GatherLogger:
    """The logging exchange of BASELINE config 4 ("RCCL gather of done/reward"), off the step stream.

    submit(done, reward) snapshots the step's per-env done (u8) / reward (f32) vectors into one of `buffers` (2) private
    snapshot buffers on the caller's stream (the env overwrites its own on the next step), then all-gathers the snapshots
    of every rank on a SIDE stream: RCCL's work is ordered behind the snapshot by an event, not behind later steps.  The
    step stream waits for the side stream only when it comes back to a snapshot buffer whose previous gather has not
    finished -- with two buffers that takes a gather longer than a whole logging interval -- and that wait is bracketed by
    its own event pair, so stats() can say how long the step stream really stood still.  Every exchange is bracketed by
    an event pair on the side stream (`collective_us`).  wait() joins the side stream and returns the gathered
    (done, reward) of the LAST submit in global env order.  Equal shards only (n_local envs on every rank).
    CPU tensors (gloo) take the same calls without streams."""

    MAX_TIMED = 512  # exchanges that get their own timing events (later ones run untimed)

    def __init__(self, n_local, device, world, buffers=2):
        import torch
        self.torch = torch
        self.world = int(world)
        self.n = int(n_local)
        self.nbuf = max(1, int(buffers))
        self.cuda = torch.device(device).type == "cuda"
        self.snap_done = [torch.zeros(self.n, dtype=torch.uint8, device=device) for _ in range(self.nbuf)]
        self.snap_reward = [torch.zeros(self.n, dtype=torch.float32, device=device) for _ in range(self.nbuf)]
        self.gath_done = [torch.zeros(self.world * self.n, dtype=torch.uint8, device=device) for _ in range(self.nbuf)]
        self.gath_reward = [torch.zeros(self.world * self.n, dtype=torch.float32, device=device) for _ in range(self.nbuf)]
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self.ready = [torch.cuda.Event() for _ in range(self.nbuf)] if self.cuda else None
        self.consumed = [torch.cuda.Event() for _ in range(self.nbuf)] if self.cuda else None
        self.in_use = [False] * self.nbuf
        self.submitted = 0
        self.last = 0
        self._coll = []        # (start, end) events on the side stream, or host seconds (gloo)
        self._waits = []       # (before, after) events around a step-stream wait
        self.not_ready_at_submit = 0

    # the snapshot the last submit took (bench.py checks its own shard inside the gathered vectors against it)
    @property
    def done(self):
        return self.snap_done[self.last]

    @property
    def reward(self):
        return self.snap_reward[self.last]

    def submit(self, done, reward):
        import time
        import torch.distributed as dist
        torch = self.torch
        b = self.submitted % self.nbuf
        many = self.world > 1 and dist.is_initialized()
        if self.cuda:
            cur = torch.cuda.current_stream(self.snap_done[b].device)
            if self.in_use[b]:
                # the gather that last read this snapshot buffer must be through before it is overwritten
                if not self.consumed[b].query():
                    self.not_ready_at_submit += 1
                if len(self._waits) < self.MAX_TIMED:
                    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    w0.record(cur)
                    cur.wait_event(self.consumed[b])
                    w1.record(cur)
                    self._waits.append((w0, w1))
                else:
                    cur.wait_event(self.consumed[b])
            self.snap_done[b].copy_(done, non_blocking=True)
            self.snap_reward[b].copy_(reward, non_blocking=True)
            self.ready[b].record(cur)
            timed = len(self._coll) < self.MAX_TIMED
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ready[b])
                if timed:
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record(self.side)
                if many:
                    dist.all_gather_into_tensor(self.gath_done[b], self.snap_done[b])
                    dist.all_gather_into_tensor(self.gath_reward[b], self.snap_reward[b])
                else:
                    self.gath_done[b].copy_(self.snap_done[b], non_blocking=True)
                    self.gath_reward[b].copy_(self.snap_reward[b], non_blocking=True)
                if timed:
                    t1.record(self.side)
                    self._coll.append((t0, t1))
                self.consumed[b].record(self.side)
            self.in_use[b] = True
        else:
            self.snap_done[b].copy_(done)
            self.snap_reward[b].copy_(reward)
            t0 = time.perf_counter()
            if many:
                dist.all_gather_into_tensor(self.gath_done[b], self.snap_done[b])
                dist.all_gather_into_tensor(self.gath_reward[b], self.snap_reward[b])
            else:
                self.gath_done[b].copy_(self.snap_done[b])
                self.gath_reward[b].copy_(self.snap_reward[b])
            if len(self._coll) < self.MAX_TIMED:
                self._coll.append(time.perf_counter() - t0)
        self.last = b
        self.submitted += 1

    def wait(self):
        if self.cuda:
            self.torch.cuda.current_stream(self.snap_done[0].device).wait_stream(self.side)
        return self.gath_done[self.last], self.gath_reward[self.last]

    def stats(self):
        """Call after a synchronise: per-exchange duration on the side stream and what the step stream waited for it."""
        if self.cuda:
            self.side.synchronize()
            coll = [a.elapsed_time(b) * 1e3 for a, b in self._coll]
            waits = [a.elapsed_time(b) * 1e3 for a, b in self._waits]
        else:
            coll = [c * 1e6 for c in self._coll]
            waits = []
        return {"exchanges": self.submitted, "buffers": self.nbuf,
                "collective_us_mean": (sum(coll) / len(coll)) if coll else None,
                "collective_us_max": max(coll) if coll else None,
                "step_stream_waits": len(waits), "step_stream_wait_us_total": sum(waits),
                "gather_unfinished_when_buffer_reused": self.not_ready_at_submit,
                "timed_on": "side stream (HIP events)" if self.cuda else "host clock (blocking gloo collective)"}
