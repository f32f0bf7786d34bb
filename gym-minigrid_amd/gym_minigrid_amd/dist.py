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


def init_process_group(backend=None):
    """Rendezvous from RANK/WORLD_SIZE/MASTER_* (torchrun).  backend: 'nccl' (RCCL) on GPUs, 'gloo' on CPU."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank_info()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this pool
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


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

    submit(done, reward) snapshots the step's per-env done (u8) / reward (f32) vectors into private buffers on the
    caller's stream (the env overwrites its own on the next step), then all-gathers the snapshots of every rank on a
    SIDE stream: RCCL's work is ordered behind the snapshot by an event, not behind later steps, and the step stream
    never waits for it.  wait() joins the side stream and returns the gathered (done, reward) in global env order.
    Equal shards only (n_local envs on every rank).  CPU tensors (gloo) take the same calls without streams."""

    def __init__(self, n_local, device, world):
        import torch
        self.torch = torch
        self.world = int(world)
        self.n = int(n_local)
        self.cuda = torch.device(device).type == "cuda"
        self.done = torch.zeros(self.n, dtype=torch.uint8, device=device)
        self.reward = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.all_done = torch.zeros(self.world * self.n, dtype=torch.uint8, device=device)
        self.all_reward = torch.zeros(self.world * self.n, dtype=torch.float32, device=device)
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self.ready = torch.cuda.Event() if self.cuda else None
        self.submitted = 0

    def submit(self, done, reward):
        import torch.distributed as dist
        torch = self.torch
        if self.cuda:
            cur = torch.cuda.current_stream(self.done.device)
            cur.wait_stream(self.side)          # the previous gather has read the snapshot buffers
        self.done.copy_(done, non_blocking=True)
        self.reward.copy_(reward, non_blocking=True)
        many = self.world > 1 and dist.is_initialized()
        if self.cuda:
            self.ready.record(cur)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ready)
                if many:
                    dist.all_gather_into_tensor(self.all_done, self.done)
                    dist.all_gather_into_tensor(self.all_reward, self.reward)
                else:
                    self.all_done.copy_(self.done, non_blocking=True)
                    self.all_reward.copy_(self.reward, non_blocking=True)
        elif many:
            dist.all_gather_into_tensor(self.all_done, self.done)
            dist.all_gather_into_tensor(self.all_reward, self.reward)
        else:
            self.all_done.copy_(self.done)
            self.all_reward.copy_(self.reward)
        self.submitted += 1

    def wait(self):
        if self.cuda:
            self.torch.cuda.current_stream(self.done.device).wait_stream(self.side)
        return self.all_done, self.all_reward
