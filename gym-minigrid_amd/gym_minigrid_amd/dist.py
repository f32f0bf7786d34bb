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


def gather_done_reward(done, reward):
    """All-gather the per-env done (uint8) and reward (f32) vectors of every shard, in global env order.
    Shards may differ in size by one env; they are padded to the largest for the collective."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return done, reward
    world = dist.get_world_size()
    n = torch.tensor([done.shape[0]], device=done.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pd = torch.zeros(m, dtype=done.dtype, device=done.device)
    pr = torch.zeros(m, dtype=reward.dtype, device=reward.device)
    pd[: done.shape[0]] = done
    pr[: reward.shape[0]] = reward
    gd = [torch.empty_like(pd) for _ in range(world)]
    gr = [torch.empty_like(pr) for _ in range(world)]
    dist.all_gather(gd, pd)
    dist.all_gather(gr, pr)
    return (torch.cat([g[:s] for g, s in zip(gd, sizes)]), torch.cat([g[:s] for g, s in zip(gr, sizes)]))
