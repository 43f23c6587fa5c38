"""Multi-GPU decomposition of the batched PNN path: independent transform blocks are split contiguously over
ranks (one process per GPU), weights are replicated, and there is no collective on the data path.  The only
exchanges are the bench barrier / max-over-ranks clock and the optional result gather for a single consumer.
(the reference itself has no multi-device code.)"""


import os
import time


def rank_env(environ=None):
    """(rank, local_rank, world_size) as torch.distributed.run exports them; (0, 0, 1) for a plain process."""
    e = os.environ if environ is None else environ
    return int(e.get("RANK", "0")), int(e.get("LOCAL_RANK", "0")), int(e.get("WORLD_SIZE", "1"))


def init_ranks(backend, device=None):
    """Joins the job's process group when WORLD_SIZE > 1 (backend "nccl" = RCCL on the GPUs, "gloo" on the CPU) and
    returns the torch.distributed module, or None for a single process.  Rendezvous defaults to 127.0.0.1."""
    rank, _, world = rank_env()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {"device_id": device} if device is not None else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def timed_steps(step, n_steps, sync, dist=None, device=None):
    """The bench contract's timed region: barrier + device synchronisation, EXACTLY n_steps calls of `step`, device
    synchronisation + barrier, and the job's time = the MAX over ranks (seconds, the same value on every rank).
    `sync` waits for this rank's device work (torch.cuda.synchronize on a GPU rank, a no-op on the CPU)."""
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    return max_over_ranks(time.perf_counter() - t0, dist, device)


def shard_bounds(n_items, rank, world_size):
    """[begin, end) of `rank`'s contiguous shard; the first n % world ranks take one extra item."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world_size")
    base, extra = divmod(int(n_items), world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def max_over_ranks(seconds, dist=None, device=None):
    """The job's step time is the slowest rank's (bench contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_predictions(local, n_total, dist=None):
    """All ranks' [n_local, w, w] predictions concatenated in rank order on every rank (equal or ragged shards)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    world = dist.get_world_size()
    sizes = [e - b for b, e in (shard_bounds(n_total, r, world) for r in range(world))]
    biggest = max(sizes)                                  # all_gather wants equal shapes: pad ragged shards
    padded = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    outs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(outs, padded)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)
