"""Multi-GPU decomposition of the batched PNN path: independent transform blocks are split contiguously over
ranks (one process per GPU), weights are replicated, and there is no collective on the data path.  The only
exchanges are the bench barrier / max-over-ranks clock and the optional result gather for a single consumer.
(the reference itself has no multi-device code.)"""


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
