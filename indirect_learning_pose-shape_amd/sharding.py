"""Mesh sharding across GPUs: the decoder has no parameters, every op is
independent per batch element, so ranks take contiguous row ranges and exchange nothing."""


def shard_range(global_batch: int, rank: int, world: int):
    """Rows [lo, hi) of rank `rank`: contiguous, sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world) or global_batch < 0:
        raise ValueError("bad shard request: batch=%d rank=%d world=%d" % (global_batch, rank, world))
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
