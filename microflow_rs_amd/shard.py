"""Batch sharding for the multi-GPU path.

The path shards as independent units: each inference is independent, the model
(~0.3 MB of weights) is replicated per GPU, and rank r of R owns a contiguous slice
of the global batch.  There is NO collective on the data path; torch.distributed
(RCCL on GPUs, gloo in the CPU tests) is used only for the timing barrier, the
max-over-ranks reduction of the elapsed time and an after-the-fact gather of
per-shard output checksums.
"""


def shard_range(total, rank, world):
    """(first, count) of rank's contiguous slice; slices differ by at most one unit."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def gather_checksums(dist, checksum, device=None):
    """all_gather one 63-bit checksum per rank (works for nccl and gloo)."""
    import torch
    t = torch.tensor([checksum & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [int(v.item()) for v in out]


def max_over_ranks(dist, seconds, device=None):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
