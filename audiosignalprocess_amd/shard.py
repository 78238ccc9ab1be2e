"""Multi-GPU sharding of independent streams (one process per GPU).

Streams never interact, so the N-GPU job is N disjoint contiguous shards of
stream ids and there is NO collective on the data path (SURVEY.md section
8(e)).  torch.distributed is used only to bracket timed regions with a barrier
and to take the max over ranks.
"""


def shard_streams(rank, world, streams_per_gpu):
    """(first stream id, count) owned by `rank`: stream s lives on GPU s // streams_per_gpu."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return rank * streams_per_gpu, streams_per_gpu


def max_over_ranks(dist, values, device="cpu"):
    """Element-wise MAX of a small list of floats over all ranks (dist may be None)."""
    import torch

    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]
