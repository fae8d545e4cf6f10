"""Sharding of independent Ed448 operations over the GPUs of one node.

The path shards embarrassingly: operation i of a batch of n belongs to
exactly one rank, no data-path collective.  Two layouts are used:
  * strong: one global batch, contiguous slices  [g*n/G, (g+1)*n/G)   (shard_range)
  * weak:   every rank owns its own batch of the same size (bench.py, per-GPU throughput)
The only collectives are control-plane: a barrier around the timed region, MAX of the elapsed
time and SUM of per-rank counters (accepted signatures, processed ops)."""


def shard_range(n, rank, world):
    """Contiguous slice of [0, n) owned by `rank`; slices are disjoint, ordered and cover [0, n)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d not in [0, %d)" % (rank, world))
    return (n * rank) // world, (n * (rank + 1)) // world


def max_over_ranks(value, dist=None, device=None):
    """MAX-reduce a python float over the process group (identity without one)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    """SUM-reduce a python int over the process group (identity without one)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
