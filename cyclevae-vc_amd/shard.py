"""Data-parallel sharding of utterance batches.

Eval / decode: batch rows are independent (the reference fans out whole file lists per GPU with no communication,
src/bin/decode_gru-cyclevae_gauss.py:190-195), so rank r takes a contiguous block of rows, weights are replicated and
NO data-path collective is needed.  The only collectives a job uses are a barrier and a MAX over ranks of the elapsed
time for reporting.
"""


def shard_rows(n_rows, world, rank):
    """Contiguous [lo, hi) block of rank `rank`; the first n_rows % world ranks get one extra row (np.array_split rule,
    which is what the reference applies to its file lists)."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(seconds, dist=None, device=None):
    """Wall time of the slowest rank (what a whole-job throughput must be divided by)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
