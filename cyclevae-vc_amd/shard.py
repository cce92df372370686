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


def spawn_ranks(script, argv, n_ranks, stdout=None):
    """Run `script argv...` as n_ranks processes of ONE node under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1 with a free port, HSA_ENABLE_IPC_MODE_LEGACY=0 kept for RCCL's dmabuf IPC) and return its exit code; the ranks
    inherit stdout, so rank 0's JSON line passes through.  What `python bench.py --gpus N` does when no launcher started it."""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)
    return subprocess.call(cmd, env=env, stdout=stdout)


def launched_world(n_expected):
    """(world, rank, local_rank) from the launcher's environment; asserts that the launcher started as many ranks as --gpus says."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == n_expected, "--gpus %d but the launcher started %d ranks" % (n_expected, world)
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def max_over_ranks(seconds, dist=None, device=None, force=False):
    """Wall time of the slowest rank (what a whole-job throughput must be divided by)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(seconds, dist=None, device=None, force=False):
    """Every rank's wall time, in rank order (diagnostics next to max_over_ranks: which rank was the slow one)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return [seconds]
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(v.item()) for v in out]


def allreduce_gradients(params, dist):
    """Data-parallel training: each rank back-propagates the SUM loss over its utterance rows
    (reference train_gru_cyclevae_gauss_batch.py:1403-1408 sums over utterances), so the global gradient is the sum over
    ranks: ONE all-reduce of a flat fp32 bucket (9.57 M floats = 38.3 MB at hu1024) per step, RCCL over xGMI on GPUs.
    Every rank then applies the identical optimiser step.  Returns the number of floats reduced."""
    import torch
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sum(g.numel() for g in grads)
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    for g, r in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(r)
    return flat.numel()


class FlatGradients(object):
    """All trainable gradients of a step as views of ONE flat fp32 buffer, so the data-parallel all-reduce needs no copy in or
    out: `p.grad` of every parameter is a slice of `flat`; zero with `flat.zero_()` (the optimiser's zero_grad must keep the
    tensors: set_to_none=False).  One RCCL all-reduce of 9.57 M floats (38.3 MB at hu1024) per step.

    Why one bucket and no overlap with the backward: the chain applies each network 4 (encoder) / 6 (decoder) times, so a
    parameter's gradient is complete only when the FIRST pass of its network has been back-propagated -- the encoder's at the very
    end of the backward, the decoder's one pass earlier.  There is nothing to overlap 38 MB (~0.3 ms over xGMI) with."""

    def __init__(self, params):
        import torch
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero(self):
        self.flat.zero_()

    def allreduce(self, dist, force=False):
        """force: issue the collective even in a group of one rank (bench.py --force-dist: RCCL executes on a one-GPU box)."""
        if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
            return self.flat.numel()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        return self.flat.numel()
