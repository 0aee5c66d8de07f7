"""Multi-GPU sharding of integration frames — host-side plumbing.

Chunks are independent (``channelize_poly`` starts every chunk with zero PFB history and the reference
processes chunk pairs one at a time with no carried state, effex/effex.py:391-410), so each rank
integrates a contiguous range of the global chunk index on its own GPU with no data-path collective.
The only exchange is one sum-reduction of the exported accumulators — ``n_baselines*nchan + 1``
complex128 (raw cross-spectra sums + the spectra count; 64 KiB for 2 antennas) — through
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests), issued once
per integration, then ``fxc_finalize_sums`` on the root.  The reduce is latency-bound at this size.
"""


def chunk_range(rank, world_size, n_chunks):
    """Contiguous [lo, hi) of the global chunk index owned by ``rank``."""
    if not 0 <= rank < world_size:
        raise ValueError("rank {} outside world of {}".format(rank, world_size))
    return (rank * n_chunks) // world_size, ((rank + 1) * n_chunks) // world_size


def reduce_sums(sums, root=0, group=None, to_all=False):
    """Sum the exported accumulators across ranks (in place).  Returns ``sums``.

    complex128 is reduced through its float64 view so every backend accepts it.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sums
    flat = torch.view_as_real(sums) if sums.is_complex() else sums
    if to_all:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(flat, dst=root, op=dist.ReduceOp.SUM, group=group)
    return sums


class ShardedIntegrator(object):
    """One per rank: integrates this rank's chunks on its GPU, then reduces and finalises."""

    def __init__(self, plan, rank=0, world_size=1, group=None):
        self.plan, self.rank, self.world_size, self.group = plan, rank, world_size, group
        self.sums = plan.new_sums()

    def my_range(self, n_chunks):
        return chunk_range(self.rank, self.world_size, n_chunks)

    def accumulate(self, x_local):
        return self.plan.fx_accumulate(x_local)

    def reduce(self, root=0, to_all=False):
        """Export this rank's accumulator and sum it across ranks (async on the device until the
        collective's own synchronisation)."""
        import torch
        self.plan.acc_export(self.sums)
        self.plan.sync()           # the plan's stream produced `sums`; the collective runs on torch's
        reduce_sums(self.sums, root=root, group=self.group, to_all=to_all)
        if self.sums.is_cuda:
            torch.cuda.current_stream(self.sums.device).synchronize()
        return self.sums

    def finalize(self, mode="SPECTRUM", bandwidth=1.0, root=0, to_all=False):
        """Returns the integrated visibilities on the root (every rank if ``to_all``), else None."""
        self.reduce(root=root, to_all=to_all)
        out = None
        if to_all or self.rank == root:
            out = self.plan.finalize_sums(self.sums, mode, bandwidth)
        self.plan.acc_reset()
        return out
