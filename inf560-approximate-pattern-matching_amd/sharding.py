"""One-process-per-GPU text sharding used by bench.py (and covered by the
world_size-2 gloo test): owner-computes ranges of window starts, an
(m_max-1)-byte halo, and ONE all-reduce (sum) of the P partial counts --
the exchange step that replaces the reference's MPI_Send/MPI_Recv + manual sum
(/root/reference/src/database_over_ranks.c:141-195).  Truncation happens only at
the end of the WHOLE text, never at a shard end."""
from . import shard_range


def rank_shard(n_total, k, m_max, rank, world):
    """(own_begin, own_end, text_lo, text_hi): the rank decides windows starting in
    [own_begin, own_end) and needs text bytes [text_lo, text_hi)."""
    ob, oe = shard_range(n_total, k, rank, world)
    lo = ob
    hi = min(n_total, oe + max(m_max, 1) - 1) if oe > ob else ob
    return ob, oe, lo, hi


def allreduce_counts(counts):
    """In-place sum of the per-rank partial counts (torch int64 tensor) over all ranks.
    backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
