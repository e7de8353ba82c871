"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Elementwise batches and batch inversion partition into independent contiguous element
ranges: no data-path collective.  Mat-vec partitions rows (the reference's own parallel
axis: rayon over `par_chunks_exact(num_cols)`, src/matrix.rs:170-184); x is replicated
and the only exchange step is an all-gather of the per-rank output slices.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) slice of n units for `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def padded_rows(rows, world):
    """Rows per rank after padding so every rank gathers the same count."""
    return (rows + world - 1) // world


def matvec_rowsharded(local_matvec, A_local, rows_total, cols, x, n64, group=None):
    """y = A x with A row-sharded.  `A_local` holds this rank's rows (shard_range of
    rows_total); `local_matvec(A_local, nrows, cols, x) -> (nrows, n64)` is the per-GPU
    kernel call (MontyField.matvec_dev wrapped by the caller).  Returns the full
    (rows_total, n64) result on every rank (all-gather)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(rows_total, rank, world)
    y_local = local_matvec(A_local, hi - lo, cols, x)
    per = padded_rows(rows_total, world)
    pad = torch.zeros((per, n64), dtype=y_local.dtype, device=y_local.device)
    pad[: hi - lo] = y_local
    gathered = torch.empty((world * per, n64), dtype=y_local.dtype, device=y_local.device)
    dist.all_gather_into_tensor(gathered, pad, group=group)
    parts = []
    for r in range(world):
        rlo, rhi = shard_range(rows_total, r, world)
        parts.append(gathered[r * per: r * per + (rhi - rlo)])
    return torch.cat(parts, dim=0)


def elementwise_sharded(local_op, n_total, *operands, group=None):
    """Applies `local_op` to this rank's contiguous slice of every operand; no collective.
    Returns (lo, hi, local_result)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_total, rank, world)
    return lo, hi, local_op(*[o[lo:hi] for o in operands])
