"""Sharded multistart evaluation and best-start selection.

The reference runs its multistart loop sequentially on one candidate batch at a time and keeps the
best one (bo.py:142-161).  Candidate batches are independent, so they shard over ranks with no
data-path collective; the only exchange is one all-gather of a small per-rank record

    (value: f64, global index: i64, X: f64[q*d])          976 bytes at q=20, d=6

followed by a local arg-min with ties going to the lowest global index, so that every rank count
selects the same start.  One process per GPU; the process group's backend is RCCL ("nccl") on GPUs
and gloo in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous block partition of `total` candidates: [lo, hi) of `rank`."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def make_record(best_value, best_index, X_best):
    """Pack (value, index, X) into one float64 tensor; the int64 index travels bit-exactly."""
    rec = torch.empty(2 + X_best.numel(), dtype=torch.float64, device=X_best.device)
    rec[0] = best_value.reshape(())
    rec[1:2] = best_index.reshape(1).to(torch.int64).view(torch.float64)
    rec[2:] = X_best.reshape(-1)
    return rec


def select_best(rec, group=None, force=False):
    """All-gather the per-rank records and return (value, index, X_flat) of the winner.
    Without an initialised process group the local record wins.  `force`: run the collective also
    in a group of one rank (exercises the RCCL path on a single GPU)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        world = dist.get_world_size(group)
        # RCCL gathers device tensors in place; gloo (CPU tests, one-GPU rehearsals) gets host copies
        src = rec.contiguous() if dist.get_backend(group) != 'gloo' else rec.detach().cpu().contiguous()
        flat = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(flat, src, group=group)
        allrec = flat.reshape(world, rec.numel()).to(rec.device)
    else:
        allrec = rec.reshape(1, -1)
    if allrec.is_cuda:                      # one launch on the device
        from . import engine
        row = engine.select_record(allrec)
        return row[0], row[1:2].view(torch.int64).reshape(()), row[2:]
    values = allrec[:, 0]
    indices = allrec[:, 1].contiguous().view(torch.int64)
    # NaN values never win; ties -> lowest global index
    values = torch.where(torch.isnan(values), torch.full_like(values, float('inf')), values)
    vmin = values.min()
    cand = torch.where(values == vmin, indices, torch.full_like(indices, torch.iinfo(torch.int64).max))
    # the winner is picked on the device: no host synchronisation inside a step
    win = torch.argmin(cand).reshape(1)
    row = allrec.index_select(0, win).reshape(-1)
    return row[0], indices.index_select(0, win).reshape(()), row[2:]


def evaluate_and_select(gp, X_local, index_base=0, group=None, force_collective=False):
    """Evaluate value+gradient of the local shard on the GPU, then pick the global best start.
    Returns (value [B], grad [B,q,d], status [B], (best_value, best_index, best_X [q,d]))."""
    from . import engine
    value, grad, status, _ = gp.value_grad(X_local, want_flags=False)
    rec = engine.best_record(value, X_local, index_base)      # one launch: arg-min + record
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective):
        best_value, best_index, best_X = select_best(rec, group, force=force_collective)
    else:                                                     # one rank: its record is the winner
        best_value, best_index, best_X = rec[0], rec[1:2].view(torch.int64).reshape(()), rec[2:]
    return value, grad, status, (best_value, best_index, best_X.reshape(X_local.shape[1:]))
