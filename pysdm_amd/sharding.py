"""Sharding of a multi-cell domain over the GPUs of one node (one process per GPU).

Pairs never span cells (`find_pairs`, PySDM/backends/impl_numba/methods/pair_methods.py:34-55)
and every per-cell quantity (`dt_left`, counters, `norm_factor`) is private to its cell, so the
collision step has no exchange step: each rank owns a contiguous block of cells with their
super-droplets and runs the unchanged (fused) step on that sub-domain.  `torch.distributed`
(backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests) is used only to assemble
per-cell diagnostics; nothing on the data path is communicated.
"""
import numpy as np


def cell_block(n_cell, rank, world_size):
    """contiguous block [first, last) of cells owned by `rank` (sizes differ by at most one)"""
    base, extra = divmod(n_cell, world_size)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def shard_attributes(attributes, n_cell, rank, world_size):
    """selects the super-droplets of this rank's cells (keeping their order) and renumbers the
    cell ids to the local range; returns (local attributes, global indices of the selected SDs,
    (first, last) cell of the block)"""
    first, last = cell_block(n_cell, rank, world_size)
    cell_id = np.asarray(attributes["cell id"])
    mine = np.flatnonzero((cell_id >= first) & (cell_id < last))
    local = {}
    for key, value in attributes.items():
        value = np.asarray(value)
        local[key] = value[..., mine].copy()
    local["cell id"] = local["cell id"] - first
    return local, mine, (first, last)


def gather_per_cell(local_values, n_cell, world_size, device=None):
    """all ranks get the full per-cell array (concatenation of the blocks in rank order)"""
    import torch  # pylint: disable=import-outside-toplevel
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel

    local = torch.as_tensor(np.asarray(local_values))
    if device is not None:
        local = local.to(device)
    sizes = [cell_block(n_cell, r, world_size) for r in range(world_size)]
    width = max(b - a for a, b in sizes)
    padded = torch.zeros(width, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    gathered = [torch.empty_like(padded) for _ in range(world_size)]
    dist.all_gather(gathered, padded)
    parts = [g[: b - a].cpu().numpy() for g, (a, b) in zip(gathered, sizes)]
    return np.concatenate(parts)


def global_sum(value, device=None):
    """all-reduce (sum) of a scalar / small array, e.g. candidate pairs or super-droplet counts"""
    import torch  # pylint: disable=import-outside-toplevel
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel

    tensor = torch.as_tensor(np.asarray(value, dtype=np.float64))
    if device is not None:
        tensor = tensor.to(device)
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor.cpu().numpy()
