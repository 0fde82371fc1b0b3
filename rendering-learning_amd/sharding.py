"""Framebuffer sharding across ranks: pixels are independent units, so image rows are
interleaved round-robin over the G ranks (row r -> rank r mod G; contiguous bands would put all the sky
on one GPU), the scene is replicated, and the ONLY exchange step is one gather of the rows to rank 0
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).  Results are bit-identical for
any G because per-pixel work does not depend on the partition."""
import torch
import torch.distributed as dist


def rows_for(height, row_first, row_step):
    return 0 if row_first >= height else (height - row_first + row_step - 1) // row_step


def shard_spec(height, rank, world_size):
    """(row_first, row_step, n_rows, max_rows) of this rank's shard."""
    return rank, world_size, rows_for(height, rank, world_size), rows_for(height, 0, world_size)


def gather_frame(shard, height, rank, world_size, frame=None, gathered=None):
    """shard: [max_rows, W, 3] tensor holding this rank's rows (padded to max_rows).
    Returns the assembled [H, W, 3] frame on rank 0 (None elsewhere)."""
    if world_size == 1:
        return shard[:height]
    if rank == 0 and gathered is None:
        gathered = [torch.empty_like(shard) for _ in range(world_size)]
    dist.gather(shard, gathered if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    if frame is None:
        frame = torch.empty((height,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    for g in range(world_size):
        n = rows_for(height, g, world_size)
        frame[g::world_size] = gathered[g][:n]
    return frame
