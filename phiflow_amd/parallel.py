"""
Multi-GPU execution of the fluid step: batch-parallel sharding.

PhiML batch dimensions are independent simulations (separate alpha / beta / convergence per entry), so the path shards
over the batch with NO data-path collective. One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI);
the only communication is ONE all-reduce per step of the global residual norm -- an 8-byte payload, latency-bound.
"""
from typing import Optional, Tuple

import torch


def local_batch_range(total_batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """ contiguous block distribution of `total_batch` simulations over `world_size` ranks -> [begin, end) of `rank` """
    base, extra = divmod(int(total_batch), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_field(field, rank: int, world_size: int):
    """ the simulations of a batched `Field` that `rank` owns """
    begin, end = local_batch_range(field.batch_size, rank, world_size)
    if field.is_staggered:
        vals = [t[begin:end].contiguous() for t in field.values]
    else:
        vals = field.values[begin:end].contiguous()
    out = field.with_values(vals)
    out.batched = True
    return out


def global_relative_residual(backend, local_batch: int, group=None) -> torch.Tensor:
    """ max over ALL simulations of ||r|| / ||rhs|| of the most recent pressure solve: device-side residuals of the local
    shard (`phihip_solve_relative_residual`, no host sync) followed by the step's single all-reduce (MAX). Returns a 1-element tensor
    on the backend's device. """
    rel = backend.zeros((1,), torch.float64)
    if local_batch > 0:
        backend.ctx.solve_relative_residual(local_batch, rel.data_ptr(), backend.stream())
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(rel, op=dist.ReduceOp.MAX, group=group)
    return rel
