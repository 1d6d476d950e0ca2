"""Multi-GPU film reduce: plumbing only (torch.distributed's "nccl" backend is RCCL on ROCm).

The path shards by film tiles; the single exchange step is a sum of the per-rank
{X,Y,Z,weight} framebuffers, the quantity Film::merge_film_tile accumulates
(src/core/film/film.rs:219-241).  The buffer lives in the library's HBM allocation; it is
wrapped zero-copy as a torch tensor via the CUDA array interface so RCCL reduces it in place.
"""
import torch


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def wrap_device_floats(ptr, n, device_index):
    """Zero-copy float32 tensor over n floats at device pointer ptr."""
    return torch.as_tensor(_DevArray(ptr, n), device=torch.device("cuda", device_index))


def all_reduce_xyzw(t, group=None):
    """The path's one collective: element-wise sum of the per-rank {X,Y,Z,weight} films.
    Works for any backend (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def reduce_film(ctx, device_index, group=None, op="all"):
    """Sum the XYZW film over all ranks in place and mark it authoritative."""
    import torch.distributed as dist
    ptr, n = ctx.film_device_xyzw()
    t = wrap_device_floats(ptr, n, device_index)
    if op == "all":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, group=group)
    ctx.film_commit_xyzw()
    return t


def partition_tiles(tiles, rank, world):
    """Round-robin deal of the reference's 16x16 tiles (sampler.rs:271-289) to ranks."""
    return tiles[rank::world]
