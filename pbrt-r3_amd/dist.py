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


_stage = {}


def reduce_film(ctx, device_index, group=None, op="all", staged=True):
    """Sum the XYZW film over all ranks in place and mark it authoritative.

    staged=True reduces a torch-owned buffer (two 16 MiB device copies at 1024^2, ~10 us each) so that RCCL
    only ever sees memory from torch's allocator; staged=False hands RCCL the library's own allocation."""
    import torch.distributed as dist
    ptr, n = ctx.film_device_xyzw()
    film = wrap_device_floats(ptr, n, device_index)
    t = film
    if staged:
        key = (device_index, n)
        if key not in _stage:
            _stage[key] = torch.empty(n, dtype=torch.float32, device=torch.device("cuda", device_index))
        t = _stage[key]
        t.copy_(film)
    if op == "all":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, group=group)
    if staged:
        film.copy_(t)
    torch.cuda.synchronize(device_index)      # the library's next film pass runs on its own stream
    ctx.film_commit_xyzw()
    return film


def partition_tiles(tiles, rank, world):
    """Round-robin deal of the reference's 16x16 tiles (sampler.rs:271-289) to ranks."""
    return tiles[rank::world]
