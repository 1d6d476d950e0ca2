"""The N>1 path on CPU: two gloo ranks split the reference's tiles round-robin, each produces
its partial XYZW film (here with the oracle standing in for the device film), one all-reduce
sums them, and the result is the single-rank film bit for bit (disjoint tiles, box filter)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import oracle_lib
    pkg = importlib.import_module("pbrt-r3_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = pkg.scenes.cornell_box(res=32, spp=4)
    sc = oracle_lib.load().scene(sd)
    tiles = pkg.scenes.all_tiles(sc.info)
    mine = pkg.dist.partition_tiles(tiles, rank, world)
    assert len(mine) in (len(tiles) // world, len(tiles) // world + 1)
    xyzw, _, _ = sc.render(mine, threads=1)
    t = torch.from_numpy(xyzw)
    pkg.dist.all_reduce_xyzw(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_and_reduce(tmp_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import oracle_lib
    pkg = importlib.import_module("pbrt-r3_amd")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    reduced = np.load(tmp_path / "reduced.npy")
    sc = oracle_lib.load().scene(pkg.scenes.cornell_box(res=32, spp=4))
    full, _, _ = sc.render(threads=2)
    assert np.array_equal(reduced.view(np.uint32), full.view(np.uint32))


def test_partition_is_a_disjoint_cover():
    import importlib
    pkg = importlib.import_module("pbrt-r3_amd")
    tiles = list(range(4225))
    for world in (1, 2, 4, 8):
        parts = [pkg.dist.partition_tiles(tiles, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == tiles
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
