"""GPU side of the multi-GPU path on one device: the zero-copy wrap of the library's film, the staged and in-place RCCL
reduce of pbrt-r3_amd/dist.py over a world-size-1 "nccl" group, film_commit_xyzw, the library's own pt_film_allreduce over a
communicator made by ncclCommInitAll, and the C++ `pbrt_gpu --gpus 1` driver (the same code path N GPUs take).
World size 1 makes the sum the identity, so the film must come back bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import bits, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 300)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _rendered(ctx):
    ctx.upload(scenes.cornell_box(res=64, spp=8))
    ctx.film_clear(); ctx.render()
    return ctx.film_xyzw()


@pytest.mark.parametrize("staged", [True, False])
def test_reduce_film_world1_is_identity(gpu_ctx, pkg, nccl_group, staged):
    want = _rendered(gpu_ctx)
    assert want[..., 3].sum() > 0
    film = pkg.dist.reduce_film(gpu_ctx, 0, staged=staged)
    assert film.numel() == want.size
    got = gpu_ctx.film_xyzw()
    assert np.array_equal(bits(got), bits(want))


def test_commit_makes_the_reduced_buffer_authoritative(gpu_ctx, pkg, nccl_group):
    want = _rendered(gpu_ctx)
    ptr, n = gpu_ctx.film_device_xyzw()
    t = pkg.dist.wrap_device_floats(ptr, n, 0)
    t.mul_(2.0)                                   # what a reduce over two identical ranks would leave
    import torch
    torch.cuda.synchronize(0)
    assert np.array_equal(bits(gpu_ctx.film_xyzw()), bits(want))          # not committed: rebuilt from the rank's own sums
    t.mul_(2.0); torch.cuda.synchronize(0)
    gpu_ctx.film_commit_xyzw()
    assert np.array_equal(bits(gpu_ctx.film_xyzw()), bits(want * np.float32(2.0)))
    rgb2 = gpu_ctx.film_rgb()                      # weights doubled too: the resolved image is unchanged
    gpu_ctx.film_clear(); gpu_ctx.render()         # a new render supersedes the committed buffer
    again = gpu_ctx.film_xyzw()                    # (edge-split samples go through float atomics: last bits may differ between renders)
    assert np.array_equal(bits(again[..., 3]), bits(want[..., 3])) and np.allclose(again, want, rtol=2e-6, atol=1e-7)
    assert np.allclose(rgb2, gpu_ctx.film_rgb(), rtol=2e-6, atol=1e-7)


def test_library_side_allreduce(gpu_ctx):
    """pt_film_allreduce with a communicator from ncclCommInitAll -- no torch in the data path."""
    want = _rendered(gpu_ctx)
    rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    try:
        gpu_ctx.film_allreduce(comm.value)              # all-reduce
        assert np.array_equal(bits(gpu_ctx.film_xyzw()), bits(want))
        gpu_ctx.film_clear(); gpu_ctx.render()
        want2 = gpu_ctx.film_xyzw()                     # a second render: its edge-split samples may differ from the first in the last bits
        gpu_ctx.film_allreduce(comm.value, root=0)      # reduce to rank 0
        assert np.array_equal(bits(gpu_ctx.film_xyzw()), bits(want2))
    finally:
        rccl.ncclCommDestroy(comm)


def test_cpp_driver_gpus_1(tmp_path):
    """`pbrt_gpu --gpus 1`: threads + ncclCommInitAll + pt_film_allreduce; same image as the single-context run."""
    exe = os.path.join(ROOT, "pbrt-r3_amd", "csrc", "pbrt_gpu")
    scene = os.path.join(ROOT, "tests", "scenes", "cornell.pbrt")
    a, b = str(tmp_path / "a.pfm"), str(tmp_path / "b.pfm")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.check_call([exe, scene, "-o", a, "--pixelsamples", "4", "--quiet"], env=env)
    subprocess.check_call([exe, scene, "-o", b, "--pixelsamples", "4", "--quiet", "--gpus", "1"], env=env)
    def pfm(path):
        with open(path, "rb") as f:
            assert f.readline().strip() == b"PF"
            w, h = map(int, f.readline().split())
            f.readline()
            return np.frombuffer(f.read(), "<f4").reshape(h, w, 3)
    ia, ib = pfm(a), pfm(b)        # two renders: pixels with edge-split samples (float atomics) may differ in the last bits
    assert ia.shape == ib.shape and np.allclose(ia, ib, rtol=2e-6, atol=1e-7) and (ia.view(np.uint32) == ib.view(np.uint32)).mean() > 0.95


def test_cpp_driver_two_ranks_on_one_device(tmp_path):
    """BASELINE config 3's flow rehearsed on the one GPU a box has: `pbrt_gpu --devices 0,0` runs TWO ranks (host threads, one library
    context each, tiles dealt round-robin, both rendering concurrently on device 0), every rank reaches the meeting point, and the
    films are summed through the host (pt_film_add_xyzw) because RCCL refuses two ranks on one device.  Disjoint tiles under the box
    filter: every pixel's weight comes from one rank, so the reduced film's weights are bit-equal to the single-context film's and the
    colours agree to the last bits (edge-split samples go through float atomics in either run).  Three ranks as well (4 225 tiles do
    not divide evenly)."""
    exe = os.path.join(ROOT, "pbrt-r3_amd", "csrc", "pbrt_gpu")
    scene = os.path.join(ROOT, "tests", "scenes", "cornell.pbrt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = str(tmp_path / "one.xyzw")
    subprocess.check_call([exe, scene, "-o", str(tmp_path / "one.pfm"), "--xyzw", one, "--pixelsamples", "8", "--quiet"], env=env)
    want = np.fromfile(one, np.float32).reshape(-1, 4)
    assert want[:, 3].sum() > 0
    for devices in ("0,0", "0,0,0", "0,0,0,0,0,0,0,0"):        # eight ranks: the scaling run's largest world, 528-529 tiles each at 1024 x 1024
        out = str(tmp_path / ("n%d.xyzw" % len(devices)))
        r = subprocess.run([exe, scene, "-o", str(tmp_path / "n.pfm"), "--xyzw", out, "--pixelsamples", "8", "--devices", devices, "--stats"],
                           env=env, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = np.fromfile(out, np.float32).reshape(-1, 4)
        assert np.array_equal(bits(got[:, 3]), bits(want[:, 3]))                  # every camera sample exactly once
        assert np.allclose(got, want, rtol=2e-6, atol=1e-7) and (bits(got) == bits(want)).mean() > 0.95
        ranks = [ln for ln in r.stderr.splitlines() if ln.strip().startswith("rank ")]
        assert len(ranks) == devices.count(",") + 1 and all("render_ms" in ln and "reduce_ms" in ln for ln in ranks)
        tiles = [int(ln.split("tiles")[1].split()[0]) for ln in ranks]
        assert sum(tiles) > 0 and max(tiles) - min(tiles) <= 1                     # round-robin deal


def test_film_add_xyzw_is_the_host_staged_sum(gpu_ctx):
    want = _rendered(gpu_ctx)
    gpu_ctx.film_add_xyzw(want)                       # a second rank with the same film
    assert np.array_equal(bits(gpu_ctx.film_xyzw()), bits(want + want))


def _bench_line(args, env_extra, timeout=600):
    import json
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout            # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_rehearsal_four_ranks_on_one_gpu():
    """bench.py's N > 1 flow end to end, four ranks on this one GPU (BENCH_REHEARSE=gloo: RCCL refuses two ranks on one device, so the film sum
    goes through the host; never a judged number).  The line must vouch for itself: one per_rank entry per rank, the ranks' tiles add up to the
    frame's, the reduced film holds every camera sample exactly once on every rank, and the oracle agrees with the REDUCED film on tiles of the
    first and of the last rank.  (Four ranks, not eight: a GPU box allows six processes on its card, this test process included; the
    eight-rank deal runs as threads of one process in test_cpp_driver_two_ranks_on_one_device.)"""
    d = _bench_line(["--gpus", "4", "--triangles", "20000", "--res", "256", "--spp", "16", "--steps", "1", "--warmup", "1", "--no-spp1024"], {"BENCH_REHEARSE": "gloo"})
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["value"] > 0
    pr = d["per_rank"]
    assert [e["rank"] for e in pr] == [0, 1, 2, 3]
    assert sum(e["tiles"] for e in pr) == d["tiles_total"] == 17 * 17 and max(e["tiles"] for e in pr) - min(e["tiles"] for e in pr) <= 1
    assert all(e["render_ms"] > 0 and e["rays_per_step"] > 0 for e in pr)
    assert sum(e["rays_per_step"] for e in pr) == d["config"]["rays_per_step"]
    assert d["reduced_film_ok"] is True and d["reduced_film"]["pixels_under_half_weight"] == 0
    assert abs(d["reduced_film"]["weight_sum"] - d["reduced_film"]["weight_sum_expected"]) <= 64.0          # a lost tile would be 256 x spp
    assert "gloo" in d["reduce"]
    assert d["parity"]["rel_l2"] <= 1e-3 and d["parity"]["n_pixels"] >= 6 * 14 * 14
    assert d["cpu_baseline"] is None               # the CPU timing leg is N = 1 only


def test_bench_library_reduce_world1():
    """BENCH_REDUCE=library: bench.py makes its own RCCL communicator (unique id from rank 0, ncclCommInitRank) and the film is summed by the
    library's pt_film_allreduce on the library's stream -- the call a Rust host would make.  World size 1 on the one GPU a box has."""
    d = _bench_line(["--gpus", "1", "--triangles", "20000", "--res", "256", "--spp", "16", "--steps", "1", "--warmup", "1", "--no-spp1024", "--no-cpu-baseline"],
                    {"BENCH_FORCE_REDUCE": "1", "BENCH_REDUCE": "library", "MASTER_PORT": str(29900 + os.getpid() % 90)})
    assert d["n_gpus"] == 1 and "pt_film_allreduce" in d["reduce"] and len(d["per_rank"]) == 1
    assert d["per_rank"][0]["tiles"] == d["tiles_total"] and d["per_rank"][0]["reduce_ms"] > 0


def test_cpp_driver_eight_ranks_deal_the_full_frame(tmp_path):
    """BASELINE config 3's deal at its real size: a 1024 x 1024 film is 65 x 65 = 4 225 tiles of 16 x 16 (sampler.rs:271-289); eight ranks -- threads of
    ONE process on this one GPU (a box allows six GPU processes, so bench.py's process-per-rank flow is rehearsed with four) -- take 529 or 528
    each, render concurrently, meet, and the films are summed.  Weights bit-equal to the single-context frame: every camera sample exactly once."""
    exe = os.path.join(ROOT, "pbrt-r3_amd", "csrc", "pbrt_gpu")
    text = open(os.path.join(ROOT, "tests", "scenes", "cornell.pbrt")).read().replace('"integer xresolution" [64] "integer yresolution" [64]', '"integer xresolution" [1024] "integer yresolution" [1024]')
    assert "[1024]" in text
    scene = str(tmp_path / "cornell_1024.pbrt")
    open(scene, "w").write(text)
    import shutil
    shutil.copy(os.path.join(ROOT, "tests", "scenes", "cornell_blocks.pbrt"), str(tmp_path / "cornell_blocks.pbrt"))          # the file it Includes
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one, eight = str(tmp_path / "one.xyzw"), str(tmp_path / "eight.xyzw")
    subprocess.check_call([exe, scene, "-o", str(tmp_path / "one.pfm"), "--xyzw", one, "--pixelsamples", "4", "--quiet"], env=env)
    r = subprocess.run([exe, scene, "-o", str(tmp_path / "eight.pfm"), "--xyzw", eight, "--pixelsamples", "4", "--devices", "0,0,0,0,0,0,0,0", "--stats"],
                       env=env, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    want, got = np.fromfile(one, np.float32).reshape(-1, 4), np.fromfile(eight, np.float32).reshape(-1, 4)
    assert want.shape[0] == 1024 * 1024 and np.array_equal(bits(got[:, 3]), bits(want[:, 3]))
    assert np.allclose(got, want, rtol=2e-6, atol=1e-7)
    tiles = [int(ln.split("tiles")[1].split()[0]) for ln in r.stderr.splitlines() if ln.strip().startswith("rank ")]
    assert len(tiles) == 8 and sum(tiles) == 65 * 65 and sorted(set(tiles)) == [528, 529]
