#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on its headline config.

Workload "RT1M" (BASELINE.md config 2): 1M random matte triangles in a lit box,
1024x1024 film, Sobol' 256 spp, path depth 8.  One STEP = one full render of that frame
(every 16x16 tile of the 1026x1026 sample bounds, all 256 spp).  With N ranks the frame's
tiles are dealt round-robin to the ranks (each GPU holds the whole scene), every rank
renders its tiles, and the per-rank XYZW films are summed with one RCCL all-reduce over
xGMI: total work is fixed, so scaling is "strong".

metric  Mrays/s  = rays counted exactly as the reference's "Regular ray intersection tests"
                   + "Shadow ray intersection tests" (src/core/scene/scene.rs:11-12), summed
                   over ranks, / wall time of the K timed steps (scene upload + BVH build
                   excluded; inputs are resident in HBM when timing starts).
roofline         = the traversal kernel (k_trace).  Primary bound: the request rate of the CU's
                   vector L1 (lane requests per launch, from the library's exact counters, / mean
                   launch duration from HIP events recorded inside the library on the stream the
                   kernel runs on, vs one request per clock per CU) -- the bound that holds while
                   the scene is cache-resident.  Beside it, `hbm`: bytes that left L2 per launch
                   (FETCH_SIZE x 2 + WRITE_SIZE from rocprofv3 --pmc passes on exactly this
                   workload, profiles/traffic_*.json) / the same duration vs 8 TB/s, and
                   SURVEY.md section 8d's algorithmic bytes (48 B/closest ray, 36 B/any-hit ray,
                   128 B/node, 48 B/triangle test) as `algorithmic_gbps` (cache-served: may
                   exceed the HBM peak).  `bound` names whichever of the two fractions is larger.
cpu_baseline     = the CPU oracle (C++ restatement of the reference; the Rust binary cannot
                   be built in this image) rendering a bounded sample of the same workload on
                   the host cores of rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def effective_cores():
    """Host threads this process may actually use: affinity mask and cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    cap = int(os.environ.get("BENCH_CPU_THREADS", "0"))
    return min(n, cap) if cap > 0 else n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--triangles", type=int, default=1000000)
    ap.add_argument("--tri-size", type=float, default=0.005, help="half-extent of a filler triangle's vertex box (0.005 = the headline).  Opacity goes with "
                    "triangles x size^2: --triangles 16000000 --tri-size 0.00125 keeps RT1M's ray depth over a 1.1 GB tree (the beyond-the-Infinity-Cache run)")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--materials", default="matte", choices=["matte", "mixed", "textured"],
                    help="matte = BASELINE config 2 (the headline); mixed = killeroo-class stand-in for config 4 (secondary number)")
    ap.add_argument("--sampler", default="sobol", choices=["sobol", "halton"], help="sobol = the headline; halton = the reference's default sampler (secondary number)")
    ap.add_argument("--light", default="quad", choices=["quad", "sphere"], help="quad = the headline; sphere = an analytic sphere light instead (secondary number: k_trace_sph / k_shade_*_sph)")
    ap.add_argument("--integrator", default="path", choices=["path", "ao", "directlighting", "whitted"],
                    help="path = the headline; ao = Integrator \"ao\" with 64 occlusion rays per camera sample; directlighting / whitted = the recursive integrators (secondary numbers)")
    ap.add_argument("--instances", type=int, default=0, help="secondary number: the filler triangles as ONE object instanced this many times (k_trace_inst, k_shade_general_inst*)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spp1024", action="store_true", help="skip the secondary 1024-spp frame (N=1 only)")
    ap.add_argument("--cpu-tiles", type=int, default=0, help="tiles in the CPU sample (0 = auto, about 15 s)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Called without a launcher: start the N ranks ourselves (fresh children of a process that has not touched the GPU)
        # and let rank 0 of that job print the line.  n_gpus in the line is always the number of ranks that ran.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    # The contract is ONE JSON line on stdout.  Native libraries (RCCL's version banner and NCCL WARN lines, gloo's
    # connection notes) write to file descriptor 1 directly, so fd 1 is pointed at stderr for the whole run and the
    # JSON line goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = world          # ranks that actually run (one GPU each), whatever --gpus says
    if args.gpus != world and rank == 0:
        print("[bench] --gpus %d but WORLD_SIZE=%d: reporting n_gpus=%d" % (args.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    # BENCH_REHEARSE=gloo: run the N>1 flow (partition, reduce, barrier, max-over-ranks timing) with every rank
    # on GPU 0 and a host-staged gloo reduce -- a 1-GPU box cannot host two RCCL ranks.  Never a judged number.
    rehearse = os.environ.get("BENCH_REHEARSE") == "gloo"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_REDUCE") == "1"     # the latter: rehearse the N>1 code path on one GPU
    if use_dist:
        os.environ["NCCL_DEBUG"] = os.environ.get("BENCH_NCCL_DEBUG", "WARN")   # keep RCCL's banner off stdout: one JSON line only
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def log(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.time() - t_prog, msg), file=sys.stderr, flush=True)

    t_prog = time.time()
    pkg = importlib.import_module("pbrt-r3_amd")
    t0 = time.time()
    sd = pkg.scenes.rt1m(args.triangles, res=args.res, spp=args.spp, max_depth=args.max_depth, s=args.tri_size, materials=args.materials, sampler=args.sampler, light=args.light, instances=args.instances)
    if args.integrator == "ao":
        sd.desc.integrator, sd.desc.ao_samples, sd.desc.ao_cos_sample = pkg.capi.PT_INTEGRATOR_AO, 64, 1
    elif args.integrator in ("directlighting", "whitted"):
        sd.desc.integrator = pkg.capi.PT_INTEGRATOR_DIRECTLIGHTING if args.integrator == "directlighting" else pkg.capi.PT_INTEGRATOR_WHITTED
        sd.desc.direct_strategy, sd.desc.max_depth = pkg.capi.PT_DIRECT_ALL, 5
    t_scene = time.time() - t0
    ctx = pkg.Context(local_rank)
    info = ctx.upload(sd)
    first_build_ms, first_upload_ms = info.bvh_build_ms, info.upload_ms
    info = ctx.upload(sd)            # the same scene again: the context's staging buffers exist now (what a rebuild of a changed scene costs)
    rebuild_ms, reupload_ms = info.bvh_build_ms, info.upload_ms
    info = ctx.upload(sd)            # ... twice, the faster one reported (host threads: the first rebuild still grows a few buffers)
    rebuild_ms, reupload_ms = min(rebuild_ms, info.bvh_build_ms), min(reupload_ms, info.upload_ms)
    tiles = pkg.scenes.all_tiles(info)
    my_tiles = tiles[rank::world]

    film_ptr, film_n = None, 0

    t_render, t_reduce = [0.0], [0.0]      # this rank's host wall time inside render / inside the film reduce, over the timed steps

    # Which reduce runs (stated in the line as `reduce`): BENCH_REDUCE=torch (default: torch.distributed all_reduce = RCCL, over a
    # torch-owned staging copy), =in_place (RCCL on the library's own allocation), =library (the library's pt_film_allreduce on the
    # library's stream, over an RCCL communicator this process creates from a unique id rank 0 broadcasts -- what a Rust host would call).
    reduce_mode = os.environ.get("BENCH_REDUCE", "in_place" if os.environ.get("BENCH_REDUCE_IN_PLACE") == "1" else "torch")
    lib_comm, rccl = None, None
    if use_dist and not rehearse and reduce_mode == "library":
        import ctypes as C

        class NcclUniqueId(C.Structure):
            _fields_ = [("internal", C.c_byte * 128)]
        rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)        # the copy torch has already mapped
        rccl.ncclGetUniqueId.argtypes = [C.POINTER(NcclUniqueId)]
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
        uid = NcclUniqueId()
        if rank == 0:
            assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
        box = [bytes(uid.internal) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        C.memmove(C.byref(uid), box[0], 128)
        lib_comm = C.c_void_p()
        rc = rccl.ncclCommInitRank(C.byref(lib_comm), world, uid, rank)
        if rc != 0:
            raise SystemExit("ncclCommInitRank failed: %d" % rc)
    reduce_desc = (None if not use_dist else
                   "gloo all_reduce of a host copy (BENCH_REHEARSE: several ranks on one GPU, never a judged number)" if rehearse else
                   {"torch": "torch.distributed all_reduce (RCCL) over a torch-owned staging copy of the film",
                    "in_place": "torch.distributed all_reduce (RCCL) on the library's own film allocation",
                    "library": "pt_film_allreduce (the library's own RCCL call on its stream; communicator from ncclCommInitRank)"}[reduce_mode])

    def step():
        ctx.film_clear()
        ta = time.time()
        ctx.render(my_tiles)               # returns with the rank's film complete on the device
        tb = time.time()
        t_render[0] += tb - ta
        if use_dist:
            # the one collective of the path: sum the per-rank XYZW films (disjoint tiles)
            if rehearse:
                ptr, n = ctx.film_device_xyzw()
                t = pkg.dist.wrap_device_floats(ptr, n, local_rank)
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                t.copy_(h)
                torch.cuda.synchronize()
                ctx.film_commit_xyzw()
            elif lib_comm is not None:
                ctx.film_allreduce(lib_comm.value)
            else:
                pkg.dist.reduce_film(ctx, local_rank, staged=reduce_mode != "in_place")
            t_reduce[0] += time.time() - tb      # includes the wait for the slowest rank's render

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def check_reduced_film():
        # every camera sample of every rank's tiles must be in the reduced film exactly once (box filter: weight 1 each, so every
        # pixel's weight is spp up to the edge-split samples; a rank's tiles lost in the reduce would leave 16x16 holes of weight 0)
        x = ctx.film_xyzw()
        want = float(info.spp) * (info.cropped_bounds[2] - info.cropped_bounds[0]) * (info.cropped_bounds[3] - info.cropped_bounds[1])
        got = float(x[..., 3].astype(np.float64).sum())
        holes = int((x[..., 3] < 0.5 * info.spp).sum())
        # (samples that land exactly on a pixel edge split their weight, and on the film's border part of it falls outside: a handful of units)
        ok = abs(got - want) <= max(1e-6 * want, 64.0) and holes == 0
        log("reduced film: sum of weights %.1f, expected %.1f, pixels under half weight %d (%s)" % (got, want, holes, "ok" if ok else "MISMATCH"))
        return {"ok": bool(ok), "weight_sum": got, "weight_sum_expected": want, "pixels_under_half_weight": holes}

    log("scene %d tris uploaded (gen %.1fs, bvh %.0f ms), %d tiles for this rank" % (sd.desc.n_triangles, t_scene, info.bvh_build_ms, len(my_tiles)))
    for i in range(args.warmup):
        step()
        log("warmup step %d done" % i)
    fence()
    ctx.reset_counters()
    t_render[0] = t_reduce[0] = 0.0
    t_start = time.time()
    for i in range(args.steps):
        step()
        log("timed step %d done" % i)
    fence()
    elapsed = time.time() - t_start
    cnt = ctx.counters()

    film_check = None
    if use_dist and world > 1 and os.environ.get("BENCH_CHECK_FILM") != "0":      # a scaling line vouches for its own film (every rank holds the sum)
        film_check = check_reduced_film()
        flag = torch.tensor([1.0 if film_check["ok"] else 0.0], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        film_check["ok_on_every_rank"] = bool(float(flag[0]) > 0.5)
    stats = torch.tensor([elapsed, cnt["regular_rays"], cnt["shadow_rays"], cnt["nodes_visited"], cnt["tris_tested"],
                          cnt["path_vertices"], cnt["trace_ms"], cnt["trace_launches"], cnt["camera_rays"], cnt["shade_ms"]],
                         dtype=torch.float64, device="cpu" if rehearse else "cuda")
    per_rank = None
    if use_dist:
        # per-rank breakdown, so that a scaling run can be read from its JSON line: who rendered how long, who waited in the reduce
        mine = torch.tensor([t_render[0] * 1e3 / max(1, args.steps), t_reduce[0] * 1e3 / max(1, args.steps), float(len(my_tiles)),
                             cnt["regular_rays"] + cnt["shadow_rays"], cnt["trace_ms"] / max(1, args.steps), cnt["shade_ms"] / max(1, args.steps)],
                            dtype=torch.float64, device="cpu" if rehearse else "cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r, "render_ms": round(float(v[0]), 3), "reduce_ms": round(float(v[1]), 3), "tiles": int(v[2]),
                     "rays_per_step": int(float(v[3]) / max(1, args.steps)), "trace_ms": round(float(v[4]), 3), "shade_ms": round(float(v[5]), 3)}
                    for r, v in enumerate(every)]
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        tot = sm.tolist()
    else:
        tot = stats.tolist()
    rays = tot[1] + tot[2]
    value = rays / elapsed / 1e6

    out = None
    if rank == 0:
        # roofline of the dominant kernel (k_trace), from rank 0's own launches.  k_trace is bound by the REQUEST RATE of the CU's
        # vector L1 (TCP): with one ray per lane every lane's node / triangle record is a cache line of its own, and the L1 takes one
        # 16-byte lane request per clock (tools/ubench/node_fetch.hip: 77 G eight-load node visits/s = 614 G requests/s = 256 CUs x
        # 2.4 GHz with nothing else running).  achieved = lane requests per launch / launch time.
        per_visit = 7.0                           # k_trace / k_trace_sph_dist (path and ao alike): six plane rows + the child references per node visit
        n_rays = cnt["regular_rays"] + cnt["shadow_rays"]
        l1_visits = cnt["nodes_visited"] - cnt.get("nodes_from_lds", 0)       # visits to the top levels of the tree read the kernel's LDS copy, not L1
        reqs = per_visit * l1_visits + 3.0 * cnt["tris_tested"] + 5.0 * n_rays      # per ray: id + 2 x 16 B in, 2 result stores out
        alg_bytes = 48.0 * cnt["regular_rays"] + 36.0 * cnt["shadow_rays"] + 128.0 * cnt["nodes_visited"] + 48.0 * cnt["tris_tested"]
        launches = max(1, cnt["trace_launches"])
        avg_launch_s = cnt["trace_ms"] / 1e3 / launches
        props = torch.cuda.get_device_properties(local_rank)
        n_cu = int(props.multi_processor_count)
        peak = n_cu * 2.4                          # G requests/s: one per clock per CU at the 2.4 GHz nominal clock (MI355X_MICROARCH.md)
        achieved = (reqs / launches) / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        scene_bytes = 128.0 * info.n_nodes + 48.0 * sd.desc.n_triangles
        compulsory = (48.0 * cnt["regular_rays"] + 36.0 * cnt["shadow_rays"]) / launches + scene_bytes
        traffic, traffic_note = None, "no PMC measurement for this workload"
        key = {"triangles": args.triangles, "res": args.res, "spp": args.spp, "max_depth": args.max_depth, "materials": args.materials,
               "sampler": args.sampler, "light": args.light, "integrator": args.integrator}
        if args.tri_size != 0.005:
            key["tri_size"] = args.tri_size
        if args.instances > 0:
            key["instances"] = args.instances
        import glob
        import hashlib
        # a PMC measurement describes the kernels it was taken on: entries carry sha256(pt_kernels.hip)[:16], and one taken on other
        # kernel source is not quoted (traffic = null) until tools/pmc_traffic_workload.sh has been re-run
        ksha = hashlib.sha256(open(os.path.join(ROOT, "pbrt-r3_amd", "csrc", "pt_kernels.hip"), "rb").read()).hexdigest()[:16]
        stale = None
        for tp in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")), reverse=True):     # newest round first
            try:
                tj = json.load(open(tp))
            except Exception:
                continue
            for ent in (tj if isinstance(tj, list) else [tj]):
                # measured on exactly this workload (passes of the same size) and this kernel source, else null
                if traffic is None and isinstance(ent, dict) and ent.get("workload") == key and ent.get("fabric_bytes_per_launch"):
                    if ent.get("kernels_sha16") == ksha:
                        traffic, traffic_note = ent["fabric_bytes_per_launch"], ent.get("note", "") + " (%s)" % os.path.basename(tp)
                    elif stale is None:
                        stale = os.path.basename(tp)
        if traffic is None and stale:
            traffic_note = "the PMC entry for this workload in %s was measured on other kernel source (pt_kernels.hip is now %s): re-run tools/pmc_traffic_workload.sh" % (stale, ksha)
        hbm_frac = (traffic / avg_launch_s / 8e12) if (traffic and avg_launch_s > 0) else None
        l1_frac = achieved / peak
        alg_gbps = (alg_bytes / launches) / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        l1_blk = {"bound": "l1_req", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "Greq/s", "frac": round(l1_frac, 4)}
        hbm_blk = ({"bound": "hbm", "achieved": round(traffic / avg_launch_s / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(hbm_frac, 4),
                    "what": "bytes that left L2 (FETCH_SIZE x 2 + WRITE_SIZE, Infinity-Cache hits included) per launch / launch time"}
                   if hbm_frac is not None else None)
        # Top level: always the contract's form -- bound "hbm", achieved = SURVEY.md section 8d's ALGORITHMIC bytes per launch / the launch
        # time measured here, against 8 TB/s.  While the scene is cache-resident those bytes are served by L2 and the Infinity Cache, so the
        # fraction can exceed 1; `traffic` (and `hbm_measured`) is what the counters saw leave L2, `l1_req` the bound that holds then.
        roofline = {"bound": "hbm", "achieved": round(alg_gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg_gbps / 8000.0, 4), "traffic": traffic}
        roofline.update({
                    "kernel": ("k_trace_inst" if args.instances > 0 else
                               "k_trace_sph_dist" if args.light == "sphere" else
                               "k_trace_far" if os.environ.get("PBRTGPU_TRACE_FAR") == "1" else
                               "k_trace" if (os.environ.get("PBRTGPU_TRACE_FAR") == "0" or scene_bytes <= 256 * 2 ** 20) else
                               "k_trace or k_trace_far (scenes beyond the Infinity Cache: the library's timed trial on the first incoherent bounce picks one)"),
                    "l1_req": l1_blk, "hbm_measured": hbm_blk, "kernels_sha16": ksha,
                    "requests_per_launch": round(reqs / launches, 1), "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": int(launches),
                    "algorithmic_bytes_per_launch": round(alg_bytes / launches, 1),
                    "nodes_per_ray": round(cnt["nodes_visited"] / max(1.0, n_rays), 2), "tris_per_ray": round(cnt["tris_tested"] / max(1.0, n_rays), 2),
                    "node_visits_from_lds": round(cnt.get("nodes_from_lds", 0) / max(1.0, cnt["nodes_visited"]), 4),
                    "trace_share_of_render": round(cnt["trace_ms"] / max(1e-9, cnt["render_ms"]), 3),
                    "shade_share_of_render": round(cnt["shade_ms"] / max(1e-9, cnt["render_ms"]), 3),
                    "scene_bytes": scene_bytes, "infinity_cache_bytes": 256 * 2 ** 20,
                    "hbm_compulsory_bytes_per_launch": round(compulsory, 1),
                    "hbm_compulsory_frac_of_8TBps": round(compulsory / avg_launch_s / 8e12, 5) if avg_launch_s > 0 else 0.0,
                    "fabric_frac_of_8TBps": round(hbm_frac, 4) if hbm_frac is not None else None,
                    "note": "top level = algorithmic bytes (48 B / closest-hit ray, 36 B / any-hit ray, 128 B / node visit, 48 B / triangle test; cache-served "
                            "while the scene fits L2 + Infinity Cache) / k_trace's mean launch time / 8 TB/s.  l1_req = lane requests to the CU's vector L1 "
                            "(one 16-byte request per clock per CU): %g per node visit that is not served from the LDS copy of the top of the tree, 3 per "
                            "triangle test, 5 per ray.  traffic / hbm_measured = FETCH_SIZE x 2 + WRITE_SIZE bytes leaving L2 (Infinity Cache hits included) "
                            "per launch: %s" % (per_visit, traffic_note)})
        cpu, parity, spp1024 = None, None, None

        def film_parity(osc_, oxyzw_, sample_, what):
            # parity of the benchmarked frame: the GPU film of the last timed step (at N > 1: the REDUCED film) against the oracle on the
            # sampled tiles (linear RGB after the division by the filter weight; north_star tolerance: 1e-3 relative L2)
            g_rgb = ctx.film_rgb().astype(np.float64)
            o_rgb = osc_.resolve_rgb(oxyzw_).astype(np.float64)
            cb = list(info.cropped_bounds)
            mask = np.zeros(g_rgb.shape[:2], bool)
            for (x0, y0, x1, y1) in sample_:
                # interior pixels of the tile only: a border pixel also receives the edge-split samples of the neighbouring tiles
                # (normalised footprints, quirk Q1), which the full GPU frame has and the oracle's partial film has not
                mask[max(y0 + 1, cb[1]) - cb[1]: max(min(y1 - 1, cb[3]) - cb[1], 0), max(x0 + 1, cb[0]) - cb[0]: max(min(x1 - 1, cb[2]) - cb[0], 0)] = True
            gd, od = g_rgb[mask], o_rgb[mask]
            den = np.maximum(np.abs(od), 1e-6)
            out_ = {"rel_l2": float(np.sqrt(((gd - od) ** 2).sum()) / max(np.sqrt((od ** 2).sum()), 1e-30)),
                    "max_rel": float((np.abs(gd - od) / den).max()) if gd.size else 0.0,
                    "pixels_over_1pct": int(((np.abs(gd - od) / den).max(axis=-1) > 0.01).sum()) if gd.size else 0,
                    "n_pixels": int(mask.sum()), "tolerance_rel_l2": 1e-3, "reference": what}
            log("parity: rel-L2 %.3e over %d pixels" % (out_["rel_l2"], out_["n_pixels"]))
            return out_

        if not args.no_cpu_baseline and world > 1:
            # N > 1: no CPU timing leg (the other ranks would wait for it), but the line still vouches for its film -- the oracle renders four
            # of RANK 0's tiles and four of the LAST rank's, and the reduced film must match on both
            import oracle_lib
            osc = oracle_lib.load().scene(sd)
            mine0, last = tiles[0::world], tiles[world - 1::world]
            sample = [mine0[(len(mine0) * k) // 5] for k in (1, 2, 3, 4)] + [last[(len(last) * k) // 5] for k in (1, 2, 3, 4)]
            oxyzw, _, secs = osc.render(sample, threads=min(effective_cores(), 16), want_image=True)
            parity = film_parity(osc, oxyzw, sample, "oracle (CPU restatement) on 4 tiles of rank 0 and 4 of rank %d against the REDUCED film, interior pixels, all %d spp (%.1f s)"
                                 % (world - 1, info.spp, secs))
            osc.close()
        if not args.no_cpu_baseline and world == 1:      # the CPU timing leg runs at N=1 only (rank 0 would keep the other ranks waiting)
            import oracle_lib
            osc = oracle_lib.load().scene(sd)
            cores = min(effective_cores(), 16)      # a 1-GPU box grants 16 host CPUs; BENCH_CPU_THREADS overrides the probe
            # The timed build is compiled HERE, for this host's cores (-O3 -march=native, still -ffp-contract=off); the portable build that
            # travels with the repo (-O2) stays the checker, and the native one must reproduce its film bit for bit before it is timed.
            native, native_note = None, None
            try:
                native = oracle_lib.load_native()
            except Exception as e:                   # no compiler on this host: time the portable build and say so
                native_note = "native build unavailable (%s): the portable -O2 build was timed" % (str(e).splitlines()[-1][:120] if str(e) else type(e).__name__)
                log("cpu baseline: " + native_note)
            tsc = native.scene(sd) if native is not None else osc
            ntile = args.cpu_tiles
            if ntile <= 0:
                # calibrate on 2 tiles, then size the sample for ~15 s on all cores
                probe = tiles[len(tiles) // 2: len(tiles) // 2 + 2]
                log("cpu baseline: oracle scene built, calibrating on 2 tiles")
                _, c0, s0 = tsc.render(probe, threads=min(2, cores), want_image=False)
                log("cpu baseline: 2 tiles took %.1f s" % s0)
                per_tile = s0 / 2 * min(2, cores)
                ntile = int(max(cores, min(len(tiles), 15.0 * cores / max(per_tile, 1e-3))))
            stride = max(1, len(tiles) // ntile)
            sample = tiles[::stride][:ntile]
            log("cpu baseline: rendering %d tiles on %d threads (%s build)" % (len(sample), cores, "native" if native is not None else "portable"))
            oxyzw, ccnt, secs = tsc.render(sample, threads=cores, want_image=True)
            log("cpu baseline: done in %.1f s" % secs)
            bit_equal = None
            if native is not None:
                sub = sample[:: max(1, len(sample) // 8)][:8]          # a bounded cross-check: eight of the sampled tiles through the portable build
                pxyzw, pcnt, psecs = osc.render(sub, threads=cores, want_image=True)
                nxyzw, ncnt, _ = tsc.render(sub, threads=cores, want_image=True)
                bit_equal = bool(np.array_equal(pxyzw.view(np.uint32), nxyzw.view(np.uint32)) and all(pcnt[k] == ncnt[k] for k in ("regular_rays", "shadow_rays", "nodes_visited", "tris_tested", "path_vertices")))
                log("cpu baseline: native film %s the portable build's on %d tiles" % ("bit-equal to" if bit_equal else "DIFFERS from", len(sub)))
                if not bit_equal:
                    raise SystemExit("the native oracle build does not reproduce the portable build: refusing to time it")
            parity = film_parity(tsc, oxyzw, sample, "oracle (CPU restatement), interior pixels of the %d sampled tiles at all %d spp" % (len(sample), info.spp))
            crays = ccnt["regular_rays"] + ccnt["shadow_rays"]
            cpu = {"value": round(crays / secs / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "mrays_s_per_thread": round(crays / secs / 1e6 / max(1, cores), 4),
                   "sample": "%d of %d 16x16 tiles (every %dth), all %d spp, %.1f s" % (len(sample), len(tiles), stride, info.spp, secs),
                   "build": (oracle_lib.NATIVE_FLAGS + " (compiled on this host at bench time)") if native is not None else oracle_lib.PORTABLE_FLAGS,
                   "bit_equal_to_portable_build": bit_equal,
                   "note": "C++ restatement of the reference (oracle/), not the Rust binary: tile-parallel over %d host threads like the reference's rayon par_iter; "
                           "a checker written for fidelity (per-call heap stacks, full interactions per candidate hit, as the reference has them), not a tuned renderer%s"
                           % (cores, ("; " + native_note) if native_note else "")}
            if native is not None:
                tsc.close()
            osc.close()
        if world == 1 and not args.no_spp1024 and args.spp != 1024:
            # north_star's target sentence quotes 1024 spp for the same scene (BASELINE config 2 says 256; Mrays/s is spp-independent,
            # wall-clock scales): one extra frame at 1024 spp, wall-clock reported beside the headline
            sd4 = pkg.scenes.rt1m(args.triangles, res=args.res, spp=1024, max_depth=args.max_depth, s=args.tri_size, materials=args.materials, sampler=args.sampler, light=args.light, instances=args.instances)
            sd4.desc.integrator, sd4.desc.ao_samples, sd4.desc.ao_cos_sample = sd.desc.integrator, sd.desc.ao_samples, sd.desc.ao_cos_sample
            sd4.desc.direct_strategy = sd.desc.direct_strategy
            if args.integrator in ("directlighting", "whitted"):
                sd4.desc.max_depth = 5
            i4 = ctx.upload(sd4)
            ctx.film_clear(); ctx.reset_counters()
            torch.cuda.synchronize()
            t4 = time.time()
            ctx.render()
            torch.cuda.synchronize()
            s4 = time.time() - t4
            c4 = ctx.counters()
            spp1024 = {"spp": int(i4.spp), "wall_s": round(s4, 3), "mrays_s": round((c4["regular_rays"] + c4["shadow_rays"]) / s4 / 1e6, 1),
                       "rays": int(c4["regular_rays"] + c4["shadow_rays"])}
            log("1024 spp: %.2f s wall, %.1f Mrays/s" % (s4, spp1024["mrays_s"]))
        out = {
            "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RT1M: %d random %s triangles%s%s, %dx%d, %s %d spp, path maxdepth %d, BVH sah/4, spatial lights"
                                   % (sd.desc.n_triangles, "matte" if args.materials == "matte" else ("mixed-material (matte/plastic/metal/glass/mirror/substrate)" + (", texture-driven colours and bump maps" if args.materials == "textured" else "")),
                                      " + a sphere area light" if args.light == "sphere" else "", "" if args.tri_size == 0.005 else " of half-extent %g" % args.tri_size, args.res, args.res, "Sobol" if args.sampler == "sobol" else "Halton", info.spp, args.max_depth),
                       "partition": "16x16 film tiles round-robin over %d rank(s), RCCL all-reduce of the XYZW film" % world,
                       "rays_per_step": int(rays / max(1, args.steps)), "camera_samples_per_step": int(tot[8] / max(1, args.steps)),
                       "bvh_build_ms": round(rebuild_ms, 1), "upload_ms": round(reupload_ms, 1),
                       "bvh_build_first_upload_ms": round(first_build_ms, 1), "first_upload_ms": round(first_upload_ms, 1),
                       "scene_gen_s": round(t_scene, 2), "workload_key": key},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "spp1024": spp1024,
        }
        if per_rank is not None:
            # reduce_ms of a rank = its wait for the slowest rank + the collective itself; min over ranks ~ the collective alone
            out["per_rank"] = per_rank
            out["reduce"] = reduce_desc
            out["reduced_film_ok"] = (film_check["ok"] and film_check["ok_on_every_rank"]) if film_check else None
            out["reduced_film"] = film_check
            out["tiles_total"] = len(tiles)
        if args.integrator in ("directlighting", "whitted"):
            out["config"]["workload"] = out["config"]["workload"].replace("path maxdepth %d" % args.max_depth, args.integrator + " maxdepth 5")
        if args.integrator == "ao":
            out["config"]["workload"] = out["config"]["workload"].replace("path maxdepth %d" % args.max_depth, "ao nsamples 64 cossample")
            if roofline:
                roofline["kernel"] = "k_trace (camera rays, then the occlusion rays as shadow work items)"
        if args.instances > 0:
            out["config"]["workload"] = out["config"]["workload"].replace("RT1M: ", "RT1M instanced: the filler triangles as one object, %d instances (%d triangles on screen); " % (args.instances, 12 + args.instances * (sd.desc.n_triangles - 12)))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
