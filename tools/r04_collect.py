#!/usr/bin/env python3
"""Copy the outputs of tools/r04_final.sh (gpurun_out/r04_final_{a,b,c,d}) into profiles/ under their round-4 names and rebuild profiles/traffic_r04.json."""
import glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(R)
a, b = "gpurun_out/r04_final_a/", "gpurun_out/r04_final_b/"
entries = []
if os.path.exists(a + "traffic_entry_default.json"):
    entries.append(json.load(open(a + "traffic_entry_default.json")))
    for src, dst in [("bench.json", "r04_a_bench.json"), ("bench_under_rocprof.json", "r04_a_bench_under_rocprof.json"), ("kernel_stats.csv", "r04_a_kernel_stats.csv"),
                     ("per_bounce.txt", "r04_a_per_bounce.txt"), ("pmc_fetch_write_default.txt", "r04_a_pmc_fetch_write_default.txt")]:
        shutil.copy(a + src, "profiles/" + dst)
if os.path.exists(b + "traffic_entry_16m_sparse.json"):
    e = json.load(open(b + "traffic_entry_16m_sparse.json"))
    e["note"] += "; PBRTGPU_TRACE_FAR=1 for the whole frame (what the library's trial picks for this scene)"
    entries.append(e)
    shutil.copy(b + "pmc_fetch_write_16m_sparse.txt", "profiles/r04_b_pmc_fetch_write_16m_sparse.txt")
    sec = {}
    for f in sorted(glob.glob(b + "bench_*.json")):
        shutil.copy(f, "profiles/r04_b_" + os.path.basename(f))
        d = json.load(open(f))
        sec[os.path.basename(f)[6:-5]] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"]}
    json.dump(sec, open("profiles/r04_b_summary.json", "w"), indent=1)
    print({k: v["value"] for k, v in sec.items()})
if entries:
    json.dump(entries, open("profiles/traffic_r04.json", "w"), indent=1)
    for e in entries:
        print(e["kernel"], e["kernels_sha16"], "fabric %.2f TB/s" % (e["fabric_bytes_per_launch"] / (e["avg_launch_ms_under_pmc"] / 1e3) / 1e12))
for src, dst in [("gpurun_out/r04_final_c/pytest_gpu.txt", "profiles/r04_c_pytest_gpu.txt"),
                 ("gpurun_out/r04_final_d/bench_rehearse_4ranks_one_gpu_gloo.json", "profiles/r04_d_bench_rehearse_4ranks_one_gpu_gloo.json"),
                 ("gpurun_out/r04_final_d/pbrt_gpu_8ranks_one_gpu.txt", "profiles/r04_d_pbrt_gpu_8ranks_one_gpu_1024x1024.txt")]:
    if os.path.exists(src):
        shutil.copy(src, dst)
