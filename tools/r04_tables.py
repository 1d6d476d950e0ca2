#!/usr/bin/env python3
"""Rewrite the numbers of BASELINE.md's round-4 table and of profiles/README.md's round-4 rows from the JSON lines under profiles/ (after tools/r04_collect.py)."""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(R)
def line(name):
    d = json.load(open("profiles/r04_%s.json" % name)); r = d["roofline"]
    return d, r
rows = [("**config 2 RT1M (default)**", "a_bench"), ("config 4 stand-in:", "b_bench_killeroo_class"), ("config 5 stand-in:", "b_bench_crown_class"), ("`--materials mixed` |", "b_bench_mixed"),
        ("beyond the Infinity Cache, dense:", "b_bench_16m"), ("beyond the Infinity Cache, sparse:", "b_bench_16m_sparse")]
s = open("BASELINE.md").read()
out = []
for ln in s.split("\n"):
    for label, name in rows:
        if ln.startswith("| " + label) and "profiles/r04_" in ln:
            d, r = line(name)
            cells = ln.split(" | ")
            fab = r["hbm_measured"] and ("%.2f" % r["hbm_measured"]["frac"]) or "--"
            cells[1:7] = ["**%.1f**" % d["value"], "%.2f" % (d["ms_per_step"] / 1e3), "%.2f" % r["l1_req"]["frac"], fab, "%.1f" % r["avg_launch_ms"], "%.0e" % d["parity"]["rel_l2"]]
            ln = " | ".join(cells)
    out.append(ln)
s = "\n".join(out)
sec = json.load(open("profiles/r04_b_summary.json"))
s = re.sub(r"`directlighting` / `whitted` \(RT1M, depth 5\): [\d.]+ / [\d.]+ Mrays/s \(683 / 728 in round 3\); `ao` [\d.]+; RT1M under Halton [\d.]+, lit by a sphere [\d.]+, textured [\d.]+ \(64 spp;",
           "`directlighting` / `whitted` (RT1M, depth 5): %.1f / %.1f Mrays/s (683 / 728 in round 3); `ao` %.1f; RT1M under Halton %.1f, lit by a sphere %.1f, textured %.1f (64 spp;" % (
               sec["integrator_directlighting"]["value"], sec["integrator_whitted"]["value"], sec["integrator_ao"]["value"], sec["sampler_halton_spp_64"]["value"], sec["light_sphere_spp_64"]["value"],
               sec["materials_textured_spp_64"]["value"]), s)
k = json.load(open("profiles/r04_b_bench_killeroo_class.json"))["config"]
s = re.sub(r"Upload of the killeroo-class stand-in: [\d.]+ \+ [\d.]+ ms", "Upload of the killeroo-class stand-in: %.1f + %.1f ms" % (k["bvh_build_ms"], k["upload_ms"]), s)
open("BASELINE.md", "w").write(s)
p = open("profiles/README.md").read()
def sp(v): return ("%d" % round(v)) if v < 1000 else ("%d %03d" % (round(v) // 1000, round(v) % 1000))
V = lambda n: json.load(open("profiles/r04_%s.json" % n))["value"]
p = re.sub(r"crown-class \*\*[\d ]+\*\*, 16 M dense [\d ]+, 16 M sparse \*\*[\d ]+\*\*, killeroo-class \*\*[\d ]+\*\*, mixed \*\*[\d ]+\*\*, `directlighting` \*\*[\d ]+\*\*, `whitted` \*\*[\d ]+\*\*, `ao` [\d ]+, Halton [\d ]+, sphere light [\d ]+, textured [\d ]+",
           "crown-class **%s**, 16 M dense %s, 16 M sparse **%s**, killeroo-class **%s**, mixed **%s**, `directlighting` **%s**, `whitted` **%s**, `ao` %s, Halton %s, sphere light %s, textured %s" % (
               sp(V("b_bench_crown_class")), sp(V("b_bench_16m")), sp(V("b_bench_16m_sparse")), sp(V("b_bench_killeroo_class")), sp(V("b_bench_mixed")), sp(sec["integrator_directlighting"]["value"]),
               sp(sec["integrator_whitted"]["value"]), sp(sec["integrator_ao"]["value"]), sp(sec["sampler_halton_spp_64"]["value"]), sp(sec["light_sphere_spp_64"]["value"]), sp(sec["materials_textured_spp_64"]["value"])), p)
a, _ = line("a_bench"); u, ur = line("a_bench_under_rocprof")
import csv
ks = [x for x in csv.DictReader(open("profiles/r04_a_kernel_stats.csv")) if x["Name"] == "k_trace"][0]
p = re.sub(r"\*\*1 [\d.]+ Mrays/s\*\* \(1 106\.7 on an earlier box\), 1\.54 s per frame; `k_trace` [\d.]+ ms per launch \([^)]*\) vs [\d.]+ ms \(`AverageNs`\)",
           "**%s Mrays/s** (1 106.7 on an earlier box), 1.54 s per frame; `k_trace` %.2f ms per launch (HIP events, the run under the profiler) vs %.2f ms (`AverageNs`)" % (
               ("1 %05.1f" % (a["value"] - 1000)), ur["avg_launch_ms"], float(ks["AverageNs"]) / 1e6), p)
open("profiles/README.md", "w").write(p)
print("tables rewritten")
