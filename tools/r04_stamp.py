#!/usr/bin/env python3
"""Merge a traffic entry (tools/pmc_traffic_workload.sh's traffic_entry.json) into profiles/traffic_r04.json, replacing the entry of the same kernel, so that a
bench.py run right behind the counter pass quotes it (bench.py only quotes entries stamped with the current pt_kernels.hip).  usage: tools/r04_stamp.py ENTRY.json [note]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(R, "profiles", "traffic_r04.json")
e = json.load(open(sys.argv[1]))
if len(sys.argv) > 2:
    e["note"] += "; " + sys.argv[2]
old = json.load(open(path)) if os.path.exists(path) else []
old = [x for x in old if x["kernel"] != e["kernel"]] + [e]
old.sort(key=lambda x: x["kernel"])
json.dump(old, open(path, "w"), indent=1)
print("stamped", e["kernel"], e["kernels_sha16"])
