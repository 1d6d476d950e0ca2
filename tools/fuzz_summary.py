#!/usr/bin/env python3
"""Condense the -s output of tests/test_gpu_fuzz.py (one "[fuzz N] ..." line per random scene) into a few lines for profiles/."""
import re, sys
from collections import Counter

lines = [l for l in open(sys.argv[1]) if l.startswith("[fuzz ")]
integ, split, leaf = Counter(), Counter(), Counter()
panics, worst, tris, spheres, inst, lights = 0, 0.0, 0, 0, 0, 0
for l in lines:
    m = re.match(r"\[fuzz (\d+)\] (\d+) triangles (\d+) spheres (\d+) instances (\d+) lights, integrator (\d+) split (\d+) leaf (\d+): (.*)", l)
    if not m:
        continue
    integ[int(m.group(6))] += 1; split[int(m.group(7))] += 1; leaf[int(m.group(8))] += 1
    tris += int(m.group(2)); spheres += int(m.group(3)) > 0; inst += int(m.group(4)) > 0; lights = max(lights, int(m.group(5)))
    if "panics" in m.group(9):
        panics += 1
    else:
        worst = max(worst, float(m.group(9).split()[-1]))
tail = [l.strip() for l in open(sys.argv[1]) if re.search(r"\d+ (passed|failed)", l)]
print("random scenes compared: %d (hits and occlusion of 60 k random rays, per-sample radiance of a 16 x 16 tile bit for bit, film, all counters)" % len(lines))
print("integrators (path, ao, directlighting, whitted): %s" % dict(sorted(integ.items())))
print("split methods (sah, hlbvh, middle, equal): %s   leaf sizes: %s" % (dict(sorted(split.items())), dict(sorted(leaf.items()))))
print("scenes with spheres: %d, with instances: %d; most lights in one scene: %d; triangles in all: %d" % (spheres, inst, lights, tris))
print("scenes where the reference panics (Halton dimensions) and both sides said so: %d" % panics)
print("largest image rel-L2 among the rest: %.2e" % worst)
for t in tail[-1:]:
    print("pytest: " + t)
