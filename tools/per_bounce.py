#!/usr/bin/env python3
"""Per-launch durations of the wavefront kernels of the LAST frame in a rocprofv3 kernel_trace.csv (one k_trace launch per bounce)."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0] for r in rows]
gens = [i for i, n in enumerate(names) if n == "k_gen"]
films = [i for i, n in enumerate(names) if n == "k_film"]
a, b = gens[-1], films[-1]
t0 = int(rows[a]["Start_Timestamp"])
frame = (int(rows[b]["End_Timestamp"]) - t0) / 1e6
print("last frame: %.2f ms from k_gen to the end of k_film, %d launches" % (frame, b - a + 1))
bounce = -1
acc = {}
for i in range(a, b + 1):
    n = names[i]
    d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e6
    if n == "k_trace":
        if acc:
            print("bounce %d: " % bounce + "  ".join("%s %.3f" % kv for kv in acc.items()) + "   | sum %.3f ms (%.2f %% of the frame)" % (sum(acc.values()), 100 * sum(acc.values()) / frame))
        bounce += 1
        acc = {}
    if bounce >= 0:
        short = n if len(n) < 28 else (n[:25] + "...")
        acc[short] = acc.get(short, 0.0) + d
if acc:
    print("bounce %d: " % bounce + "  ".join("%s %.3f" % kv for kv in acc.items()) + "   | sum %.3f ms (%.2f %% of the frame)" % (sum(acc.values()), 100 * sum(acc.values()) / frame))
