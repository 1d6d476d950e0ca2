#!/usr/bin/env python3
"""Per-kernel and per-level times of the last device SAH build in a rocprofv3 results database (rocpd .db)."""
import re, sqlite3, sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_sah_items" in r[0]]
last = rows[idx[-1]:]


def short(n):
    m = re.search(r"(k_\w+)", n)
    return m.group(1) if m else n[:40]


agg = defaultdict(lambda: [0, 0.0])
for n, s, e in last:
    agg[short(n)][0] += 1
    agg[short(n)][1] += (e - s) / 1e3
print("span %.2f ms, busy %.2f ms" % ((last[-1][2] - last[0][1]) / 1e6, sum(v[1] for v in agg.values()) / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-30s %4d %9.1f us  avg %7.2f" % (k, v[0], v[1], v[1] / v[0]))
names = ["k_sah_bounds", "k_sah_decide", "k_sah_buckets", "k_sah_split", "k_sah_equal_rank", "k_sah_flagscan", "k_scan_totals", "k_sah_scatter"]
print("lev " + " ".join("%9s" % n[2:11] for n in names) + "    span")
lev, cur, st = 0, {}, None
for n, s, e in last:
    k = short(n)
    if k == "k_sah_bounds":
        if cur:
            print("%3d " % lev + " ".join("%9.1f" % cur.get(n, 0) for n in names) + "  %6.1f" % ((s - st) / 1e3))
            lev += 1
        cur, st = {}, s
    if k in names:
        cur[k] = (e - s) / 1e3
