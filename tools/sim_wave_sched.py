#!/usr/bin/env python3
"""What would a different lane-scheduling policy buy k_trace?  Replays the real traversal step sequences of
path-traced rays (from the CPU oracle: same BVH, same order) through a model of one 64-lane wave and reports
lane utilisation and modelled time for
  k=1  the current kernel: one ray per lane, one phase (node or leaf) per iteration, leaf phase once PT_LEAF_MIN
       lanes are parked on a leaf, wave-level refill once PT_REFILL_MIN lanes are idle
  k=2  two rays per lane: a lane takes part in a phase if either of its rays wants it
Costs are issue slots per wave-instruction stream, taken from the k_trace ISA (node visit ~250, a pair of
triangle tests ~450, ray start ~150)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("pbrt-r3_amd")
import oracle_lib

C_NODE, C_PAIR, C_LEAF0, C_START, C_LOOP = 250.0, 450.0, 40.0, 150.0, 30.0


def rays_from_oracle(n_tris, res, spp, tile):
    orc = oracle_lib.load()
    sd = pkg.scenes.rt1m(n_tris, res=res, spp=spp, max_depth=8)
    sc = orc.scene(sd)
    lib = orc.lib
    lib.orc_step_log.argtypes = [C.c_void_p, C.POINTER(pkg.capi.pt_tile), C.c_void_p, C.c_uint64]
    lib.orc_step_log.restype = C.c_uint64
    cap = 1 << 28
    buf = np.empty(cap, np.uint8)
    t = pkg.capi.pt_tile(*tile)
    n = lib.orc_step_log(sc.h, C.byref(t), buf.ctypes.data_as(C.c_void_p), cap)
    assert n <= cap
    buf = buf[:n]
    ends = np.flatnonzero(buf == 255)
    rays, s = [], 0
    for e in ends:
        if e > s:
            rays.append(buf[s:e].copy())
        s = e + 1
    sc.close()
    return rays


def simulate(rays, k, leaf_min=24, refill_min=16, select_overhead=0.0, seed=0, dist=None):
    """One wave, `k` ray slots per lane.  Returns (time, useful lane-cost, issued lane-cost)."""
    rng = np.random.default_rng(seed)
    order = rng.permutation(len(rays))
    nxt = 0
    L = 64
    cur = [[None] * k for _ in range(L)]       # step arrays
    pos = np.zeros((L, k), np.int64)
    time = useful = issued = 0.0
    stat = {"node_t": 0.0, "node_lanes": 0.0, "node_steps": 0, "leaf_t": 0.0, "leaf_useful": 0.0, "leaf_steps": 0, "start_t": 0.0, "loop_t": 0.0}
    so = 1.0 + select_overhead

    def want(l, j):
        a = cur[l][j]
        if a is None:
            return -1
        return 0 if a[pos[l, j]] == 0 else 1

    while True:
        idle = [(l, j) for l in range(L) for j in range(k) if cur[l][j] is None]
        if nxt < len(order) and (len(idle) >= refill_min * k or len(idle) == L * k):
            started = 0
            for (l, j) in idle:
                if nxt >= len(order):
                    break
                cur[l][j] = rays[order[nxt]]; pos[l, j] = 0; nxt += 1; started += 1
            time += C_START; issued += C_START * L; useful += C_START * min(started, L); stat["start_t"] += C_START
        w = [[want(l, j) for j in range(k)] for l in range(L)]
        n_node = sum(1 for l in range(L) if 0 in w[l])
        n_leaf = sum(1 for l in range(L) if 1 in w[l])
        if n_node == 0 and n_leaf == 0:
            if nxt >= len(order):
                break
            continue
        time += C_LOOP; issued += C_LOOP * L; stat["loop_t"] += C_LOOP
        n_tri_parked = 0
        if dist is not None:
            for l in range(L):
                if 1 in w[l]:
                    j = w[l].index(1)
                    n_tri_parked += int(cur[l][j][pos[l, j]])
        go_node = (n_node != 0 and n_leaf < leaf_min) if dist is None else (n_node != 0 and n_tri_parked < dist[0])
        if go_node:
            cost = C_NODE * so
            for l in range(L):
                if 0 in w[l]:
                    j = w[l].index(0)
                    pos[l, j] += 1
                    if pos[l, j] >= len(cur[l][j]):
                        cur[l][j] = None
            time += cost; issued += cost * L; useful += C_NODE * n_node
            stat["node_t"] += cost; stat["node_lanes"] += n_node; stat["node_steps"] += 1
        else:
            pairs = 0
            lanes = []
            for l in range(L):
                if 1 in w[l]:
                    j = w[l].index(1)
                    kk = int(cur[l][j][pos[l, j]])
                    lanes.append((kk + 1) // 2)
                    pairs = max(pairs, (kk + 1) // 2)
                    pos[l, j] += 1
                    if pos[l, j] >= len(cur[l][j]):
                        cur[l][j] = None
            cost = (C_LEAF0 + C_PAIR * pairs) * so
            if dist is not None:
                rounds = (n_tri_parked + 63) // 64
                cost = C_LEAF0 + rounds * dist[1]
                lanes = []
                useful += (C_PAIR / 2) * n_tri_parked + C_LEAF0 * n_leaf
            time += cost; issued += cost * L; useful += sum(C_LEAF0 + C_PAIR * p for p in lanes)
            stat["leaf_t"] += cost; stat["leaf_useful"] += sum(C_LEAF0 + C_PAIR * p for p in lanes) / 64.0; stat["leaf_steps"] += 1
    simulate.last = stat
    return time, useful, issued


def main():
    n_tris = int(os.environ.get("SIM_TRIS", "1000000"))
    rays = rays_from_oracle(n_tris, 256, 2, (120, 120, 136, 136))
    lens = np.array([len(r) for r in rays])
    nodes = sum(int((r == 0).sum()) for r in rays); leaves = sum(int((r != 0).sum()) for r in rays)
    print("%d rays, %.1f node visits and %.1f leaf visits per ray (median length %d, p95 %d)" %
          (len(rays), nodes / len(rays), leaves / len(rays), np.median(lens), np.percentile(lens, 95)))
    rays = rays[:6000]
    base = None
    for name, k, kw in (("k=1 current", 1, {}), ("k=1 leaf_min 8", 1, {"leaf_min": 8}), ("k=1 leaf_min 40", 1, {"leaf_min": 40}), ("k=1 leaf_min 64", 1, {"leaf_min": 64}),
                        ("k=1 refill 4", 1, {"refill_min": 4}), ("k=1 refill 32", 1, {"refill_min": 32}),
                        ("dist leaves T>=64 c=300", 1, {"dist": (64, 300.0)}), ("dist leaves T>=48 c=300", 1, {"dist": (48, 300.0)}),
                        ("dist leaves T>=96 c=300", 1, {"dist": (96, 300.0)}), ("dist leaves T>=64 c=350", 1, {"dist": (64, 350.0)}),
                        ("dist leaves T>=128 c=300", 1, {"dist": (128, 300.0)}),
                        ("k=2, no select cost", 2, {}), ("k=2, +15% select cost", 2, {"select_overhead": 0.15}),
                        ("k=2, +15%, leaf_min 32", 2, {"select_overhead": 0.15, "leaf_min": 32}),
                        ("k=3, +20% select cost", 3, {"select_overhead": 0.20}), ("k=4, +25% select cost", 4, {"select_overhead": 0.25})):
        t, u, i = simulate(rays, k, **kw)
        base = base or t
        st = simulate.last
        print("%-26s time %.3e  lane utilisation %.3f  speed-up %.2fx | node %.0f%% of time at %.0f%% lanes, leaf %.0f%% at %.0f%%, start %.0f%%, loop %.0f%%"
              % (name, t, u / i, base / t, 100 * st["node_t"] / t, 100 * st["node_lanes"] / max(1, st["node_steps"]) / 64,
                 100 * st["leaf_t"] / t, 100 * st["leaf_useful"] / max(1e-9, st["leaf_t"]), 100 * st["start_t"] / t, 100 * st["loop_t"] / t))


if __name__ == "__main__":
    main()
