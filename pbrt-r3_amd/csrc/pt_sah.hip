// pt_sah.hip -- the SAH binary build on the GPU (src/accelerators/bvh/build/sah.rs:31-170 via build/node.rs; the host version is
// Builder::build in pt_bvh.cpp).
//
// The reference recurses over ranges of one primitive array; a range's split depends only on the range, so all ranges of one tree
// level can be split at once.  One level = seven small kernels over the item array:
//   k_sah_bounds     every item of a new node adds its box and centroid to the node's bounds (ordered-uint atomicMin / Max; waves whose
//                    64 items belong to one node reduce in registers first, so the root costs 16 K atomics per word, not 1 M)
//   k_sah_decide     per new node: leaf (n <= maxnodeprims, or no centroid extent), else the split axis = widest centroid axis
//   k_sah_buckets    every item of a splitting node: its bucket (12 buckets over the centroid bounds: one subtract, one IEEE divide,
//                    floor), bucket counts and bucket boxes by atomics
//   k_sah_split      per splitting node: the 11 candidate costs with the reference's own loops and f32 order, the cheapest bucket, the two
//                    children's ranges (left count = the counts of the buckets up to it); new node numbers from one atomic counter
//   k_sah_flags + exclusive scan + k_sah_scatter   the stable partition of every splitting range at once: an item's new place is the
//                    range's start plus its rank among the range's left items (or the split point plus its rank among the right ones)
// Everything is integer work, exact min / max, or the same f32 expressions as the host's (-ffp-contract=off on both sides), and a stable
// partition has one result -- so the binary tree and the item order equal the host builder's.  Node NUMBERING differs (level order
// here); it is only ever link-walked.  What is not done here: the equal-counts fallback of a split that leaves one side empty (it
// needs a stable sort of the range by centroid) and non-finite bounds -- both raise `fallback` and the caller builds on the host.
// -0.0 and +0.0 bounds compare equal on the host (first come, first kept) and ordered here (-0 < +0): pt_context.cpp writes zeros of
// node boxes as +0 for both builders.
#include <sys/syscall.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pt_bvh.h"

namespace ptbvh {

namespace {

constexpr int kB = 12;                   // SAH buckets (sah.rs)
constexpr uint32_t ST_NEW = 0, ST_SPLIT = 1, ST_LEAF = 2;
constexpr uint32_t kEqualMode = 0xffffffffu;   // SNodes::minb of a node split by equal counts
constexpr uint32_t kEqualMax = 4096;           // longest range the equal-counts fallback orders on the device
constexpr uint32_t kEqualGrid = 256;           // workgroups of k_sah_equal_rank (each takes every 256th listed node)

__device__ inline uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
__device__ inline float fmin_le(float a, float b) { return a <= b ? a : b; }
__device__ inline float fmax_ge(float a, float b) { return a >= b ? a : b; }

struct SItem { float lo[3], hi[3], c[3]; uint32_t prim; };      // 40 bytes, as the host's Item
struct SNodes {                           // binary nodes, SoA over node id
    uint32_t *lo, *hi;                    // item range
    uint32_t* nb;                         // [12][cap]: ordered-uint bounds of the items (lo xyz, hi xyz), then of their centroids
    uint32_t* state;
    uint32_t *axis, *left, *right, *mid, *minb, *slot;
    uint32_t cap;
};

__global__ __launch_bounds__(256) void k_sah_items(const float* __restrict__ raw, uint32_t n, SItem* items, uint32_t* node_of, uint32_t* flags) {
    const float eps = 1.1920929e-7f * 2.0f;          // BOUND_EPS (build/node.rs:13)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        SItem it;
        bool bad = false;
        for (int a = 0; a < 3; a++) {
            it.lo[a] = raw[(size_t)i * 6 + a] - eps;
            it.hi[a] = raw[(size_t)i * 6 + 3 + a] + eps;
            it.c[a] = (it.lo[a] + it.hi[a]) * 0.5f;
            bad |= !(fabsf(it.lo[a]) <= 3.4028235e38f) || !(fabsf(it.hi[a]) <= 3.4028235e38f);
        }
        it.prim = i;
        items[i] = it;
        node_of[i] = 0;
        if (bad) atomicOr(flags, 1u);
    }
}
__global__ void k_sah_root(SNodes N, uint32_t n, uint32_t* counters) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        N.lo[0] = 0; N.hi[0] = n; N.state[0] = ST_NEW;
        for (int q = 0; q < 12; q++) N.nb[(size_t)q * N.cap] = (q % 6) < 3 ? 0xffffffffu : 0u;
        counters[0] = 1;      // nodes allocated
    }
}
// bounds of the new nodes
constexpr int kBoundsT = 1024;
__global__ __launch_bounds__(kBoundsT) void k_sah_bounds(const SItem* __restrict__ items, const uint32_t* __restrict__ node_of, uint32_t n, SNodes N) {
    __shared__ uint32_t s_v[12];
    __shared__ uint32_t s_node;
    const uint32_t i = blockIdx.x * kBoundsT + threadIdx.x;
    if (threadIdx.x < 12) s_v[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xffffffffu : 0u;
    if (threadIdx.x == 0) s_node = node_of[blockIdx.x * kBoundsT];
    __syncthreads();
    const uint32_t first = s_node;
    const bool in = i < n;
    const uint32_t nd = in ? node_of[i] : 0xffffffffu;
    const bool act = in && N.state[nd] == ST_NEW;
    if (__ballot(act) != 0ull) {
        uint32_t v[12];
        if (act) {
            const SItem it = items[i];
            for (int a = 0; a < 3; a++) { v[a] = f2ord(it.lo[a]); v[3 + a] = f2ord(it.hi[a]); v[6 + a] = f2ord(it.c[a]); v[9 + a] = v[6 + a]; }
        } else {
            for (int a = 0; a < 3; a++) { v[a] = 0xffffffffu; v[3 + a] = 0u; v[6 + a] = 0xffffffffu; v[9 + a] = 0u; }
        }
        // Items of one node are consecutive, so a wave holds a few runs of equal node ids: a segmented scan (min / max are exact in any
        // order) leaves each run's reduction in its last lane.  Near the root every wave of the launch would then queue on the same twelve
        // words (187 us a level at a million items), so the runs of the group's FIRST node meet in LDS and go out once per group.
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t key = act ? nd : 0xffffffffu;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t ko = (uint32_t)__shfl_up((int)key, off, 64);
            const bool take = lane >= (uint32_t)off && ko == key;
            for (int k = 0; k < 12; k++) {
                const uint32_t o = (uint32_t)__shfl_up((int)v[k], off, 64);
                if (take) v[k] = (k % 6) < 3 ? min(v[k], o) : max(v[k], o);
            }
        }
        const uint32_t kn = (uint32_t)__shfl_down((int)key, 1, 64);
        if (act && (lane == 63 || kn != key)) {
            if (nd == first) {
                for (int k = 0; k < 12; k++) {
                    if ((k % 6) < 3) atomicMin(&s_v[k], v[k]);
                    else atomicMax(&s_v[k], v[k]);
                }
            } else {
                for (int k = 0; k < 12; k++) {
                    if ((k % 6) < 3) atomicMin(&N.nb[(size_t)k * N.cap + nd], v[k]);
                    else atomicMax(&N.nb[(size_t)k * N.cap + nd], v[k]);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 12 && N.state[first] == ST_NEW) {
        const uint32_t k = threadIdx.x;
        if ((k % 6) < 3) atomicMin(&N.nb[(size_t)k * N.cap + first], s_v[k]);
        else atomicMax(&N.nb[(size_t)k * N.cap + first], s_v[k]);
    }
}
__device__ inline int max_extent(const float* lo, const float* hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx > dy && dx > dz) return 0;
    if (dy > dz) return 1;
    return 2;
}
// per new node: leaf or split axis; splitting nodes get a slot of the bucket arrays
__global__ __launch_bounds__(256) void k_sah_decide(SNodes N, const uint32_t* act, uint32_t max_prims, uint32_t* bcnt, uint32_t* bbox, uint32_t* counters) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= counters[5]) return;             // this level's new nodes (k_sah_roll): the host launches an upper bound and does not wait for the count
    const uint32_t nd = act[k];
    const uint32_t n = N.hi[nd] - N.lo[nd];
    float clo[3], chi[3];
    for (int a = 0; a < 3; a++) { clo[a] = ord2f(N.nb[(size_t)(6 + a) * N.cap + nd]); chi[a] = ord2f(N.nb[(size_t)(9 + a) * N.cap + nd]); }
    const int dim = max_extent(clo, chi);
    if (n <= max_prims || clo[dim] == chi[dim]) { N.state[nd] = ST_LEAF; atomicAdd(&counters[3], 1u); return; }
    N.state[nd] = ST_SPLIT;
    N.axis[nd] = (uint32_t)dim;
    const uint32_t s = atomicAdd(&counters[1], 1u);      // slot among this level's splitting nodes
    N.slot[nd] = s;
    for (int b = 0; b < kB; b++) {
        bcnt[(size_t)s * kB + b] = 0;
        for (int a = 0; a < 3; a++) { bbox[((size_t)s * kB + b) * 6 + a] = 0xffffffffu; bbox[((size_t)s * kB + b) * 6 + 3 + a] = 0u; }
    }
}
__device__ inline int bucket_of(const float* clo, const float* chi, const float* c, int dim) {      // sah.rs bucket index (pt_bvh.cpp bucket_index)
    float o = c[dim] - clo[dim];
    if (chi[dim] > clo[dim]) o = o / (chi[dim] - clo[dim]);
    int b = (int)floorf((float)kB * o);
    if (b > kB - 1) b = kB - 1;
    if (b < 0) b = 0;
    return b;
}
// Bucket counts and boxes: 7 words per item.  Sent straight to memory they queue on 84 words near the root, and deep in the tree -- every
// item its own atomics, across the fabric -- they took 0.2 ms a level at a million items.  A workgroup therefore keeps a table per RUN of
// items of one splitting node (a node's items are consecutive, so a node is one run of a group) in LDS, up to R of them, and writes each
// table out once: with plain stores when the whole node lies inside the group (k_sah_decide left the words at their identities), with
// atomics for the first and last run, which may continue in the neighbours.  Runs beyond R go to memory item by item.
template <int T, int R>
__global__ __launch_bounds__(T) void k_sah_buckets(const SItem* __restrict__ items, const uint32_t* __restrict__ node_of, uint32_t n, SNodes N, uint32_t* bcnt, uint32_t* bbox,
                                                  uint8_t* bucket) {
    constexpr int W = kB * 7;                      // words per table: kB counts, then kB boxes of 6
    __shared__ uint32_t s_tab[R * W];
    __shared__ uint32_t s_run_node[R];
    __shared__ uint32_t s_wruns[T / 64];
    const uint32_t i = blockIdx.x * T + threadIdx.x;
    for (uint32_t e = threadIdx.x; e < (uint32_t)(R * W); e += T) { const uint32_t w = e % W; s_tab[e] = w < (uint32_t)kB ? 0u : (((w - kB) % 6) < 3 ? 0xffffffffu : 0u); }
    const bool in = i < n;
    const uint32_t nd = in ? node_of[i] : 0xffffffffu;
    const bool split = in && N.state[nd] == ST_SPLIT;
    const bool head = split && (threadIdx.x == 0 || node_of[i - 1] != nd);
    const uint64_t hb = __ballot(head);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wruns[wave] = (uint32_t)__popcll(hb);
    __syncthreads();
    uint32_t rbase = 0, total = 0;
    for (uint32_t w = 0; w < (uint32_t)(T / 64); w++) { const uint32_t c = s_wruns[w]; if (w < wave) rbase += c; total += c; }
    const uint32_t r = rbase + (uint32_t)__popcll(hb & ((2ull << lane) - 1ull)) - 1u;        // the run of this item, if its node splits
    if (head && r < (uint32_t)R) s_run_node[r] = nd;
    if (split) {
        const SItem it = items[i];
        const int dim = (int)N.axis[nd];
        float clo[3], chi[3];
        for (int a = 0; a < 3; a++) { clo[a] = ord2f(N.nb[(size_t)(6 + a) * N.cap + nd]); chi[a] = ord2f(N.nb[(size_t)(9 + a) * N.cap + nd]); }
        const int b = bucket_of(clo, chi, it.c, dim);
        bucket[i] = (uint8_t)b;
        if (r < (uint32_t)R) {
            uint32_t* t = s_tab + r * W;
            atomicAdd(&t[b], 1u);
            for (int a = 0; a < 3; a++) { atomicMin(&t[kB + b * 6 + a], f2ord(it.lo[a])); atomicMax(&t[kB + b * 6 + 3 + a], f2ord(it.hi[a])); }
        } else {
            const size_t s = (size_t)N.slot[nd] * kB + b;
            atomicAdd(&bcnt[s], 1u);
            for (int a = 0; a < 3; a++) { atomicMin(&bbox[s * 6 + a], f2ord(it.lo[a])); atomicMax(&bbox[s * 6 + 3 + a], f2ord(it.hi[a])); }
        }
    }
    __syncthreads();
    const uint32_t n_runs = total < (uint32_t)R ? total : (uint32_t)R;
    for (uint32_t e = threadIdx.x; e < n_runs * W; e += T) {
        const uint32_t q = e / W, w = e % W;
        const uint32_t bk = w < (uint32_t)kB ? w : (w - kB) / 6;
        if (s_tab[q * W + bk] == 0) continue;                    // an empty bucket keeps its identities
        const size_t s0 = (size_t)N.slot[s_run_node[q]] * kB;
        const uint32_t v = s_tab[e];
        const bool own = q > 0 && q + 1 < total;                 // the node's items all lie in this group
        if (w < (uint32_t)kB) {
            if (own) bcnt[s0 + w] = v; else atomicAdd(&bcnt[s0 + w], v);
        } else {
            uint32_t* dst = &bbox[s0 * 6 + (w - kB)];
            if (own) *dst = v;
            else if (((w - kB) % 6) < 3) atomicMin(dst, v);
            else atomicMax(dst, v);
        }
    }
}
struct Box { float lo[3], hi[3]; };
__device__ inline void box_grow(Box& b, const Box& o) { for (int i = 0; i < 3; i++) { b.lo[i] = fmin_le(b.lo[i], o.lo[i]); b.hi[i] = fmax_ge(b.hi[i], o.hi[i]); } }
__device__ inline float box_area(const Box& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}
// per splitting node: the SAH costs in the host's order of operations, the split, the children
__global__ __launch_bounds__(64) void k_sah_split(SNodes N, const uint32_t* act, const uint32_t* bcnt, const uint32_t* bbox, uint32_t* next_act, uint32_t* counters,
                                                 uint32_t* flags, uint32_t* eq_list) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= counters[5]) return;
    const uint32_t nd = act[k];
    if (N.state[nd] != ST_SPLIT) return;
    const size_t s = (size_t)N.slot[nd] * kB;
    int count[kB];
    Box bb[kB];
    for (int b = 0; b < kB; b++) {
        count[b] = (int)bcnt[s + b];
        for (int a = 0; a < 3; a++) {
            // an empty bucket keeps box_empty()'s +max / lowest (the atomics started from the ordered images of +inf-ish extremes: map back)
            const uint32_t lo = bbox[(s + b) * 6 + a], hi = bbox[(s + b) * 6 + 3 + a];
            bb[b].lo[a] = count[b] ? ord2f(lo) : 3.4028235e38f;
            bb[b].hi[a] = count[b] ? ord2f(hi) : -3.4028235e38f;
        }
    }
    Box bounds;
    for (int a = 0; a < 3; a++) { bounds.lo[a] = ord2f(N.nb[(size_t)a * N.cap + nd]); bounds.hi[a] = ord2f(N.nb[(size_t)(3 + a) * N.cap + nd]); }
    const float total_area = box_area(bounds);
    float cost[kB - 1];
    for (int i = 0; i < kB - 1; i++) {
        Box b0 = bb[i], b1 = bb[i + 1];
        int c0 = 0, c1 = 0;
        for (int j = 0; j <= i; j++) { box_grow(b0, bb[j]); c0 += count[j]; }
        for (int j = i + 1; j < kB; j++) { box_grow(b1, bb[j]); c1 += count[j]; }
        cost[i] = 1.0f + ((float)c0 * box_area(b0) + (float)c1 * box_area(b1)) / total_area;
    }
    float min_cost = cost[0];
    int min_bucket = 0;
    for (int i = 1; i < kB - 1; i++)
        if (cost[i] < min_cost) { min_cost = cost[i]; min_bucket = i; }
    uint32_t nl = 0;
    for (int b = 0; b <= min_bucket; b++) nl += (uint32_t)count[b];
    const uint32_t lo = N.lo[nd], hi = N.hi[nd];
    bool equal_counts = false;
    if (nl == 0 || nl == hi - lo) {
        // The cheapest split leaves one side empty: split_equal_counts (build/equal_counts.rs; pt_bvh.cpp "!have_split") -- the range in
        // stable centroid order, cut in the middle.  Ranges k_sah_equal_rank can hold in LDS are ordered here; longer ones (many primitives
        // with one centroid bucket between them: not seen outside constructed inputs) go back to the host.
        if (hi - lo > kEqualMax) { atomicOr(flags, 2u); N.state[nd] = ST_LEAF; return; }
        equal_counts = true;
        nl = (hi - lo) / 2u;
        const uint32_t e = atomicAdd(&counters[4], 1u);
        eq_list[e] = nd;
    }
    const uint32_t c = atomicAdd(&counters[0], 2u);
    if (c + 2u > N.cap) { atomicOr(flags, 4u); N.state[nd] = ST_LEAF; return; }
    N.minb[nd] = equal_counts ? kEqualMode : (uint32_t)min_bucket;
    N.mid[nd] = lo + nl;
    N.left[nd] = c; N.right[nd] = c + 1u;
    for (uint32_t ch = 0; ch < 2; ch++) {
        const uint32_t id = c + ch;
        N.lo[id] = ch ? lo + nl : lo;
        N.hi[id] = ch ? hi : lo + nl;
        N.state[id] = ST_NEW;
        for (int q = 0; q < 12; q++) N.nb[(size_t)q * N.cap + id] = (q % 6) < 3 ? 0xffffffffu : 0u;
    }
    const uint32_t a = atomicAdd(&counters[2], 2u);
    next_act[a] = c; next_act[a + 1] = c + 1u;
}
// split_equal_counts: one workgroup per such node ranks its items by (centroid along the split axis, position) -- the order std::stable_sort
// leaves -- and k_sah_scatter sends item i to lo + rank
__global__ __launch_bounds__(256) void k_sah_equal_rank(const SItem* __restrict__ items, SNodes N, const uint32_t* __restrict__ eq_list, const uint32_t* __restrict__ counters,
                                                       uint32_t* eq_rank) {
    __shared__ float s_key[kEqualMax];
    const uint32_t n_eq = counters[4];               // this level's equal-counts nodes (almost always none: the launch costs a few microseconds)
    for (uint32_t e = blockIdx.x; e < n_eq; e += gridDim.x) {
        const uint32_t nd = eq_list[e];
        const uint32_t lo = N.lo[nd], n = N.hi[nd] - lo;
        const int dim = (int)N.axis[nd];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_key[i] = items[lo + i].c[dim];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const float k = s_key[i];
            uint32_t r = 0;
            for (uint32_t j = 0; j < n; j++) r += (s_key[j] < k || (j < i && !(k < s_key[j]))) ? 1u : 0u;
            eq_rank[lo + i] = r;
        }
    }
}
// Which items go left, and the exclusive prefix sum of that flag in three steps: 1024-item blocks, the block totals by one workgroup, add back
__global__ __launch_bounds__(256) void k_sah_flagscan(const uint32_t* __restrict__ node_of, const uint8_t* __restrict__ bucket, uint32_t n, SNodes N, uint32_t* flag, uint32_t* out,
                                                     uint32_t* totals) {
    __shared__ uint32_t s_w[4];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint32_t v[4], sum = 0;
    for (int k = 0; k < 4; k++) {
        v[k] = 0u;
        if (base + k < n) {
            const uint32_t nd = node_of[base + k];
            v[k] = (N.state[nd] == ST_SPLIT && N.minb[nd] != kEqualMode && (uint32_t)bucket[base + k] <= N.minb[nd]) ? 1u : 0u;
            flag[base + k] = v[k];
        }
        sum += v[k];
    }
    uint32_t inc = sum;
    const uint32_t lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64); if (lane >= (uint32_t)off) inc += o; }
    if (lane == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += s_w[w];
    uint32_t run = wbase + inc - sum;
    for (int k = 0; k < 4; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) totals[blockIdx.x] = wbase + inc;
}
// Between two levels, on the device: the next-active count becomes the coming level's node count, the node total is noted (level l's
// nodes are numbered [lvl_totals[l - 1], lvl_totals[l])), the per-level counters start at zero.  The host launches a level's kernels over
// an upper bound of the count and looks at the counters only every few levels.
__device__ inline void sah_roll(uint32_t* counters, uint32_t* lvl_totals, uint32_t level) {
    lvl_totals[level] = counters[0];
    counters[5] = level == 0 ? 1u : counters[2];
    counters[1] = 0; counters[2] = 0; counters[4] = 0;
}
__global__ void k_sah_roll(uint32_t* counters, uint32_t* lvl_totals, uint32_t level) { sah_roll(counters, lvl_totals, level); }
__global__ __launch_bounds__(1024) void k_scan_totals(uint32_t* totals, uint32_t n_blocks, uint32_t* counters, uint32_t* lvl_totals, uint32_t next_level) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) { s_carry = 0; sah_roll(counters, lvl_totals, next_level); }       // this level's k_sah_split is done: roll for the next one
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? totals[i] : 0u;
        uint32_t inc = v;
        const uint32_t lane = threadIdx.x & 63;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64); if (lane >= (uint32_t)off) inc += o; }
        if (lane == 63) s_w[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wbase += s_w[w];
        const uint32_t carry = s_carry;
        if (i < n_blocks) totals[i] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wbase + inc;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_sah_scatter(const SItem* __restrict__ src, const uint32_t* __restrict__ node_of, const uint32_t* __restrict__ flag,
                                                    const uint32_t* __restrict__ pre, const uint32_t* __restrict__ totals, uint32_t n, SNodes N, SItem* dst, uint32_t* node_of_dst,
                                                    const uint32_t* __restrict__ eq_rank) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t nd = node_of[i];
    uint32_t to = i, child = nd;
    if (N.state[nd] == ST_SPLIT && N.minb[nd] == kEqualMode) {
        to = N.lo[nd] + eq_rank[i];
        child = to < N.mid[nd] ? N.left[nd] : N.right[nd];
    } else if (N.state[nd] == ST_SPLIT) {
        const uint32_t lo = N.lo[nd];
        const uint32_t rank_l = (pre[i] + totals[i >> 10]) - (pre[lo] + totals[lo >> 10]);        // left items of the range before this one
        if (flag[i]) { to = lo + rank_l; child = N.left[nd]; }
        else { to = N.mid[nd] + (i - lo - rank_l); child = N.right[nd]; }
    }
    dst[to] = src[i];
    node_of_dst[to] = child;
}
__global__ __launch_bounds__(256) void k_sah_export(SNodes N, uint32_t n_nodes, LbvhNode* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) return;
    LbvhNode o;
    for (int a = 0; a < 3; a++) { o.lo[a] = ord2f(N.nb[(size_t)a * N.cap + k]); o.hi[a] = ord2f(N.nb[(size_t)(3 + a) * N.cap + k]); }
    if (N.state[k] == ST_LEAF) { o.left = -1; o.right = -1; o.first = N.lo[k]; o.count = N.hi[k] - N.lo[k]; o.axis = 0; }
    else { o.left = (int32_t)N.left[k]; o.right = (int32_t)N.right[k]; o.first = 0; o.count = 0; o.axis = (uint8_t)N.axis[k]; }
    out[k] = o;
}
__global__ __launch_bounds__(256) void k_sah_order(const SItem* __restrict__ items, uint32_t n, uint32_t* order) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = items[i].prim;
}


// ---- the rest of an upload on the device (triangle-only world lists): bounds from the vertices, leaf records and shading records in leaf
// order, the collapse of the binary tree into 128-byte 4-wide nodes in the reference's depth-first numbering (qbvh_x86.rs:93-176, the host
// version is Collapser in pt_bvh.cpp), the breadth-first renumbering of the top of the tree and the finishing pass of pt_context.cpp --
// so that neither the tree nor the primitive order ever crosses PCIe.  Byte for byte the arrays the host path uploads.
// A world list with analytic spheres spliced in: sphere j is primitive sph_prim[j] (ascending).  Returns the number of spheres listed before
// primitive q and whether q is one of them; a triangle's index is q minus that count.  (A handful of spheres: a binary search per primitive.)
__device__ inline uint32_t sc_spheres_before(const uint32_t* __restrict__ sph_prim, uint32_t n_s, uint32_t q, bool* is_sphere) {
    uint32_t lo = 0, hi = n_s;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sph_prim[mid] < q) lo = mid + 1u; else hi = mid; }
    *is_sphere = lo < n_s && sph_prim[lo] == q;
    return lo;
}
__global__ __launch_bounds__(256) void k_sc_bounds(const float* __restrict__ P, const uint32_t* __restrict__ idx, uint32_t n, float* raw, const uint32_t* __restrict__ sph_prim,
                                                  const float* __restrict__ sph_bounds, uint32_t n_s) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    uint32_t t = q;
    if (n_s) {
        bool sphere;
        const uint32_t before = sc_spheres_before(sph_prim, n_s, q, &sphere);
        if (sphere) { for (int a = 0; a < 6; a++) raw[(size_t)q * 6 + a] = sph_bounds[(size_t)before * 6 + a]; return; }      // Sphere::world_bound, from the host
        t = q - before;
    }
    const float* p0 = P + 3 * (size_t)idx[3 * (size_t)t];
    const float* p1 = P + 3 * (size_t)idx[3 * (size_t)t + 1];
    const float* p2 = P + 3 * (size_t)idx[3 * (size_t)t + 2];
    for (int a = 0; a < 3; a++) {                        // union3 (triangle.rs:189-200)
        raw[(size_t)q * 6 + a] = fminf(fminf(p0[a], p1[a]), p2[a]);
        raw[(size_t)q * 6 + 3 + a] = fmaxf(fmaxf(p0[a], p1[a]), p2[a]);
    }
}
// record r = primitive order[r]: the 48-byte leaf record, the 32-byte shading record, the primitive -> record map
__global__ __launch_bounds__(256) void k_sc_records(const uint32_t* __restrict__ order, const float* __restrict__ P, const uint32_t* __restrict__ idx,
                                                   const uint32_t* __restrict__ tri_mesh, const uint32_t* __restrict__ mesh_triflags, const int32_t* __restrict__ mesh_material,
                                                   const uint32_t* __restrict__ mesh_flags, uint32_t n, PtTri* tris, PtTriInfo* tinfo, uint32_t* rec_of_prim, uint32_t* small,
                                                   const uint32_t* __restrict__ sph_prim, const uint32_t* __restrict__ sph_rec, uint32_t n_s) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    if (r == n) {                                        // one zero pad record: the kernels fetch triangle records two at a time
        PtTri pad;
        pad.p0[0] = pad.p0[1] = pad.p0[2] = 0.0f; pad.p1[0] = pad.p1[1] = pad.p1[2] = 0.0f; pad.p2[0] = pad.p2[1] = pad.p2[2] = 0.0f;
        pad.prim = 0; pad.flags = PT_TRI_LAST; pad.light1 = 0;
        tris[n] = pad;
        return;
    }
    const uint32_t q = order[r];
    uint32_t t = q;
    if (n_s) {
        bool sphere;
        const uint32_t before = sc_spheres_before(sph_prim, n_s, q, &sphere);
        if (sphere) {                                    // ptbvh::sphere_record + the shading record of the host path's `tinfo` loop
            PtTri tr;
            tr.p0[0] = __uint_as_float(sph_rec[4 * (size_t)before]); tr.p0[1] = tr.p0[2] = 0.0f;
            tr.p1[0] = tr.p1[1] = tr.p1[2] = 0.0f; tr.p2[0] = tr.p2[1] = tr.p2[2] = 0.0f;
            tr.prim = q;
            tr.flags = (sph_rec[4 * (size_t)before + 1] | PT_TRI_SPHERE) & ~PT_TRI_LAST;
            tr.light1 = 0;
            tris[r] = tr;
            PtTriInfo ti;
            ti.v[0] = ti.v[1] = ti.v[2] = 0; ti.mesh = 0; ti.light = -1; ti.material = (int32_t)sph_rec[4 * (size_t)before + 2]; ti.mesh_flags = 0; ti.pad = 0;
            tinfo[r] = ti;
            rec_of_prim[q] = r;
            return;
        }
        t = q - before;
    }
    const uint32_t v0 = idx[3 * (size_t)t], v1 = idx[3 * (size_t)t + 1], v2 = idx[3 * (size_t)t + 2], m = tri_mesh[t];
    PtTri tr;
    for (int a = 0; a < 3; a++) { tr.p0[a] = P[3 * (size_t)v0 + a]; tr.p1[a] = P[3 * (size_t)v1 + a]; tr.p2[a] = P[3 * (size_t)v2 + a]; }
    const uint32_t f = mesh_triflags[m];
    tr.prim = q;
    tr.flags = f & ~(PT_TRI_LAST | PT_TRI_SPHERE | PT_TRI_INSTANCE);
    tr.light1 = 0;
    tris[r] = tr;
    PtTriInfo ti;
    ti.v[0] = v0; ti.v[1] = v1; ti.v[2] = v2; ti.mesh = m; ti.light = -1; ti.material = mesh_material[m]; ti.mesh_flags = mesh_flags[m]; ti.pad = 0;
    tinfo[r] = ti;
    rec_of_prim[q] = r;
    if (f & PT_TRI_ONE_SIDED) small[4] = 1u;             // PtScene::any_one_sided
}
// PT_TRI_LAST closes each leaf; leaves and the largest leaf are counted
__global__ __launch_bounds__(256) void k_sc_leafmark(const LbvhNode* __restrict__ nodes, uint32_t n_nodes, PtTri* tris, uint32_t* small) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool leaf = k < n_nodes && nodes[k].count > 0;
    uint32_t cnt = 0;
    if (leaf) { cnt = nodes[k].count; tris[nodes[k].first + cnt - 1u].flags |= PT_TRI_LAST; }
    __shared__ uint32_t s_n, s_mx;
    if (threadIdx.x == 0) { s_n = 0; s_mx = 0; }
    __syncthreads();
    const unsigned long long m = __ballot(leaf);
    uint32_t mx = cnt;
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if ((threadIdx.x & 63) == 0 && m) { atomicAdd(&s_n, (uint32_t)__popcll(m)); atomicMax(&s_mx, mx); }
    __syncthreads();
    if (threadIdx.x == 0 && s_n) { atomicAdd(&small[0], s_n); atomicMax(&small[1], s_mx); }
}
// the (up to four) binary nodes that fill the slots of the 4-wide node made from binary node b (Collapser::slots_of)
__device__ inline void sc_slots(const LbvhNode* __restrict__ nodes, uint32_t b, int32_t s[4]) {
    const int32_t l = nodes[b].left, r = nodes[b].right;
    s[0] = s[1] = s[2] = s[3] = -1;
    if (nodes[l].count > 0) s[0] = l; else { s[0] = nodes[l].left; s[1] = nodes[l].right; }
    if (nodes[r].count > 0) s[2] = r; else { s[2] = nodes[r].left; s[3] = nodes[r].right; }
}
// the next level of a binary tree as a list of node numbers (HLBVH trees are not numbered level by level)
// (begin, count) of level l sit in lv[2 l], lv[2 l + 1] on the device: k_sc_level_roll writes the next pair after each level, the host launches
// over an upper bound of the count and reads the table back every few levels
__global__ __launch_bounds__(256) void k_sc_next_level(const LbvhNode* __restrict__ nodes, uint32_t* list, const uint32_t* __restrict__ lv, uint32_t level, uint32_t* counter) {
    const uint32_t begin = lv[2u * level], count = lv[2u * level + 1u];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const LbvhNode n = nodes[list[begin + i]];
    if (n.count > 0) return;
    const uint32_t at = atomicAdd(counter, 2u);
    list[at] = (uint32_t)n.left; list[at + 1] = (uint32_t)n.right;
}
__global__ void k_sc_level_roll(uint32_t* lv, uint32_t level, const uint32_t* counter) {
    const uint32_t end = lv[2u * level] + lv[2u * level + 1u];
    lv[2u * level + 2u] = end;
    lv[2u * level + 3u] = counter[0] - end;
}
// bottom-up, one launch per even binary level: 4-wide nodes in the subtree of each interior node of the level (itself included)
__global__ __launch_bounds__(256) void k_sc_size4(const LbvhNode* __restrict__ nodes, const uint32_t* __restrict__ list, uint32_t begin, uint32_t count, uint32_t* size4,
                                                 uint32_t level, uint32_t* small) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t b = list ? list[begin + i] : begin + i;          // the level's nodes: a list (HLBVH), or consecutive numbers (SAH build: level order)
    if (nodes[b].count > 0) return;
    int32_t s[4];
    sc_slots(nodes, b, s);
    uint32_t c = 1;
    for (int k = 0; k < 4; k++) if (s[k] >= 0 && nodes[s[k]].count == 0) c += size4[s[k]];
    size4[b] = c;
    atomicMax(&small[2], level / 2u + 1u);               // deepest 4-wide level
}
// top-down: the node takes the index it was given, its subtrees follow in slot order (depth-first numbering)
__global__ __launch_bounds__(256) void k_sc_idx4(const LbvhNode* __restrict__ nodes, const uint32_t* __restrict__ list, uint32_t begin, uint32_t count,
                                                const uint32_t* __restrict__ size4, uint32_t* idx4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t b = list ? list[begin + i] : begin + i;
    if (nodes[b].count > 0) return;
    int32_t s[4];
    sc_slots(nodes, b, s);
    uint32_t next = idx4[b] + 1u;
    for (int k = 0; k < 4; k++)
        if (s[k] >= 0 && nodes[s[k]].count == 0) { idx4[s[k]] = next; next += size4[s[k]]; }
}
// The first PT_TOP_BFS_NODES 4-wide nodes in level order (children in slot order), as pt_context.cpp numbers them: top[k] = depth-first index
// of the k-th, sorted[] the same indices in increasing order (a node that is not among them moves up by the number of those below it).
__global__ __launch_bounds__(1024) void k_sc_top(const LbvhNode* __restrict__ nodes, uint32_t root, const uint32_t* __restrict__ idx4, uint32_t* top_bin, uint32_t* top_sorted,
                                                 uint32_t* small) {
    __shared__ uint32_t s_list[PT_TOP_BFS_NODES];       // binary node ids, level order
    __shared__ uint32_t s_old[PT_TOP_BFS_NODES];
    __shared__ uint32_t s_scan[1024];
    __shared__ uint32_t s_total, s_lvl_begin, s_lvl_end;
    if (threadIdx.x == 0) { s_list[0] = root; s_total = 1; s_lvl_begin = 0; s_lvl_end = 1; }
    __syncthreads();
    for (;;) {
        const uint32_t lb = s_lvl_begin, le = s_lvl_end, total = s_total;
        if (lb == le || total >= PT_TOP_BFS_NODES) break;
        uint32_t carry = 0;                              // children appended so far for this level
        for (uint32_t base = lb; base < le; base += 1024u) {
            const uint32_t j = base + threadIdx.x;
            int32_t s[4] = {-1, -1, -1, -1};
            uint32_t c = 0;
            if (j < le) {
                sc_slots(nodes, s_list[j], s);
                for (int k = 0; k < 4; k++) { if (s[k] >= 0 && nodes[s[k]].count == 0) c++; else s[k] = -1; }
            }
            s_scan[threadIdx.x] = c;
            __syncthreads();
            for (uint32_t off = 1; off < 1024u; off <<= 1) {
                const uint32_t v = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0u;
                __syncthreads();
                s_scan[threadIdx.x] += v;
                __syncthreads();
            }
            uint32_t at = total + carry + s_scan[threadIdx.x] - c;
            for (int k = 0; k < 4; k++) if (s[k] >= 0) { if (at < PT_TOP_BFS_NODES) s_list[at] = (uint32_t)s[k]; at++; }
            carry += s_scan[1023];
            __syncthreads();
        }
        if (threadIdx.x == 0) { s_lvl_begin = le; s_total = min(total + carry, (uint32_t)PT_TOP_BFS_NODES); s_lvl_end = s_total; }
        __syncthreads();
    }
    const uint32_t n_top = s_total;
    for (uint32_t k = threadIdx.x; k < n_top; k += 1024u) { s_old[k] = idx4[s_list[k]]; top_bin[k] = s_list[k]; }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_top; k += 1024u) {
        const uint32_t v = s_old[k];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n_top; j++) rank += s_old[j] < v ? 1u : 0u;
        top_sorted[rank] = v;
    }
    if (threadIdx.x == 0) small[3] = n_top;
}
// where depth-first index `old` lives after the renumbering
__device__ inline uint32_t sc_new_index(uint32_t old, const uint32_t* s_sorted, const uint32_t* s_rank_of_sorted, uint32_t n_top) {
    uint32_t lo = 0, hi = n_top;                          // lower bound
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_sorted[mid] < old) lo = mid + 1; else hi = mid; }
    if (lo < n_top && s_sorted[lo] == old) return s_rank_of_sorted[lo];
    return n_top + old - lo;
}
// one 4-wide node per interior node of an even level, finished as pt_context.cpp finishes it: order tables, +0 for zeros, the split axes
// in bits 26..27 of the child references, inverted boxes in the empty slots
__global__ __launch_bounds__(256) void k_sc_emit(const LbvhNode* __restrict__ nodes, uint32_t n_nodes, const uint32_t* __restrict__ idx4, const uint32_t* __restrict__ top_bin,
                                                const uint32_t* __restrict__ top_sorted, const uint32_t* __restrict__ small, PtNode* out) {
    __shared__ uint32_t s_sorted[PT_TOP_BFS_NODES], s_rank[PT_TOP_BFS_NODES];
    const uint32_t n_top = small[3];
    for (uint32_t k = threadIdx.x; k < n_top; k += blockDim.x) s_sorted[k] = top_sorted[k];
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_top; k += blockDim.x) {      // breadth-first rank of the node whose old index is the j-th smallest
        const uint32_t old = idx4[top_bin[k]];
        uint32_t lo = 0, hi = n_top;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_sorted[mid] < old) lo = mid + 1; else hi = mid; }
        s_rank[lo] = k;
    }
    __syncthreads();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_nodes || idx4[b] == 0xffffffffu) return;
    const LbvhNode n = nodes[b];
    const LbvhNode c0 = nodes[n.left], c1 = nodes[n.right];
    int32_t s[4];
    sc_slots(nodes, b, s);
    PtNode nd;
    uint32_t axes = (uint32_t)n.axis | ((uint32_t)c0.axis << 2) | ((uint32_t)c1.axis << 4);
    const uint32_t ax_of[4] = {axes & 3u, (axes >> 2) & 3u, 0u, (axes >> 4) & 3u};
    for (int k = 0; k < 4; k++) {
        if (s[k] < 0) {
            nd.child[k] = PT_EMPTY_REF;
            for (int a = 0; a < 3; a++) { nd.bmin[a][k] = __builtin_inff(); nd.bmax[a][k] = -__builtin_inff(); }
            continue;
        }
        const LbvhNode sl = nodes[s[k]];
        for (int a = 0; a < 3; a++) {
            nd.bmin[a][k] = sl.lo[a] == 0.0f ? 0.0f : sl.lo[a];
            nd.bmax[a][k] = sl.hi[a] == 0.0f ? 0.0f : sl.hi[a];
        }
        uint32_t ref;
        if (sl.count > 0) ref = PT_LEAF_BIT | ((min(sl.count, 8u) - 1u) << PT_LEAF_COUNT_SHIFT) | sl.first;
        else ref = sc_new_index(idx4[s[k]], s_sorted, s_rank, n_top);
        nd.child[k] = (ref & ~(3u << PT_REF_AXIS_SHIFT)) | (ax_of[k] << PT_REF_AXIS_SHIFT);
        axes |= 1u << (8 + k);
    }
    nd.axes = axes;
    uint32_t lut = 0;
    for (uint32_t oct = 0; oct < 8; oct++)
        for (uint32_t k = 0; k < 3; k++)
            if ((oct >> ((axes >> (2 * k)) & 3u)) & 1u) lut |= 1u << (8 * k + oct);
    nd.order_lut = lut;
    nd.pad[0] = 0; nd.pad[1] = 0;
    out[sc_new_index(idx4[b], s_sorted, s_rank, n_top)] = nd;
}
// emissive primitives: record of each (for the host's light list), then the light numbers back into the records
__global__ __launch_bounds__(256) void k_sc_light_recs(const uint32_t* __restrict__ prims, uint32_t n, const uint32_t* __restrict__ rec_of_prim, uint32_t* recs, PtTri* tris,
                                                      PtTriInfo* tinfo) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = rec_of_prim[prims[i]];
    recs[i] = r;
    tris[r].light1 = i + 1u;
    tinfo[r].light = (int32_t)i;
}

// Device scratch for one build.  The blocks come out of an arena the calling thread keeps per device (one hipMalloc, reused by every
// later build: the 14 hipMalloc / hipFree pairs cost 7 ms per build, more than the build's kernels); what does not fit -- the first
// build, or a bigger scene -- is allocated the slow way and the arena is regrown to the high-water mark afterwards.
constexpr size_t kArenaKeepBytes = (size_t)1 << 30;          // device scratch a thread keeps between builds
constexpr size_t kStagingKeepBytes = (size_t)256 << 20;      // page-locked staging a thread keeps between uploads
struct ScratchArena {
    char* base = nullptr;
    size_t cap = 0, used = 0, wanted = 0;
    int dev = -1;
    // an upload thread that ends (`pbrt_gpu --gpus N`) gives its block back; the process's first thread ends during runtime teardown and keeps it
    ~ScratchArena() { if (base && getpid() != (pid_t)syscall(SYS_gettid)) (void)hipFree(base); }
    void begin() {
        int d = 0;
        (void)hipGetDevice(&d);
        if (d != dev) { if (base) (void)hipFree(base); base = nullptr; cap = 0; dev = d; }      // another device: its own arena
        used = 0; wanted = 0;
    }
    // a first build knows roughly what it will ask for: one block now instead of two dozen separate allocations (0.3-0.5 ms each)
    void reserve(size_t bytes) {
        if (used != 0 || cap >= bytes || bytes > kArenaKeepBytes) return;
        if (base) (void)hipFree(base);
        base = nullptr; cap = 0;
        void* q = nullptr;
        if (hipMalloc(&q, bytes) == hipSuccess) { base = (char*)q; cap = bytes; }
    }
    void end() {                                                   // every block handed out has been returned (Scratch destructors ran)
        // kept between uploads while it is small: a 16 M-triangle build leaves ~3 GB here, which no later 1 M-triangle rebuild needs (the arena is a
        // 7 ms saving per build, not worth gigabytes of an idle thread's device memory)
        if (cap > kArenaKeepBytes && wanted <= kArenaKeepBytes) { (void)hipFree(base); base = nullptr; cap = 0; }
        if (wanted > cap && wanted <= kArenaKeepBytes) {
            if (base) (void)hipFree(base);
            base = nullptr; cap = 0;
            void* q = nullptr;
            if (hipMalloc(&q, wanted) == hipSuccess) { base = (char*)q; cap = wanted; }
        }
    }
};
static thread_local ScratchArena t_arena;
struct Scratch {
    void* p = nullptr;
    bool own = false;
    ~Scratch() { if (p && own) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) {
        bytes = (bytes ? bytes : 16) + 255 & ~(size_t)255;
        t_arena.wanted += bytes;
        if (t_arena.used + bytes <= t_arena.cap) { p = t_arena.base + t_arena.used; t_arena.used += bytes; own = false; return hipSuccess; }
        own = true;
        return hipMalloc(&p, bytes);
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};
struct ArenaScope { ArenaScope() { t_arena.begin(); } ~ArenaScope() { t_arena.end(); } };
// page-locked staging for a scene's vertices / indices, kept by the calling thread and regrown
static float* scene_staging(size_t bytes) {
    struct Holder {
        void* buf = nullptr;
        size_t cap = 0;
        ~Holder() { if (buf && getpid() != (pid_t)syscall(SYS_gettid)) (void)hipHostFree(buf); }
    };
    static thread_local Holder h;
    if (bytes == 0) {                          // trim request (after an upload): a big scene's staging does not stay pinned for the thread's life
        if (h.cap > kStagingKeepBytes) { (void)hipHostFree(h.buf); h.buf = nullptr; h.cap = 0; }
        return nullptr;
    }
    if (bytes > h.cap) {
        if (h.buf) (void)hipHostFree(h.buf);
        h.buf = nullptr; h.cap = 0;
        if (hipHostMalloc(&h.buf, bytes, hipHostMallocDefault) != hipSuccess) { h.buf = nullptr; return nullptr; }
        h.cap = bytes;
    }
    return (float*)h.buf;
}
#define SAH_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { if (err) *err = e_; return -1; } } while (0)

}  // namespace

// SAH binary build on the device.  raw_bounds = n x {lo[3], hi[3]} (host memory, not yet inflated).  On success (0): order[k] = primitive
// stored k-th, nodes[0] the root.  1: the host has to build instead (equal-counts fallback needed, non-finite bounds, too few items);
// -1: HIP error.
// The scene's arrays go up through page-locked staging filled by the host's threads (from pageable memory the runtime stages the copy itself at
// a few GB/s); the per-mesh tables are tiny.  Then the triangles' bounds, on the device.
static int upload_scene_arrays(hipStream_t st, const SceneIn& sc, Scratch& d_P, Scratch& d_idx, Scratch& d_tmesh, Scratch& d_mtab, Scratch& d_sph, float* d_raw, hipError_t* err) {
    const uint32_t n = sc.n_tris, ns = sc.n_spheres;
    const size_t nP = 3 * (size_t)sc.n_vertices, nI = 3 * (size_t)n, nm = sc.n_meshes;
    SAH_TRY(d_P.alloc(nP * 4)); SAH_TRY(d_idx.alloc(nI * 4)); SAH_TRY(d_tmesh.alloc((size_t)n * 4)); SAH_TRY(d_mtab.alloc(nm * 12 + 16));
    SAH_TRY(d_sph.alloc((size_t)ns * 44 + 16));          // per sphere: primitive number, bound (6 floats), record words (4)
    float* stage = scene_staging((nP + nI + n + 3 * nm + 11 * (size_t)ns) * 4);
    if (!stage) return 1;
    uint32_t* su = reinterpret_cast<uint32_t*>(stage);
    parallel_for(nP, [&](size_t a, size_t b) { std::memcpy(stage + a, sc.P + a, (b - a) * 4); });
    parallel_for(nI, [&](size_t a, size_t b) { std::memcpy(su + nP + a, sc.indices + a, (b - a) * 4); });
    parallel_for((size_t)n, [&](size_t a, size_t b) { std::memcpy(su + nP + nI + a, sc.tri_mesh + a, (b - a) * 4); });
    std::memcpy(su + nP + nI + n, sc.mesh_triflags, nm * 4);
    std::memcpy(su + nP + nI + n + nm, sc.mesh_material, nm * 4);
    std::memcpy(su + nP + nI + n + 2 * nm, sc.mesh_flags, nm * 4);
    uint32_t* ssph = su + nP + nI + n + 3 * nm;          // [ns] primitive numbers, [6 ns] bounds, [4 ns] record words
    if (ns) {
        std::memcpy(ssph, sc.sph_prim, (size_t)ns * 4);
        std::memcpy(ssph + ns, sc.sph_bounds, (size_t)ns * 24);
        std::memcpy(ssph + 7 * (size_t)ns, sc.sph_rec, (size_t)ns * 16);
    }
    if (nP) SAH_TRY(hipMemcpyAsync(d_P.p, stage, nP * 4, hipMemcpyHostToDevice, st));
    if (nI) SAH_TRY(hipMemcpyAsync(d_idx.p, su + nP, nI * 4, hipMemcpyHostToDevice, st));
    if (n) SAH_TRY(hipMemcpyAsync(d_tmesh.p, su + nP + nI, (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (nm) SAH_TRY(hipMemcpyAsync(d_mtab.p, su + nP + nI + n, nm * 12, hipMemcpyHostToDevice, st));
    if (ns) SAH_TRY(hipMemcpyAsync(d_sph.p, ssph, (size_t)ns * 44, hipMemcpyHostToDevice, st));
    const uint32_t np = n + ns;
    k_sc_bounds<<<(np + 255u) / 256u, 256, 0, st>>>(d_P.as<float>(), d_idx.as<uint32_t>(), np, d_raw, d_sph.as<uint32_t>(), d_sph.as<float>() + ns, ns);
    SAH_TRY(hipGetLastError());
    return 0;
}

// Records, shading records, collapse and finishing pass for a binary tree that sits on the device (`levels`: its nodes level by level, as
// ranges of `list`, or of the node numbers themselves when `list` is null).  The blocks of `sout` belong to the caller afterwards.
static int finish_scene(hipStream_t st, const LbvhNode* bn, uint32_t n_nodes, uint32_t root, const std::vector<std::pair<uint32_t, uint32_t>>& levels, const uint32_t* list,
                        const uint32_t* d_order, const float* d_P, const uint32_t* d_idx, const uint32_t* d_tmesh, const uint32_t* mtab, uint32_t nm, uint32_t n, SceneOut* sout,
                        hipError_t* err, const uint32_t* d_sph = nullptr, uint32_t n_s = 0) {          // n: primitives of the merged list (triangles + spheres)
        struct Owned { void* p = nullptr; ~Owned() { if (p) (void)hipFree(p); } void* release() { void* q = p; p = nullptr; return q; } };
        Owned o_tris, o_tinfo, o_rop, o_nodes;
        SAH_TRY(hipMalloc(&o_tris.p, ((size_t)n + 1) * sizeof(PtTri)));
        SAH_TRY(hipMalloc(&o_tinfo.p, (size_t)n * sizeof(PtTriInfo)));
        SAH_TRY(hipMalloc(&o_rop.p, (size_t)n * 4));
        Scratch d_size4, d_idx4, d_topbin, d_topsorted, d_s2;
        SAH_TRY(d_size4.alloc((size_t)n_nodes * 4)); SAH_TRY(d_idx4.alloc((size_t)n_nodes * 4));
        SAH_TRY(d_topbin.alloc(PT_TOP_BFS_NODES * 4)); SAH_TRY(d_topsorted.alloc(PT_TOP_BFS_NODES * 4)); SAH_TRY(d_s2.alloc(64));
        uint32_t* small = d_s2.as<uint32_t>();          // [0] leaves, [1] largest leaf, [2] deepest 4-wide level, [3] n_top, [4] any one-sided
        SAH_TRY(hipMemsetAsync(small, 0, 64, st));
        SAH_TRY(hipMemsetAsync(d_size4.p, 0, (size_t)n_nodes * 4, st));
        SAH_TRY(hipMemsetAsync(d_idx4.p, 0xff, (size_t)n_nodes * 4, st));
        k_sc_records<<<(n + 1u + 255u) / 256u, 256, 0, st>>>(d_order, d_P, d_idx, d_tmesh, mtab,
                                                           reinterpret_cast<const int32_t*>(mtab + nm), mtab + 2 * nm, n, (PtTri*)o_tris.p, (PtTriInfo*)o_tinfo.p, (uint32_t*)o_rop.p, small,
                                                           d_sph, d_sph ? d_sph + 7 * (size_t)n_s : nullptr, n_s);
        k_sc_leafmark<<<(n_nodes + 255u) / 256u, 256, 0, st>>>(bn, n_nodes, (PtTri*)o_tris.p, small);
        const size_t n_lv = levels.size();
        for (size_t l = n_lv; l-- > 0;)
            if ((l & 1) == 0 && levels[l].second) k_sc_size4<<<(levels[l].second + 255u) / 256u, 256, 0, st>>>(bn, list, levels[l].first, levels[l].second, d_size4.as<uint32_t>(), (uint32_t)l, small);
        SAH_TRY(hipMemsetAsync(d_idx4.as<uint32_t>() + root, 0, 4, st));                // the root is 4-wide node 0
        for (size_t l = 0; l < n_lv; l += 2)
            if (levels[l].second) k_sc_idx4<<<(levels[l].second + 255u) / 256u, 256, 0, st>>>(bn, list, levels[l].first, levels[l].second, d_size4.as<uint32_t>(), d_idx4.as<uint32_t>());
        k_sc_top<<<1, 1024, 0, st>>>(bn, root, d_idx4.as<uint32_t>(), d_topbin.as<uint32_t>(), d_topsorted.as<uint32_t>(), small);
        SAH_TRY(hipGetLastError());
        uint32_t h_small[8], n4 = 0;
        LbvhNode h_root;
        SAH_TRY(hipMemcpyAsync(h_small, small, 32, hipMemcpyDeviceToHost, st));
        SAH_TRY(hipMemcpyAsync(&n4, d_size4.as<uint32_t>() + root, 4, hipMemcpyDeviceToHost, st));
        SAH_TRY(hipMemcpyAsync(&h_root, bn + root, sizeof(LbvhNode), hipMemcpyDeviceToHost, st));
        SAH_TRY(hipStreamSynchronize(st));
        if (n4 == 0 || n4 >= (1u << 25)) return 1;
        SAH_TRY(hipMalloc(&o_nodes.p, (size_t)n4 * sizeof(PtNode)));
        k_sc_emit<<<(n_nodes + 255u) / 256u, 256, 0, st>>>(bn, n_nodes, d_idx4.as<uint32_t>(), d_topbin.as<uint32_t>(), d_topsorted.as<uint32_t>(), small, (PtNode*)o_nodes.p);
        SAH_TRY(hipGetLastError());
        SAH_TRY(hipStreamSynchronize(st));
        sout->n_nodes4 = n4; sout->n_leaves = h_small[0]; sout->max_leaf = h_small[1]; sout->max_depth4 = h_small[2]; sout->n_top = h_small[3]; sout->any_one_sided = h_small[4];
        for (int a = 0; a < 3; a++) { sout->root_lo[a] = h_root.lo[a]; sout->root_hi[a] = h_root.hi[a]; }
        sout->d_nodes = o_nodes.release(); sout->d_tris = o_tris.release(); sout->d_tinfo = o_tinfo.release(); sout->d_rec_of_prim = o_rop.release();
        return 0;
}

// With `scene` the bounds come from the scene's vertices (uploaded here) and nothing is read back: the leaf records, the shading records and
// the finished 4-wide node array are produced on the device and handed over in `sout` (plain hipMalloc blocks the caller owns from then on).
static int sah_build(hipStream_t st, const float* raw_bounds, uint32_t n, uint32_t max_prims, NoInitVec<uint32_t>* order, NoInitVec<LbvhNode>* nodes, hipError_t* err,
                     const SceneIn* scene, SceneOut* sout) {
    ArenaScope arena_scope;          // declared before every Scratch: destroyed after them
    const bool trace = std::getenv("PBRTGPU_BUILD_TRACE") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (err) *err = hipSuccess;
    if (n < 2 || n > (1u << 26) || max_prims < 2) return 1;
    const uint32_t cap = 2u * n + 2u;
    const uint32_t max_split = n / (max_prims + 1u) + 2u;          // nodes splitting in one level hold more than max_prims items each
    t_arena.reserve((size_t)n * (scene ? 480 : 400) + (size_t)max_split * 340 + (scene ? (size_t)scene->n_vertices * 12 : 0) + ((size_t)2 << 20));
    Scratch d_raw, d_items[2], d_nodeof[2], d_nodes, d_bcnt, d_bbox, d_bucket, d_flag, d_pre, d_tot, d_act[2], d_small, d_export, d_order;
    SAH_TRY(d_raw.alloc((size_t)n * 24));
    for (int i = 0; i < 2; i++) { SAH_TRY(d_items[i].alloc((size_t)n * sizeof(SItem))); SAH_TRY(d_nodeof[i].alloc((size_t)n * 4)); SAH_TRY(d_act[i].alloc((size_t)cap * 4)); }
    SAH_TRY(d_nodes.alloc((size_t)cap * 4 * 21));
    SAH_TRY(d_bcnt.alloc((size_t)max_split * kB * 4));
    SAH_TRY(d_bbox.alloc((size_t)max_split * kB * 6 * 4));
    SAH_TRY(d_bucket.alloc(n));
    SAH_TRY(d_flag.alloc((size_t)n * 4));
    SAH_TRY(d_pre.alloc((size_t)n * 4));
    const uint32_t n_sblocks = (n + 1023u) / 1024u;
    SAH_TRY(d_tot.alloc((size_t)n_sblocks * 4));
    SAH_TRY(d_small.alloc(64));
    Scratch d_eqlist, d_eqrank;
    SAH_TRY(d_eqlist.alloc((size_t)max_split * 4));
    SAH_TRY(d_eqrank.alloc((size_t)n * 4));
    SNodes N;
    {
        uint32_t* b = d_nodes.as<uint32_t>();
        N.cap = cap;
        N.lo = b; N.hi = b + cap; N.nb = b + 2 * (size_t)cap;
        uint32_t* r = b + 14 * (size_t)cap;
        N.state = r; N.axis = r + cap; N.left = r + 2 * (size_t)cap; N.right = r + 3 * (size_t)cap; N.mid = r + 4 * (size_t)cap; N.minb = r + 5 * (size_t)cap; N.slot = r + 6 * (size_t)cap;
    }
    uint32_t* counters = d_small.as<uint32_t>();       // [0] nodes, [1] splitting this level, [2] next active, [3] leaves
    uint32_t* flags = counters + 8;
    SAH_TRY(hipMemsetAsync(d_small.p, 0, 64, st));
    const uint32_t blocks = (n + 255u) / 256u;
    Scratch d_P, d_idx, d_tmesh, d_mtab, d_sph;
    if (scene) {
        const int urc = upload_scene_arrays(st, *scene, d_P, d_idx, d_tmesh, d_mtab, d_sph, d_raw.as<float>(), err);
        if (urc != 0) return urc;
        if (trace) { const double ta = now(); (void)hipStreamSynchronize(st); std::fprintf(stderr, "[bvh] device scene: vertices + indices staged and uploaded, bounds on the device: %.2f ms since entry, %.2f ms waiting\n", now() - t0, now() - ta); }
    } else {
        SAH_TRY(hipMemcpyAsync(d_raw.p, raw_bounds, (size_t)n * 24, hipMemcpyHostToDevice, st));
        if (trace) { const double ta = now(); (void)hipStreamSynchronize(st); std::fprintf(stderr, "[bvh] device SAH: bounds upload %.2f ms (%.1f MB)\n", now() - ta, n * 24e-6); }
    }
    k_sah_items<<<blocks < 4096u ? blocks : 4096u, 256, 0, st>>>(d_raw.as<float>(), n, d_items[0].as<SItem>(), d_nodeof[0].as<uint32_t>(), flags);
    k_sah_root<<<1, 64, 0, st>>>(N, n, counters);
    const uint32_t root_act = 0;
    SAH_TRY(hipMemcpyAsync(d_act[0].p, &root_act, 4, hipMemcpyHostToDevice, st));
    const double t1 = now();
    int cur = 0, acur = 0;
    uint32_t levels = 0;
    uint32_t host_small[12];
    Scratch d_totals;
    SAH_TRY(d_totals.alloc(520 * 4));
    uint64_t bound = 1;                                 // upper bound of a level's node count: twice the level before, never more than the nodes there can be
    k_sah_roll<<<1, 1, 0, st>>>(counters, d_totals.as<uint32_t>(), 0u);      // later rolls ride on k_scan_totals
    for (;;) {
        if (levels >= 512u) return 1;
        // a look at the counters every fourth level (every second one deep in the tree, where levels are short): a level launched after the
        // last one finds no new node and moves nothing, so looking late costs little and looking every time cost a round trip per level
        if (levels > 0 && (levels % 4u == 0u || (levels > 16u && levels % 2u == 0u))) {
            SAH_TRY(hipMemcpyAsync(host_small, counters, 48, hipMemcpyDeviceToHost, st));
            SAH_TRY(hipStreamSynchronize(st));
            if (host_small[8] != 0) {
                if (trace) std::fprintf(stderr, "[bvh] device SAH gives up near level %u: flags %u (1 non-finite bounds, 2 an equal-counts range too long, 4 node capacity)\n", levels, host_small[8]);
                return 1;
            }
            if (host_small[5] == 0) break;
        }
        const uint32_t nb = (uint32_t)std::min<uint64_t>(bound, cap);
        SItem* items = d_items[cur].as<SItem>();
        uint32_t* node_of = d_nodeof[cur].as<uint32_t>();
        const uint32_t* act = d_act[acur].as<uint32_t>();
        uint32_t* next_act = d_act[acur ^ 1].as<uint32_t>();
        const bool wide = (uint64_t)n >= bound * 64u;         // nodes of 64 items and more on average: few runs in a wide group
        k_sah_bounds<<<(n + kBoundsT - 1u) / kBoundsT, kBoundsT, 0, st>>>(items, node_of, n, N);
        k_sah_decide<<<(nb + 255u) / 256u, 256, 0, st>>>(N, act, max_prims, d_bcnt.as<uint32_t>(), d_bbox.as<uint32_t>(), counters);
        if (wide) k_sah_buckets<1024, 32><<<(n + 1023u) / 1024u, 1024, 0, st>>>(items, node_of, n, N, d_bcnt.as<uint32_t>(), d_bbox.as<uint32_t>(), d_bucket.as<uint8_t>());
        else k_sah_buckets<256, 56><<<blocks, 256, 0, st>>>(items, node_of, n, N, d_bcnt.as<uint32_t>(), d_bbox.as<uint32_t>(), d_bucket.as<uint8_t>());
        k_sah_split<<<(nb + 63u) / 64u, 64, 0, st>>>(N, act, d_bcnt.as<uint32_t>(), d_bbox.as<uint32_t>(), next_act, counters, flags, d_eqlist.as<uint32_t>());
        k_sah_equal_rank<<<kEqualGrid, 256, 0, st>>>(items, N, d_eqlist.as<uint32_t>(), counters, d_eqrank.as<uint32_t>());
        k_sah_flagscan<<<n_sblocks, 256, 0, st>>>(node_of, d_bucket.as<uint8_t>(), n, N, d_flag.as<uint32_t>(), d_pre.as<uint32_t>(), d_tot.as<uint32_t>());
        k_scan_totals<<<1, 1024, 0, st>>>(d_tot.as<uint32_t>(), n_sblocks, counters, d_totals.as<uint32_t>(), levels + 1u);
        k_sah_scatter<<<blocks, 256, 0, st>>>(items, node_of, d_flag.as<uint32_t>(), d_pre.as<uint32_t>(), d_tot.as<uint32_t>(), n, N, d_items[cur ^ 1].as<SItem>(),
                                              d_nodeof[cur ^ 1].as<uint32_t>(), d_eqrank.as<uint32_t>());
        SAH_TRY(hipGetLastError());
        levels++;
        cur ^= 1; acur ^= 1;
        if (bound < cap) bound *= 2;                      // (deep, thin trees run to hundreds of levels)
    }
    // the node totals per level (totals[l] = nodes numbered before level l's were made; the loop's last rolls saw no growth)
    std::vector<uint32_t> totals(levels + 1);
    SAH_TRY(hipMemcpyAsync(totals.data(), d_totals.p, (size_t)(levels + 1) * 4, hipMemcpyDeviceToHost, st));
    SAH_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> level_begin(1, 0u);          // binary nodes of level l: [level_begin[l], level_begin[l + 1])
    for (uint32_t l = 0; l <= levels; l++) if (totals[l] > level_begin.back()) level_begin.push_back(totals[l]);
    const double t2 = now();
    const uint32_t n_nodes = totals[levels];
    SAH_TRY(d_export.alloc((size_t)n_nodes * sizeof(LbvhNode)));
    SAH_TRY(d_order.alloc((size_t)n * 4));
    k_sah_export<<<(n_nodes + 255u) / 256u, 256, 0, st>>>(N, n_nodes, d_export.as<LbvhNode>());
    k_sah_order<<<blocks, 256, 0, st>>>(d_items[cur].as<SItem>(), n, d_order.as<uint32_t>());
    SAH_TRY(hipGetLastError());
    if (scene) {
        std::vector<std::pair<uint32_t, uint32_t>> lv;
        for (size_t l = 0; l + 1 < level_begin.size(); l++) lv.push_back({level_begin[l], level_begin[l + 1] - level_begin[l]});
        const int frc = finish_scene(st, d_export.as<LbvhNode>(), n_nodes, 0u, lv, nullptr, d_order.as<uint32_t>(), d_P.as<float>(), d_idx.as<uint32_t>(), d_tmesh.as<uint32_t>(),
                                     d_mtab.as<uint32_t>(), scene->n_meshes, n, sout, err, d_sph.as<uint32_t>(), scene->n_spheres);
        (void)scene_staging(0);
        if (trace && frc == 0) std::fprintf(stderr, "[bvh] device scene (sah): %u items, %u binary nodes in %zu levels -> %u 4-wide nodes (top %u renumbered), %u leaves: setup %.2f levels %.2f records + collapse %.2f ms\n", n,
                                            n_nodes, lv.size(), sout->n_nodes4, sout->n_top, sout->n_leaves, t1 - t0, t2 - t1, now() - t2);
        return frc;
    }
    nodes->resize(n_nodes);
    order->resize(n);
    SAH_TRY(hipMemcpyAsync(nodes->data(), d_export.p, (size_t)n_nodes * sizeof(LbvhNode), hipMemcpyDeviceToHost, st));
    SAH_TRY(hipMemcpyAsync(order->data(), d_order.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    SAH_TRY(hipStreamSynchronize(st));
    if (trace) std::fprintf(stderr, "[bvh] device SAH: %u items, %u nodes, %u levels: setup %.2f levels %.2f read-back %.2f ms\n", n, n_nodes, levels, t1 - t0, t2 - t1, now() - t2);
    return 0;
}

int device_sah(hipStream_t st, const float* raw_bounds, uint32_t n, uint32_t max_prims, NoInitVec<uint32_t>* order, NoInitVec<LbvhNode>* nodes, hipError_t* err) {
    return sah_build(st, raw_bounds, n, max_prims, order, nodes, err, nullptr, nullptr);
}
int device_sah_scene(hipStream_t st, const SceneIn& in, uint32_t max_prims, SceneOut* out, hipError_t* err) {
    return sah_build(st, nullptr, in.n_prims(), max_prims, nullptr, nullptr, err, &in, out);
}
// "splitmethod" "hlbvh" the same way: bounds from the vertices on the device, Morton sort / treelets / emit_lbvh (pt_hlbvh.hip) with the tree
// and the order kept there, the upper SAH over the <= 4096 treelet roots on the host (their boxes come back, the joining nodes go up behind
// the treelets' nodes), then the same records / collapse / finishing kernels -- over level lists, since these nodes are not numbered by level.
int device_hlbvh_scene(hipStream_t st, const SceneIn& in, uint32_t max_prims, SceneOut* out, hipError_t* err) {
    ArenaScope arena_scope;
    const bool trace = std::getenv("PBRTGPU_BUILD_TRACE") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (err) *err = hipSuccess;
    const uint32_t n = in.n_prims();
    if (n < 2 || n > (1u << 26)) return 1;
    const uint32_t node_cap = 2u * n + 2u * 4096u + 8u;
    t_arena.reserve((size_t)n * 200 + (size_t)in.n_vertices * 12 + ((size_t)2 << 20));
    Scratch d_raw, d_P, d_idx, d_tmesh, d_mtab, d_sph, d_bn, d_order, d_list, d_cnt;
    SAH_TRY(d_raw.alloc((size_t)n * 24));
    SAH_TRY(d_bn.alloc((size_t)node_cap * sizeof(LbvhNode)));
    SAH_TRY(d_order.alloc((size_t)n * 4));
    const int urc = upload_scene_arrays(st, in, d_P, d_idx, d_tmesh, d_mtab, d_sph, d_raw.as<float>(), err);
    if (urc != 0) return urc;
    uint32_t n_nodes = 0, n_roots = 0;
    std::vector<LbvhNode> roots;
    const int rc = device_lbvh_keep(st, d_raw.as<float>(), n, max_prims, d_order.as<uint32_t>(), d_bn.as<LbvhNode>(), node_cap, &n_nodes, &n_roots, &roots, err);
    if (rc != 0) return rc;
    const double t1 = now();
    // upper SAH on the host; its nodes refer to roots (< n_roots) and to each other (>= n_roots there, >= n_nodes on the device)
    NoInitVec<LbvhNode> tree(roots.begin(), roots.end());
    const int32_t root_h = hlbvh_upper_tree(tree, n_roots);
    if (root_h < 0) return -2;
    const uint32_t n_upper = (uint32_t)tree.size() - n_roots;
    if (n_nodes + n_upper > node_cap) return 1;
    auto remap = [&](int32_t k) { return k < 0 || (uint32_t)k < n_roots ? k : (int32_t)((uint32_t)k - n_roots + n_nodes); };
    for (uint32_t k = n_roots; k < tree.size(); k++) { tree[k].left = remap(tree[k].left); tree[k].right = remap(tree[k].right); }
    if (n_upper) SAH_TRY(hipMemcpyAsync(d_bn.as<LbvhNode>() + n_nodes, tree.data() + n_roots, (size_t)n_upper * sizeof(LbvhNode), hipMemcpyHostToDevice, st));
    const uint32_t root = (uint32_t)remap(root_h), total = n_nodes + n_upper;
    // the tree level by level, as lists of node numbers
    SAH_TRY(d_list.alloc((size_t)total * 4 + 16)); SAH_TRY(d_cnt.alloc(16));
    SAH_TRY(hipMemcpyAsync(d_list.p, &root, 4, hipMemcpyHostToDevice, st));
    const uint32_t one = 1;
    SAH_TRY(hipMemcpyAsync(d_cnt.p, &one, 4, hipMemcpyHostToDevice, st));
    std::vector<std::pair<uint32_t, uint32_t>> levels;
    {
        constexpr uint32_t kMaxLevels = 4096;
        Scratch d_lv;
        SAH_TRY(d_lv.alloc((size_t)(2 * kMaxLevels + 4) * 4));
        const uint32_t lv0[2] = {0u, 1u};
        SAH_TRY(hipMemcpyAsync(d_lv.p, lv0, 8, hipMemcpyHostToDevice, st));
        std::vector<uint32_t> h_lv;
        uint64_t bound = 1;
        uint32_t l = 0;
        for (;;) {
            if (l >= kMaxLevels) return 1;
            const uint32_t nb = (uint32_t)std::min<uint64_t>(bound, total);
            k_sc_next_level<<<(nb + 255u) / 256u, 256, 0, st>>>(d_bn.as<LbvhNode>(), d_list.as<uint32_t>(), d_lv.as<uint32_t>(), l, d_cnt.as<uint32_t>());
            k_sc_level_roll<<<1, 1, 0, st>>>(d_lv.as<uint32_t>(), l, d_cnt.as<uint32_t>());
            SAH_TRY(hipGetLastError());
            l++;
            if (bound < total) bound *= 2;
            if (l % 8u == 0u || l >= 40u) {             // a look at the table every eighth level (a level past the last one finds nothing to do)
                h_lv.resize(2 * (size_t)l + 2);
                SAH_TRY(hipMemcpyAsync(h_lv.data(), d_lv.p, h_lv.size() * 4, hipMemcpyDeviceToHost, st));
                SAH_TRY(hipStreamSynchronize(st));
                if (h_lv[2 * (size_t)l + 1] == 0) break;
            }
        }
        for (uint32_t k = 0; k < l && h_lv[2 * (size_t)k + 1] > 0; k++) levels.push_back({h_lv[2 * (size_t)k], h_lv[2 * (size_t)k + 1]});
    }
    const double t2 = now();
    const int frc = finish_scene(st, d_bn.as<LbvhNode>(), total, root, levels, d_list.as<uint32_t>(), d_order.as<uint32_t>(), d_P.as<float>(), d_idx.as<uint32_t>(), d_tmesh.as<uint32_t>(),
                                 d_mtab.as<uint32_t>(), in.n_meshes, n, out, err, d_sph.as<uint32_t>(), in.n_spheres);
    (void)scene_staging(0);
    if (trace && frc == 0) std::fprintf(stderr, "[bvh] device scene (hlbvh): %u items, %u + %u binary nodes in %zu levels -> %u 4-wide nodes (top %u renumbered), %u leaves: upload + lower half %.2f upper + levels %.2f records + collapse %.2f ms\n",
                                        n, n_nodes, n_upper, levels.size(), out->n_nodes4, out->n_top, out->n_leaves, t1 - t0, t2 - t1, now() - t2);
    return frc;
}
void SceneOut::free_all() {
    if (d_nodes) (void)hipFree(d_nodes);
    if (d_tris) (void)hipFree(d_tris);
    if (d_tinfo) (void)hipFree(d_tinfo);
    if (d_rec_of_prim) (void)hipFree(d_rec_of_prim);
    d_nodes = d_tris = d_tinfo = d_rec_of_prim = nullptr;
}
// light i = emissive primitive prims[i] (primitive order): its record comes back for the host's light list, its number goes into the records
int device_scene_lights(hipStream_t st, SceneOut* out, const uint32_t* prims, uint32_t n, uint32_t* recs_host, hipError_t* err) {
    if (n == 0) return 0;
    void* d = nullptr;
    SAH_TRY(hipMalloc(&d, (size_t)n * 8));
    uint32_t* d_prims = (uint32_t*)d;
    hipError_t e = hipMemcpyAsync(d_prims, prims, (size_t)n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        k_sc_light_recs<<<(n + 255u) / 256u, 256, 0, st>>>(d_prims, n, (const uint32_t*)out->d_rec_of_prim, d_prims + n, (PtTri*)out->d_tris, (PtTriInfo*)out->d_tinfo);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(recs_host, d_prims + n, (size_t)n * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) { if (err) *err = e; return -1; }
    return 0;
}

}  // namespace ptbvh
