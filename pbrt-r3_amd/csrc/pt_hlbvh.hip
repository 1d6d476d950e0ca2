// pt_hlbvh.hip -- the lower half of the HLBVH build on the GPU (src/accelerators/bvh/build/hlbvh.rs:20-101, :157-245, :354-428).
//
// What runs here: primitive bound inflation + centroids (build/node.rs:13, :88-97), the scene bound, the 30-bit Morton codes
// (hlbvh.rs:20-47, :367-373), a stable 5-pass x 6-bit LSD radix sort of (code, primitive) pairs (hlbvh.rs:49-101 does the same
// passes on the CPU), the cut into treelets on the top 12 bits (hlbvh.rs:383-404) and emit_lbvh for every treelet at once
// (hlbvh.rs:159-246): one work item per tree node, one launch per bit level going down, one per level coming back up for the
// interior boxes.  What stays on the host: the 12-bucket SAH over the <= 4096 treelet roots and the 4-wide collapse (pt_bvh.cpp).
//
// Everything is integer work or exact min / max / compare, apart from the centroid offset (one subtract, one IEEE divide) which is the
// same f32 arithmetic as the host's; a stable sort has exactly one result.  So the binary tree equals the host builder's node for node
// (node numbering differs -- it is link-walked by the collapse, never compared).  The one thing emit_lbvh does that is not done here
// is the centroid-median fallback below the last code bit (split_node, hlbvh.rs:102-157, it re-sorts items): a range that needs it
// raises `fallback` and the caller rebuilds on the host.  So do non-finite bounds, whose min / max are order dependent.
#include <sys/syscall.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pt_bvh.h"

namespace ptbvh {

namespace {

constexpr int kRadixBits = 6;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kSortTile = 1024;          // keys per one-wave workgroup and pass
constexpr uint32_t kTop12 = 0x3ffc0000u;

struct Work { uint32_t lo, hi; int32_t bit; int32_t node; };

// order-preserving float <-> uint map for atomicMin / atomicMax
__device__ inline uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__device__ inline float fmin_le(float a, float b) { return a <= b ? a : b; }
__device__ inline float fmax_ge(float a, float b) { return a >= b ? a : b; }

// raw[n][6] -> ib[n][6] (lo - eps, hi + eps), the scene bound of the inflated boxes, and a flag for non-finite input.
__global__ __launch_bounds__(256) void k_hl_prepare(const float* __restrict__ raw, uint32_t n, float* __restrict__ ib, uint32_t* gb, uint32_t* flags) {
    const float eps = 1.1920929e-7f * 2.0f;
    float lo[3] = {3.4028235e38f, 3.4028235e38f, 3.4028235e38f}, hi[3] = {-3.4028235e38f, -3.4028235e38f, -3.4028235e38f};
    bool bad = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        for (int a = 0; a < 3; a++) {
            const float l = raw[(size_t)i * 6 + a] - eps, h = raw[(size_t)i * 6 + 3 + a] + eps;
            ib[(size_t)i * 6 + a] = l;
            ib[(size_t)i * 6 + 3 + a] = h;
            bad |= !(fabsf(l) <= 3.4028235e38f) || !(fabsf(h) <= 3.4028235e38f);
            lo[a] = fmin_le(lo[a], l);
            hi[a] = fmax_ge(hi[a], h);
        }
    }
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    __shared__ float part[4][6];
    __shared__ int any_bad;
    if (threadIdx.x == 0) any_bad = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) { part[threadIdx.x >> 6][a] = lo[a]; part[threadIdx.x >> 6][3 + a] = hi[a]; }
    if (bad) any_bad = 1;
    __syncthreads();
    if (threadIdx.x < 6) {               // one atomic per bound component and workgroup
        float v = part[0][threadIdx.x];
        for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, part[w][threadIdx.x]) : fmaxf(v, part[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&gb[threadIdx.x], f2ord(v)); else atomicMax(&gb[threadIdx.x], f2ord(v));
    }
    if (threadIdx.x == 0 && any_bad) atomicOr(flags, 1u);
}

__device__ inline uint32_t spread3(uint32_t x) {          // left_shift3 (hlbvh.rs:23-40); 1024 is clamped to 1023
    if (x >= 1024u) x = 1023u;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(256) void k_hl_codes(const float* __restrict__ ib, uint32_t n, const uint32_t* __restrict__ gb, uint32_t* __restrict__ code, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t g[3];
    for (int a = 0; a < 3; a++) {
        const float blo = ord2f(gb[a]), bhi = ord2f(gb[3 + a]);
        const float c = (ib[(size_t)i * 6 + a] + ib[(size_t)i * 6 + 3 + a]) * 0.5f;
        float o = c - blo;                                  // Bounds3::offset (bounds3.rs)
        if (bhi > blo) o = o / (bhi - blo);
        const float cl = o < 0.0f ? 0.0f : (o > 1.0f ? 1.0f : o);
        const float f = ceilf(cl * 1024.0f);
        g[a] = f > 0.0f ? (uint32_t)f : 0u;                 // NaN -> 0
    }
    code[i] = (spread3(g[2]) << 2) | (spread3(g[1]) << 1) | spread3(g[0]);
    idx[i] = i;
}

// ---- stable LSD radix sort, one wave per tile ---------------------------------------------------
__global__ __launch_bounds__(64) void k_hl_hist(const uint32_t* __restrict__ key, uint32_t n, int shift, uint32_t* __restrict__ hist, uint32_t n_tiles) {
    __shared__ uint32_t cnt[kRadix];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortTile / 64; r++) {
        const uint32_t i = base + r * 64 + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(key[i] >> shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = cnt[threadIdx.x];
}

// In-place exclusive scan of m counters by one 1024-thread workgroup.
__global__ __launch_bounds__(1024) void k_hl_scan(uint32_t* __restrict__ v, uint32_t m) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < m; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t x = i < m ? v[i] : 0u;
        uint32_t s = x;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(s, off);
            if (lane >= off) s += y;
        }
        if (lane == 63) wave_sum[wave] = s;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < wave; w++) before += wave_sum[w];
        if (i < m) v[i] = before + s - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + s;
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_hl_scatter(const uint32_t* __restrict__ key_in, const uint32_t* __restrict__ val_in, uint32_t* __restrict__ key_out,
                                                   uint32_t* __restrict__ val_out, uint32_t n, int shift, const uint32_t* __restrict__ offs, uint32_t n_tiles) {
    __shared__ uint32_t next[kRadix];
    next[threadIdx.x] = offs[(size_t)threadIdx.x * n_tiles + blockIdx.x];
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    const uint64_t below = (1ull << threadIdx.x) - 1ull;
    for (int r = 0; r < kSortTile / 64; r++) {
        const uint32_t i = base + r * 64 + threadIdx.x;
        const bool live = i < n;
        const uint32_t k = live ? key_in[i] : 0u;
        const uint32_t d = (k >> shift) & (kRadix - 1);
        uint64_t same = __ballot(live);
        for (int b = 0; b < kRadixBits; b++) {
            const uint64_t m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t dst = 0;
        if (live) dst = next[d] + (uint32_t)__popcll(same & below);
        __syncthreads();
        if (live && (same >> threadIdx.x) <= 1ull) next[d] += (uint32_t)__popcll(same);      // the group's highest lane
        __syncthreads();
        if (live) { key_out[dst] = k; val_out[dst] = val_in[i]; }
    }
}

// ---- treelets ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hl_treelets(const uint32_t* __restrict__ code, uint32_t n, uint32_t* __restrict__ tl_start, uint32_t* __restrict__ tl_end) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = (code[i] & kTop12) >> 18;
    if (i == 0 || ((code[i - 1] & kTop12) >> 18) != v) tl_start[v] = i;
    if (i + 1 == n || ((code[i + 1] & kTop12) >> 18) != v) tl_end[v] = i + 1;
}

// counters[0] = nodes allocated, counters[1] = work items queued
__global__ __launch_bounds__(64) void k_hl_roots(const uint32_t* __restrict__ tl_start, const uint32_t* __restrict__ tl_end, Work* __restrict__ work, uint32_t* __restrict__ counters) {
    uint32_t t = 0;                      // one wave compacts the non-empty cells in cell (= Morton) order
    for (uint32_t base = 0; base < 4096u; base += 64) {
        const uint32_t v = base + threadIdx.x;
        const uint32_t end = tl_end[v];
        const uint64_t live = __ballot(end != 0);
        if (end != 0) {
            const uint32_t k = t + (uint32_t)__popcll(live & ((1ull << threadIdx.x) - 1ull));
            work[k] = Work{tl_start[v], end, 29 - 12, (int32_t)k};
        }
        t += (uint32_t)__popcll(live);
    }
    if (threadIdx.x == 0) { counters[0] = t; counters[1] = t; }
}

// emit_lbvh for the work items [begin, begin + count): a leaf, or an interior node whose two halves are queued for the next launch.
// A range that runs out of code bits with more than max_prims items goes on by centroid medians (split_node, hlbvh.rs:102-157; the
// host version is Hlbvh::median_split): Work::bit = -2 - dim marks such an item, its range is put in stable centroid order along `dim`
// by k_hl_median_sort right after this launch (the halves are cut in the middle whatever the order), the halves cycle the axis.
constexpr uint32_t kMedianMax = 4096;        // longest range the median fallback orders on the device
__global__ __launch_bounds__(256) void k_hl_level(Work* __restrict__ work, uint32_t begin, uint32_t count, const uint32_t* __restrict__ code, const uint32_t* __restrict__ idx,
                                                  const float* __restrict__ ib, uint32_t max_prims, LbvhNode* __restrict__ nodes, uint32_t node_cap, uint32_t work_cap,
                                                  uint32_t* __restrict__ counters, uint32_t* __restrict__ flags, uint32_t* __restrict__ med_list) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= count) return;
    const Work e = work[begin + w];
    int bit = e.bit;
    bool leaf = false, give_up = false, median = e.bit <= -2;
    int dim = median ? -2 - e.bit : 2;                   // emit_lbvh enters split_node with dim 2 (hlbvh.rs:190)
    if (median) leaf = e.hi - e.lo <= max_prims;
    else
        for (;;) {
            if (e.hi - e.lo <= max_prims) { leaf = true; break; }
            if (bit < 0) { median = true; break; }
            const uint32_t mask = 1u << bit;
            if ((code[e.lo] & mask) != (code[e.hi - 1] & mask)) break;
            bit--;
        }
    LbvhNode nd;
    nd.left = -1; nd.right = -1; nd.first = 0; nd.count = 0; nd.axis = 0;
    if (!leaf) {
        uint32_t b;
        if (median) {
            b = e.lo + (e.hi - e.lo) / 2u;
            if (e.hi - e.lo > kMedianMax) { atomicOr(flags, 2u); give_up = true; }      // longer than the sort kernel's LDS: the host rebuilds
        } else {
            const uint32_t mask = 1u << bit;
            uint32_t a = e.lo;
            b = e.hi - 1;
            const uint32_t first_bit = code[a] & mask;
            while (a + 1 != b) {
                const uint32_t m = (uint32_t)(((uint64_t)a + b) / 2);
                if ((code[m] & mask) == first_bit) a = m; else b = m;
            }
        }
        if (!give_up) {
            const uint32_t l = atomicAdd(&counters[0], 2u);
            const uint32_t q = atomicAdd(&counters[1], 2u);
            if (l + 2 > node_cap || q + 2 > work_cap) { atomicOr(flags, 4u); give_up = true; }
            else {
                const int child_bit = median ? -2 - (dim + 2) % 3 : bit - 1;
                nd.left = (int32_t)l; nd.right = (int32_t)l + 1; nd.axis = (uint8_t)(median ? dim : bit % 3);
                for (int i = 0; i < 3; i++) { nd.lo[i] = 0; nd.hi[i] = 0; }
                work[q] = Work{e.lo, b, child_bit, (int32_t)l};
                work[q + 1] = Work{b, e.hi, child_bit, (int32_t)l + 1};
                if (median) med_list[atomicAdd(&flags[1], 1u)] = begin + w;
            }
        }
    }
    if (leaf) {
        const float* p = ib + (size_t)idx[e.lo] * 6;
        for (int i = 0; i < 3; i++) { nd.lo[i] = p[i]; nd.hi[i] = p[3 + i]; }
        for (uint32_t k = e.lo + 1; k < e.hi; k++) {
            const float* q = ib + (size_t)idx[k] * 6;
            for (int i = 0; i < 3; i++) { nd.lo[i] = fmin_le(nd.lo[i], q[i]); nd.hi[i] = fmax_ge(nd.hi[i], q[3 + i]); }
        }
        nd.first = e.lo; nd.count = e.hi - e.lo;
    }
    if (give_up) {                      // the result is discarded; keep the node well formed for the launches still in flight
        for (int i = 0; i < 3; i++) { nd.lo[i] = 0; nd.hi[i] = 0; }
        nd.left = -1; nd.right = -1; nd.first = e.lo; nd.count = 1; nd.axis = 0;
    }
    nodes[e.node] = nd;
}
// The ranges k_hl_level has just split by medians, in stable centroid order along the node's axis: one workgroup per range ranks its
// items by (centroid, position) -- what std::stable_sort leaves -- and rewrites that part of the primitive order.
__global__ __launch_bounds__(256) void k_hl_median_sort(const Work* __restrict__ work, const uint32_t* __restrict__ med_list, const LbvhNode* __restrict__ nodes,
                                                        const float* __restrict__ ib, uint32_t* idx) {
    __shared__ float s_key[kMedianMax];
    __shared__ uint32_t s_idx[kMedianMax];
    const Work e = work[med_list[blockIdx.x]];
    const uint32_t n = e.hi - e.lo;
    const int dim = (int)nodes[e.node].axis;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t p = idx[e.lo + i];
        s_idx[i] = p;
        s_key[i] = (ib[(size_t)p * 6 + dim] + ib[(size_t)p * 6 + 3 + dim]) * 0.5f;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float k = s_key[i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < n; j++) r += (s_key[j] < k || (j < i && !(k < s_key[j]))) ? 1u : 0u;
        idx[e.lo + r] = s_idx[i];
    }
}

// Interior boxes of one level from its children (finished by the launches for the deeper levels).
__global__ __launch_bounds__(256) void k_hl_up(const Work* __restrict__ work, uint32_t begin, uint32_t count, LbvhNode* __restrict__ nodes) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= count) return;
    LbvhNode& nd = nodes[work[begin + w].node];
    if (nd.count > 0) return;
    const LbvhNode& l = nodes[nd.left];
    const LbvhNode& r = nodes[nd.right];
    for (int i = 0; i < 3; i++) { nd.lo[i] = fmin_le(l.lo[i], r.lo[i]); nd.hi[i] = fmax_ge(l.hi[i], r.hi[i]); }
}

// Device scratch for one build.  The blocks come out of an arena the calling thread keeps per device (one hipMalloc, reused by every
// later build: the 14 hipMalloc / hipFree pairs cost 7 ms per build, more than the build's kernels); what does not fit -- the first
// build, or a bigger scene -- is allocated the slow way and the arena is regrown to the high-water mark afterwards.
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
    void end() {                                                   // every block handed out has been returned (Scratch destructors ran)
        if (wanted > cap && wanted <= ((size_t)4 << 30)) {
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

}  // namespace

#define HL_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { if (err) *err = e_; return e_ == hipErrorOutOfMemory ? 1 : -1; } } while (0)

// raw_dev != nullptr: the bounds are on the device already and the results stay there -- the primitive order in order_dev[n], the binary
// nodes in nodes_dev (room for node_cap_dev of them; the first *n_treelets are the treelet roots in Morton order), *n_nodes_out of them
// written; only the roots come back to the host (roots_host), for the upper SAH.
static int lbvh_core(hipStream_t st, const float* raw_bounds, const float* raw_dev, uint32_t n, uint32_t max_prims, NoInitVec<uint32_t>* order, NoInitVec<LbvhNode>* nodes,
                     uint32_t* n_treelets, hipError_t* err, uint32_t* order_dev, LbvhNode* nodes_dev, uint32_t node_cap_dev, uint32_t* n_nodes_out, std::vector<LbvhNode>* roots_host) {
    ArenaScope arena_scope;          // declared before every Scratch: destroyed after them
    const bool trace = std::getenv("PBRTGPU_BUILD_TRACE") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (err) *err = hipSuccess;
    if (n < 2 || n > (1u << 30)) return 1;
    const uint32_t n_tiles = (n + kSortTile - 1) / kSortTile;
    const uint32_t node_cap = 2u * n, work_cap = 2u * n;
    Scratch d_raw, d_ib, d_gb, d_key[2], d_val[2], d_hist, d_tl, d_work, d_nodes, d_small;
    if (raw_dev && node_cap_dev < node_cap) return 1;
    if (!raw_dev) HL_TRY(d_raw.alloc((size_t)n * 24));
    HL_TRY(d_ib.alloc((size_t)n * 24));
    HL_TRY(d_small.alloc(64));                       // gb[6], flags, counters[2]
    for (int i = 0; i < 2; i++) { HL_TRY(d_key[i].alloc((size_t)n * 4)); HL_TRY(d_val[i].alloc((size_t)n * 4)); }
    HL_TRY(d_hist.alloc((size_t)kRadix * n_tiles * 4));
    HL_TRY(d_tl.alloc(2 * 4096 * 4));
    HL_TRY(d_work.alloc((size_t)work_cap * sizeof(Work)));
    if (!raw_dev) HL_TRY(d_nodes.alloc((size_t)node_cap * sizeof(LbvhNode)));
    LbvhNode* const bnodes = raw_dev ? nodes_dev : d_nodes.as<LbvhNode>();
    Scratch d_med;
    HL_TRY(d_med.alloc(((size_t)n / 2 + 2) * 4));          // work items split by medians in one level (each holds more than max_prims >= 1 items)
    uint32_t* gb = d_small.as<uint32_t>();
    uint32_t* flags = gb + 6;
    uint32_t* counters = gb + 8;
    const double t1 = now();
    const uint32_t init[10] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    HL_TRY(hipMemcpyAsync(gb, init, sizeof(init), hipMemcpyHostToDevice, st));
    if (!raw_dev) HL_TRY(hipMemcpyAsync(d_raw.p, raw_bounds, (size_t)n * 24, hipMemcpyHostToDevice, st));
    HL_TRY(hipMemsetAsync(d_tl.p, 0, 2 * 4096 * 4, st));
    const uint32_t blocks = (n + 255) / 256;
    k_hl_prepare<<<blocks < 1024u ? blocks : 1024u, 256, 0, st>>>(raw_dev ? raw_dev : d_raw.as<float>(), n, d_ib.as<float>(), gb, flags);
    k_hl_codes<<<blocks, 256, 0, st>>>(d_ib.as<float>(), n, gb, d_key[0].as<uint32_t>(), d_val[0].as<uint32_t>());
    int cur = 0;
    for (int pass = 0; pass < 30 / kRadixBits; pass++) {
        const int shift = pass * kRadixBits;
        k_hl_hist<<<n_tiles, 64, 0, st>>>(d_key[cur].as<uint32_t>(), n, shift, d_hist.as<uint32_t>(), n_tiles);
        k_hl_scan<<<1, 1024, 0, st>>>(d_hist.as<uint32_t>(), (uint32_t)kRadix * n_tiles);
        k_hl_scatter<<<n_tiles, 64, 0, st>>>(d_key[cur].as<uint32_t>(), d_val[cur].as<uint32_t>(), d_key[cur ^ 1].as<uint32_t>(), d_val[cur ^ 1].as<uint32_t>(), n, shift,
                                            d_hist.as<uint32_t>(), n_tiles);
        cur ^= 1;
    }
    const uint32_t* code = d_key[cur].as<uint32_t>();
    uint32_t* idx = d_val[cur].as<uint32_t>();
    uint32_t* tl_start = d_tl.as<uint32_t>();
    uint32_t* tl_end = tl_start + 4096;
    k_hl_treelets<<<blocks, 256, 0, st>>>(code, n, tl_start, tl_end);
    k_hl_roots<<<1, 64, 0, st>>>(tl_start, tl_end, d_work.as<Work>(), counters);
    HL_TRY(hipGetLastError());
    uint32_t host_small[4];
    HL_TRY(hipMemcpyAsync(host_small, flags, 16, hipMemcpyDeviceToHost, st));
    HL_TRY(hipStreamSynchronize(st));
    if (host_small[0] != 0) return 1;
    const uint32_t n_roots = host_small[3];
    std::vector<uint32_t> level_begin, level_count;
    uint32_t begin = 0, count = n_roots;
    while (count > 0) {               // bits 17..0 and the leaves below: at most 19 levels, plus what the median fallback adds
        if (level_begin.size() > 96) return -1;
        level_begin.push_back(begin); level_count.push_back(count);
        HL_TRY(hipMemsetAsync(flags + 1, 0, 4, st));            // this level's median splits
        k_hl_level<<<(count + 255) / 256, 256, 0, st>>>(d_work.as<Work>(), begin, count, code, idx, d_ib.as<float>(), max_prims, bnodes, node_cap, work_cap,
                                                        counters, flags, d_med.as<uint32_t>());
        HL_TRY(hipGetLastError());
        HL_TRY(hipMemcpyAsync(host_small, flags, 16, hipMemcpyDeviceToHost, st));
        HL_TRY(hipStreamSynchronize(st));
        if (host_small[0] != 0) return 1;
        if (host_small[1] != 0) {         // their ranges into centroid order before the next level reads them
            k_hl_median_sort<<<host_small[1], 256, 0, st>>>(d_work.as<Work>(), d_med.as<uint32_t>(), bnodes, d_ib.as<float>(), idx);
            HL_TRY(hipGetLastError());
        }
        const uint32_t total = host_small[3];
        begin += count;
        count = total - begin;
    }
    for (size_t l = level_begin.size(); l-- > 0;)
        k_hl_up<<<(level_count[l] + 255) / 256, 256, 0, st>>>(d_work.as<Work>(), level_begin[l], level_count[l], bnodes);
    HL_TRY(hipGetLastError());
    if (trace) (void)hipStreamSynchronize(st);
    const double t2 = now();
    const uint32_t n_nodes = host_small[2];
    if (raw_dev) {
        roots_host->resize(n_roots);
        HL_TRY(hipMemcpyAsync(order_dev, idx, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        HL_TRY(hipMemcpyAsync(roots_host->data(), bnodes, (size_t)n_roots * sizeof(LbvhNode), hipMemcpyDeviceToHost, st));
        HL_TRY(hipStreamSynchronize(st));
        *n_treelets = n_roots;
        *n_nodes_out = n_nodes;
        if (trace) std::fprintf(stderr, "[lbvh] n=%u nodes=%u treelets=%u levels=%zu (kept on the device): alloc %.2f kernels %.2f ms\n", n, n_nodes, n_roots, level_begin.size(), t1 - t0, now() - t1);
        return 0;
    }
    order->resize(n);
    nodes->resize(n_nodes);
    HL_TRY(hipMemcpyAsync(order->data(), idx, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HL_TRY(hipMemcpyAsync(nodes->data(), bnodes, (size_t)n_nodes * sizeof(LbvhNode), hipMemcpyDeviceToHost, st));
    HL_TRY(hipStreamSynchronize(st));
    *n_treelets = n_roots;
    if (trace) std::fprintf(stderr, "[lbvh] n=%u nodes=%u treelets=%u levels=%zu: alloc %.2f upload+kernels %.2f readback %.2f ms\n", n, n_nodes, n_roots, level_begin.size(),
                            t1 - t0, t2 - t1, now() - t2);
    return 0;
}

int device_lbvh(hipStream_t st, const float* raw_bounds, uint32_t n, uint32_t max_prims, NoInitVec<uint32_t>* order, NoInitVec<LbvhNode>* nodes,
                uint32_t* n_treelets, hipError_t* err) {
    return lbvh_core(st, raw_bounds, nullptr, n, max_prims, order, nodes, n_treelets, err, nullptr, nullptr, 0, nullptr, nullptr);
}
int device_lbvh_keep(hipStream_t st, const float* raw_dev, uint32_t n, uint32_t max_prims, uint32_t* order_dev, LbvhNode* nodes_dev, uint32_t node_cap, uint32_t* n_nodes,
                     uint32_t* n_treelets, std::vector<LbvhNode>* roots_host, hipError_t* err) {
    return lbvh_core(st, nullptr, raw_dev, n, max_prims, nullptr, nullptr, n_treelets, err, order_dev, nodes_dev, node_cap, n_nodes, roots_host);
}

}  // namespace ptbvh
