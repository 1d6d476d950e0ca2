// pt_bvh.cpp -- host-side BVH construction for the gfx950 traversal kernels.
//
// The tree topology must be the one pbrt-r3 builds (same leaves, same child order,
// same split axes), because the traversal order and hence every f32 result depends
// on it:  src/accelerators/bvh/build/{node,sah,middle,equal_counts,hlbvh}.rs and the
// two-level collapse of src/accelerators/bvh/accel/qbvh/qbvh_x86.rs:93-176.
// The implementation is our own: primitives are partitioned in place inside one
// array (so the final array order IS the leaf order), binary nodes go into a flat
// vector, large subtrees are built on worker threads, and the output is the 128-byte
// PtNode / 48-byte PtTri layout of pt_device.h with leaves folded into child refs.
#include "pt_bvh.h"
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>
#include <memory>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace ptbvh {

namespace {

struct Item {            // one primitive during construction (40 bytes, moved physically)
    float lo[3], hi[3], c[3];
    uint32_t prim;
};
typedef LbvhNode BNode;
struct Box {
    float lo[3], hi[3];
};
inline float fmin_le(float a, float b) { return a <= b ? a : b; }   // Bounds3::union's comparisons (bounds3.rs:40-46)
inline float fmax_ge(float a, float b) { return a >= b ? a : b; }
inline void box_empty(Box& b) {
    for (int i = 0; i < 3; i++) { b.lo[i] = std::numeric_limits<float>::max(); b.hi[i] = std::numeric_limits<float>::lowest(); }
}
inline void box_grow(Box& b, const float* lo, const float* hi) {
    for (int i = 0; i < 3; i++) { b.lo[i] = fmin_le(b.lo[i], lo[i]); b.hi[i] = fmax_ge(b.hi[i], hi[i]); }
}
inline void box_grow_pt(Box& b, const float* p) {
    for (int i = 0; i < 3; i++) { b.lo[i] = fmin_le(b.lo[i], p[i]); b.hi[i] = fmax_ge(b.hi[i], p[i]); }
}
inline float box_area(const Box& b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}
inline int box_max_extent(const Box& b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx > dy && dx > dz) return 0;
    if (dy > dz) return 1;
    return 2;
}
inline float box_offset(const Box& b, const float* p, int dim) {
    float o = p[dim] - b.lo[dim];
    if (b.hi[dim] > b.lo[dim]) o = o / (b.hi[dim] - b.lo[dim]);
    return o;
}

const int kBuckets = 12;
inline int bucket_index(const Box& cb, const float* c, int dim) {
    int b = (int)std::floor((float)kBuckets * box_offset(cb, c, dim));
    if (b > kBuckets - 1) b = kBuckets - 1;
    if (b < 0) b = 0;
    return b;
}

struct Builder {
    Item* items;
    Item* scratch;
    uint32_t max_prims;
    int method;
    size_t par_threshold;

    typedef NoInitVec<BNode> Tree;

    // Append a finished subtree (root at sub[0], links relative to sub) to dst; returns its new root index.
    static int32_t splice(Tree& dst, const Tree& sub) {
        int32_t base = (int32_t)dst.size();
        for (const BNode& n : sub) {
            BNode m = n;
            if (m.left >= 0) m.left += base;
            if (m.right >= 0) m.right += base;
            dst.push_back(m);
        }
        return base;
    }
    static int32_t push_leaf(Tree& t, size_t lo, size_t hi, const Box& b) {
        BNode n = BNode::blank();
        std::memcpy(n.lo, b.lo, 12); std::memcpy(n.hi, b.hi, 12);
        n.first = (uint32_t)lo; n.count = (uint32_t)(hi - lo);
        t.push_back(n);
        return (int32_t)t.size() - 1;
    }
    void stable_sort_axis(size_t lo, size_t hi, int dim) {
        std::stable_sort(items + lo, items + hi, [dim](const Item& a, const Item& b) { return a.c[dim] < b.c[dim]; });
    }

    // Builds the subtree over items[lo, hi) into t (appending) and returns its root index.
    // Large subtrees near the top hand their left half to a worker thread, which builds into
    // its own vector (no shared storage) that is spliced in afterwards.
    int32_t build(Tree& t, size_t lo, size_t hi, int depth) {
        size_t n = hi - lo;
        Box bounds;
        std::memcpy(bounds.lo, items[lo].lo, 12); std::memcpy(bounds.hi, items[lo].hi, 12);
        for (size_t i = lo + 1; i < hi; i++) box_grow(bounds, items[i].lo, items[i].hi);
        if (n <= max_prims) return push_leaf(t, lo, hi, bounds);
        Box cb;
        std::memcpy(cb.lo, items[lo].c, 12); std::memcpy(cb.hi, items[lo].c, 12);
        for (size_t i = lo + 1; i < hi; i++) box_grow_pt(cb, items[i].c);
        int dim = box_max_extent(cb);
        if (cb.lo[dim] == cb.hi[dim]) return push_leaf(t, lo, hi, bounds);

        size_t mid = 0;
        bool have_split = false;
        if (method == PT_SPLIT_MIDDLE) {
            float p_mid = (cb.lo[dim] + cb.hi[dim]) / 2.0f;
            stable_sort_axis(lo, hi, dim);
            size_t m = lo;
            while (m < hi && !(p_mid <= items[m].c[dim])) m++;
            if (m != lo && m != hi) { mid = m; have_split = true; }
        } else if (method == PT_SPLIT_SAH) {
            if (n == 1) return push_leaf(t, lo, hi, bounds);
            if (n == 2) {
                stable_sort_axis(lo, hi, dim);
                mid = lo + 1;
                have_split = true;
            } else {
                int count[kBuckets];
                Box bb[kBuckets];
                for (int i = 0; i < kBuckets; i++) { count[i] = 0; box_empty(bb[i]); }
                for (size_t i = lo; i < hi; i++) {
                    int b = bucket_index(cb, items[i].c, dim);
                    count[b]++;
                    box_grow(bb[b], items[i].lo, items[i].hi);
                }
                float cost[kBuckets - 1];
                const float total_area = box_area(bounds);
                for (int i = 0; i < kBuckets - 1; i++) {
                    Box b0 = bb[i], b1 = bb[i + 1];
                    int c0 = 0, c1 = 0;
                    for (int j = 0; j <= i; j++) { box_grow(b0, bb[j].lo, bb[j].hi); c0 += count[j]; }
                    for (int j = i + 1; j < kBuckets; j++) { box_grow(b1, bb[j].lo, bb[j].hi); c1 += count[j]; }
                    cost[i] = 1.0f + ((float)c0 * box_area(b0) + (float)c1 * box_area(b1)) / total_area;
                }
                float min_cost = cost[0];
                int min_bucket = 0;
                for (int i = 1; i < kBuckets - 1; i++)
                    if (cost[i] < min_cost) { min_cost = cost[i]; min_bucket = i; }
                if (n > max_prims || min_cost < (float)n) {
                    // stable partition: left part compacts in place, right part detours through scratch
                    size_t nl = 0, nr = 0;
                    for (size_t i = lo; i < hi; i++) {
                        if (bucket_index(cb, items[i].c, dim) <= min_bucket) items[lo + nl++] = items[i];
                        else scratch[lo + nr++] = items[i];
                    }
                    std::memcpy(items + lo + nl, scratch + lo, nr * sizeof(Item));
                    if (nl != 0 && nr != 0) { mid = lo + nl; have_split = true; }
                } else {
                    return push_leaf(t, lo, hi, bounds);
                }
            }
        }
        if (!have_split) {   // split_equal_counts, also the fallback of the other methods
            stable_sort_axis(lo, hi, dim);
            mid = lo + n / 2;
        }
        int32_t me = (int32_t)t.size();
        {
            BNode nd = BNode::blank();
            nd.axis = (uint8_t)dim;
            t.push_back(nd);
        }
        int32_t l, r;
        if (n >= par_threshold && depth < 6) {
            auto fut = std::async(std::launch::async, [this, lo, mid, depth]() {
                Tree sub;
                build(sub, lo, mid, depth + 1);
                return sub;
            });
            Tree right_sub;            // keep preorder numbering: left subtree first
            build(right_sub, mid, hi, depth + 1);
            Tree left_sub = fut.get();
            l = splice(t, left_sub);
            r = splice(t, right_sub);
        } else {
            l = build(t, lo, mid, depth + 1);
            r = build(t, mid, hi, depth + 1);
        }
        t[me].left = l; t[me].right = r;
        for (int i = 0; i < 3; i++) {   // init_interior: union of the two children
            t[me].lo[i] = fmin_le(t[l].lo[i], t[r].lo[i]);
            t[me].hi[i] = fmax_ge(t[l].hi[i], t[r].hi[i]);
        }
        return me;
    }
};

// ---- HLBVH (src/accelerators/bvh/build/hlbvh.rs) ----------------------------------------------
// Items are radix-sorted by the 30-bit Morton code of their centroid, cut into treelets on the top
// 12 bits, each treelet is split on successive code bits (falling back to centroid medians below the
// last bit), and the treelet roots are joined by a 12-bucket SAH.  Leaves only ever cover consecutive
// items, so -- as for the other builders -- the item array ends up in leaf order.
struct Hlbvh {
    Item* items;
    const uint32_t* code;      // Morton code of items[i] (valid until a median fallback reorders a range)
    uint32_t max_prims;
    NoInitVec<BNode>& t;
    bool failed = false;

    static uint32_t spread3(uint32_t x) {          // left_shift3 (hlbvh.rs:23-40); 1024 is clamped to 1023
        if (x >= 1024u) x = 1023u;
        x = (x | (x << 16)) & 0x030000ffu;
        x = (x | (x << 8)) & 0x0300f00fu;
        x = (x | (x << 4)) & 0x030c30c3u;
        x = (x | (x << 2)) & 0x09249249u;
        return x;
    }
    static uint32_t grid_coord(float offset) {     // clamp to [0,1], scale by 2^10, ceil, saturating cast (hlbvh.rs:42-47, :367-373)
        float c = offset < 0.0f ? 0.0f : (offset > 1.0f ? 1.0f : offset);
        float f = std::ceil(c * 1024.0f);
        return f > 0.0f ? (uint32_t)f : 0u;        // NaN -> 0
    }
    int32_t leaf(size_t lo, size_t hi) {
        Box b;
        std::memcpy(b.lo, items[lo].lo, 12); std::memcpy(b.hi, items[lo].hi, 12);
        for (size_t i = lo + 1; i < hi; i++) box_grow(b, items[i].lo, items[i].hi);
        return Builder::push_leaf(t, lo, hi, b);
    }
    int32_t join(int axis, int32_t l, int32_t r) {
        BNode nd = BNode::blank();
        nd.axis = (uint8_t)axis; nd.left = l; nd.right = r;
        for (int i = 0; i < 3; i++) { nd.lo[i] = fmin_le(t[l].lo[i], t[r].lo[i]); nd.hi[i] = fmax_ge(t[l].hi[i], t[r].hi[i]); }
        t.push_back(nd);
        return (int32_t)t.size() - 1;
    }
    int32_t median_split(size_t lo, size_t hi, int dim) {        // split_node (hlbvh.rs:102-157)
        if (hi - lo <= max_prims) return leaf(lo, hi);
        std::stable_sort(items + lo, items + hi, [dim](const Item& a, const Item& b) { return a.c[dim] < b.c[dim]; });
        size_t mid = lo + (hi - lo) / 2;
        int next = (dim + 2) % 3;
        int32_t l = median_split(lo, mid, next);
        int32_t r = median_split(mid, hi, next);
        return join(dim, l, r);
    }
    int32_t emit(size_t lo, size_t hi, int bit) {                 // emit_lbvh (hlbvh.rs:159-246)
        for (;;) {
            if (hi - lo <= max_prims) return leaf(lo, hi);
            if (bit < 0) return median_split(lo, hi, 2);
            const uint32_t mask = 1u << bit;
            if ((code[lo] & mask) != (code[hi - 1] & mask)) break;
            bit--;
        }
        const uint32_t mask = 1u << bit;
        size_t a = lo, b = hi - 1;                 // first item whose bit differs from the range's first item
        while (a + 1 != b) {
            size_t m = (a + b) / 2;
            if ((code[a] & mask) == (code[m] & mask)) a = m; else b = m;
        }
        int32_t l = emit(lo, b, bit - 1);
        int32_t r = emit(b, hi, bit - 1);
        return join(bit % 3, l, r);
    }
    static size_t sah_bucket(const BNode& n, const Box& cb, int dim) {
        float c = (n.lo[dim] + n.hi[dim]) * 0.5f;
        float f = (float)kBuckets * ((c - cb.lo[dim]) / (cb.hi[dim] - cb.lo[dim]));
        size_t b = f > 0.0f ? (f >= 1.8446744e19f ? ~(size_t)0 : (size_t)f) : 0;    // Rust's saturating `as usize`, NaN -> 0
        return b < (size_t)kBuckets - 1 ? b : (size_t)kBuckets - 1;
    }
    // build_upper_sah (hlbvh.rs:254-352), including its own cost bookkeeping: the left box includes bucket i
    // but the left count does not, the right count starts at one, and the traversal constant is 1/8.
    int32_t upper(const std::vector<int32_t>& roots) {
        if (roots.empty()) { failed = true; return -1; }          // the reference indexes an empty slice here
        if (roots.size() == 1) return roots[0];
        Box bounds, cb;
        box_empty(bounds);
        for (int32_t r : roots) box_grow(bounds, t[r].lo, t[r].hi);
        bool first = true;
        for (int32_t r : roots) {
            float c[3];
            for (int i = 0; i < 3; i++) c[i] = (t[r].lo[i] + t[r].hi[i]) * 0.5f;
            if (first) { std::memcpy(cb.lo, c, 12); std::memcpy(cb.hi, c, 12); first = false; }
            else box_grow_pt(cb, c);
        }
        const int dim = box_max_extent(cb);
        uint32_t count[kBuckets];
        Box bb[kBuckets];
        for (int i = 0; i < kBuckets; i++) { count[i] = 0; box_empty(bb[i]); }
        for (int32_t r : roots) {
            size_t b = sah_bucket(t[r], cb, dim);
            if (count[b]++ == 0) { std::memcpy(bb[b].lo, t[r].lo, 12); std::memcpy(bb[b].hi, t[r].hi, 12); }
            else box_grow(bb[b], t[r].lo, t[r].hi);
        }
        const float total = box_area(bounds);
        float best = 0.0f;
        size_t best_bucket = 0;
        for (int i = 0; i < kBuckets - 1; i++) {
            Box b0 = bb[i], b1 = bb[i];
            uint32_t c0 = 0, c1 = 1;
            for (int j = 0; j < i; j++) { box_grow(b0, bb[j].lo, bb[j].hi); c0 += count[j]; }
            for (int j = i + 1; j < kBuckets; j++) { box_grow(b1, bb[j].lo, bb[j].hi); c1 += count[j]; }
            float cost = 0.125f + (((float)c0 * box_area(b0) + (float)c1 * box_area(b1)) / total);
            if (i == 0 || cost < best) { best = cost; best_bucket = (size_t)i; }
        }
        std::vector<int32_t> left, right;
        for (int32_t r : roots) (sah_bucket(t[r], cb, dim) <= best_bucket ? left : right).push_back(r);
        int32_t l = upper(left);
        int32_t r = upper(right);
        if (l < 0 || r < 0) { failed = true; return -1; }
        return join(dim, l, r);
    }
};

// Morton codes of the item centroids, then a stable LSD radix sort of the items by code (3 passes of 10 bits).
void morton_sort(std::vector<Item>& items, std::vector<Item>& scratch, std::vector<uint32_t>& code) {
    const size_t n = items.size();
    Box bounds;
    std::memcpy(bounds.lo, items[0].lo, 12); std::memcpy(bounds.hi, items[0].hi, 12);
    for (size_t i = 1; i < n; i++) box_grow(bounds, items[i].lo, items[i].hi);
    code.resize(n);
    std::vector<uint32_t> code2(n);
    for (size_t i = 0; i < n; i++) {
        uint32_t g[3];
        for (int a = 0; a < 3; a++) g[a] = Hlbvh::grid_coord(box_offset(bounds, items[i].c, a));
        code[i] = (Hlbvh::spread3(g[2]) << 2) | (Hlbvh::spread3(g[1]) << 1) | Hlbvh::spread3(g[0]);
    }
    Item* src = items.data(); Item* dst = scratch.data();
    uint32_t* csrc = code.data(); uint32_t* cdst = code2.data();
    for (int pass = 0; pass < 3; pass++) {
        const int shift = 10 * pass;
        std::vector<size_t> start(1025, 0);
        for (size_t i = 0; i < n; i++) start[((csrc[i] >> shift) & 1023u) + 1]++;
        for (int b = 0; b < 1024; b++) start[b + 1] += start[b];
        for (size_t i = 0; i < n; i++) {
            size_t o = start[(csrc[i] >> shift) & 1023u]++;
            dst[o] = src[i];
            cdst[o] = csrc[i];
        }
        std::swap(src, dst);
        std::swap(csrc, cdst);
    }
    if (src != items.data()) {       // odd number of passes: the result sits in scratch
        std::memcpy(items.data(), src, n * sizeof(Item));
        std::memcpy(code.data(), csrc, n * 4);
    }
}

inline uint32_t leaf_reference(uint32_t first, uint32_t count) {
    return PT_LEAF_BIT | ((std::min(count, 8u) - 1u) << PT_LEAF_COUNT_SHIFT) | first;
}

struct Collapser {
    const NoInitVec<BNode>& b;
    Result& out;
    uint32_t max_depth4 = 0;

    uint32_t leaf_ref(const BNode& n) const { return leaf_reference(n.first, n.count); }

    // the (up to four) binary nodes that fill the slots of the 4-wide node made from binary node bi (a leaf child fills slot 0 / 2 and
    // leaves slot 1 / 3 empty, exactly like flatten_qbvh_tree)
    void slots_of(int32_t bi, int32_t slot_idx[4]) const {
        const BNode& n = b[bi];
        const BNode& c0 = b[n.left];
        const BNode& c1 = b[n.right];
        slot_idx[0] = slot_idx[1] = slot_idx[2] = slot_idx[3] = -1;
        if (c0.count > 0) slot_idx[0] = n.left; else { slot_idx[0] = c0.left; slot_idx[1] = c0.right; }
        if (c1.count > 0) slot_idx[2] = n.right; else { slot_idx[2] = c1.left; slot_idx[3] = c1.right; }
    }
    // 4-wide nodes in the subtree of binary node bi (itself included)
    uint32_t count4(int32_t bi) const {
        int32_t si[4];
        slots_of(bi, si);
        uint32_t c = 1;
        for (int k = 0; k < 4; k++) if (si[k] >= 0 && b[si[k]].count == 0) c += count4(si[k]);
        return c;
    }
    // One 4-wide node from a binary node and its two children, numbered depth-first: the node takes index `next`, its subtrees follow.
    // Subtrees listed in `cut` (binary node -> first index, in depth-first order) are skipped: somebody else emits them.
    uint32_t emit(int32_t bi, uint32_t depth, uint32_t& next, uint32_t& deepest, const std::vector<std::pair<int32_t, uint32_t>>* cut, size_t* cut_pos) {
        if (depth > deepest) deepest = depth;
        const BNode& n = b[bi];
        const uint32_t me = next++;
        PtNode nd;
        std::memset(&nd, 0, sizeof(nd));
        const BNode& c0 = b[n.left];
        const BNode& c1 = b[n.right];
        int32_t slot_idx[4];
        slots_of(bi, slot_idx);
        for (int k = 0; k < 4; k++) {
            if (slot_idx[k] < 0) { nd.child[k] = PT_EMPTY_REF; continue; }   // box stays all-zero as in the reference
            const BNode& sl = b[slot_idx[k]];
            for (int a = 0; a < 3; a++) { nd.bmin[a][k] = sl.lo[a]; nd.bmax[a][k] = sl.hi[a]; }
            if (sl.count > 0) nd.child[k] = leaf_ref(sl);
            else if (cut && *cut_pos < cut->size() && (*cut)[*cut_pos].first == slot_idx[k]) { nd.child[k] = (*cut)[*cut_pos].second; next = (*cut)[*cut_pos].second + cut_size[*cut_pos]; (*cut_pos)++; }
            else nd.child[k] = emit(slot_idx[k], depth + 1, next, deepest, cut, cut_pos);
        }
        nd.axes = (uint32_t)n.axis | ((uint32_t)c0.axis << 2) | ((uint32_t)c1.axis << 4);
        for (int k = 0; k < 4; k++)
            if (nd.child[k] != PT_EMPTY_REF) nd.axes |= 1u << (8 + k);
        out.nodes[me] = nd;
        return me;
    }
    std::vector<uint32_t> cut_size;      // 4-wide nodes under each entry of the cut

    // The whole tree.  The top levels are walked on this thread; the subtrees hanging below 4-wide depth kCutDepth are counted and then
    // emitted by the host's threads, each into the index range the depth-first numbering gives it.
    uint32_t run(int32_t root) {
        const uint32_t kCutDepth = 5;
        struct Sub { int32_t bi; uint32_t depth; };
        std::vector<Sub> subs;
        {   // the cut, in depth-first order
            struct It { int32_t bi; uint32_t depth; };
            std::vector<It> stack{{root, 1}};
            while (!stack.empty()) {
                const It it = stack.back();
                stack.pop_back();
                if (it.depth > kCutDepth) { subs.push_back({it.bi, it.depth}); continue; }
                int32_t si[4];
                slots_of(it.bi, si);
                for (int k = 3; k >= 0; k--) if (si[k] >= 0 && b[si[k]].count == 0) stack.push_back({si[k], it.depth + 1});
            }
        }
        cut_size.assign(subs.size(), 0);
        parallel_tasks(subs.size(), [&](size_t i) { cut_size[i] = count4(subs[i].bi); });
        // first indices: walk the top again, counting; then the real emission of the top with the cut in place
        std::vector<std::pair<int32_t, uint32_t>> cut(subs.size());
        {
            uint32_t next = 0;
            size_t pos = 0;
            struct It { int32_t bi; uint32_t depth; };
            std::vector<It> stack{{root, 1}};
            while (!stack.empty()) {
                const It it = stack.back();
                stack.pop_back();
                if (it.depth > kCutDepth) { cut[pos] = {it.bi, next}; next += cut_size[pos]; pos++; continue; }
                next++;
                int32_t si[4];
                slots_of(it.bi, si);
                for (int k = 3; k >= 0; k--) if (si[k] >= 0 && b[si[k]].count == 0) stack.push_back({si[k], it.depth + 1});
            }
            out.nodes.resize(next);
        }
        uint32_t next = 0, deepest = 0;
        size_t pos = 0;
        const uint32_t root_index = emit(root, 1, next, deepest, &cut, &pos);
        std::atomic<uint32_t> deep{deepest};
        parallel_tasks(subs.size(), [&](size_t i) {
            uint32_t nx = cut[i].second, dp = 0;
            emit(subs[i].bi, subs[i].depth, nx, dp, nullptr, nullptr);
            uint32_t cur = deep.load();
            while (dp > cur && !deep.compare_exchange_weak(cur, dp)) {}
        });
        max_depth4 = deep.load();
        return root_index;
    }
};

}  // namespace

int32_t hlbvh_upper_tree(NoInitVec<LbvhNode>& tree, uint32_t n_roots) {
    Hlbvh h{nullptr, nullptr, 0, tree};
    std::vector<int32_t> roots(n_roots);
    for (uint32_t k = 0; k < n_roots; k++) roots[k] = (int32_t)k;
    const int32_t root = h.upper(roots);
    return (h.failed || root < 0) ? -1 : root;
}

// Page-locked staging for the bounds a device build uploads (24 bytes per primitive): from pageable memory that copy is staged by
// the runtime at 2-3 GB/s -- 10 ms per million triangles, more than the build itself.  One buffer per calling thread, kept and regrown.
static float* pinned_floats(size_t n) {
    struct Holder {       // freed when the thread ends: `pbrt_gpu --gpus N` uploads from N short-lived threads
        float* buf = nullptr;
        size_t cap = 0;
        // (not for the process's first thread: it ends while the runtime is being torn down, and the process's memory goes with it)
        ~Holder() { if (buf && getpid() != (pid_t)syscall(SYS_gettid)) (void)hipHostFree(buf); }
    };
    static thread_local Holder h;
    if (n > h.cap) {
        if (h.buf) (void)hipHostFree(h.buf);
        h.buf = nullptr; h.cap = 0;
        void* q = nullptr;
        if (hipHostMalloc(&q, n * sizeof(float), hipHostMallocDefault) != hipSuccess) return nullptr;
        h.buf = (float*)q; h.cap = n;
    }
    return h.buf;
}

// Generic entry: any mix of primitives, each with its world bound and its ready-made 48-byte leaf record.
bool build_prims(const Prim* prims, uint32_t n_prims, int split_method, int max_node_prims, Result* out, DeviceBuild* dev) {
    const bool trace = std::getenv("PBRTGPU_BUILD_TRACE") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt[6] = {now(), 0, 0, 0, 0, 0};
    tt[1] = tt[2] = tt[0];
    out->nodes.clear(); out->tris.clear(); out->rec_of_prim.clear();
    out->n_leaves = 0; out->max_stack = 1;
    if (n_prims == 0) { out->root_ref = PT_EMPTY_REF; return true; }
    if (n_prims >= PT_LEAF_FIRST_MASK - 16u) return false;      // a leaf reference keeps 26 bits for the first record
    std::vector<Item> items, scratch;
    // order[r] = primitive stored r-th; the binary tree.  Kept by the calling thread between builds (cleared, capacity retained) while they
    // stay below 256 MB: mapping and unmapping 60 MB per million primitives on every build cost more than reading the tree back.
    static thread_local NoInitVec<uint32_t> order_keep;
    static thread_local NoInitVec<BNode> tree_keep;
    struct Trim { ~Trim() { if (tree_keep.capacity() * sizeof(BNode) > ((size_t)256 << 20)) { NoInitVec<BNode>().swap(tree_keep); NoInitVec<uint32_t>().swap(order_keep); } } } trim;
    NoInitVec<uint32_t>& order = order_keep;
    order.clear();
    const uint32_t n_items = n_prims;
    auto init_items = [&]() {
        items.resize(n_prims); scratch.resize(n_prims);
        const float eps = std::numeric_limits<float>::epsilon() * 2.0f;   // BOUND_EPS (build/node.rs:13)
        for (uint32_t pi = 0; pi < n_prims; pi++) {
            Item& it = items[pi];
            for (int i = 0; i < 3; i++) {
                it.lo[i] = prims[pi].lo[i] - eps;
                it.hi[i] = prims[pi].hi[i] + eps;
                it.c[i] = (it.lo[i] + it.hi[i]) * 0.5f;
            }
            it.prim = pi;
        }
    };
    Builder bld;
    bld.max_prims = (uint32_t)std::min(std::max(max_node_prims, 0), 255);
    bld.method = split_method;
    bld.par_threshold = 32768;
    NoInitVec<BNode>& tree = tree_keep;
    tree.clear();
    tree.reserve((size_t)n_items);
    int32_t root = 0;
    bool on_device = false;
    if (split_method == PT_SPLIT_HLBVH && dev && dev->mode != PT_BVH_BUILD_HOST && n_items >= 2 &&
        (dev->mode == PT_BVH_BUILD_DEVICE || n_items >= kDeviceMinPrims)) {
        float* raw = pinned_floats((size_t)n_items * 6);
        if (!raw) return false;
        parallel_for(n_items, [&](size_t p0, size_t p1) {
            for (size_t pi = p0; pi < p1; pi++) { std::memcpy(&raw[pi * 6], prims[pi].lo, 12); std::memcpy(&raw[pi * 6 + 3], prims[pi].hi, 12); }
        });
        uint32_t n_treelets = 0;
        tt[1] = now();
        const int rc = device_lbvh(dev->stream, raw, n_items, bld.max_prims, &order, &tree, &n_treelets, &dev->err);
        tt[2] = now();
        if (rc < 0) return false;
        if (rc == 0) {
            on_device = true;
            dev->used = true;
            Hlbvh h{nullptr, nullptr, bld.max_prims, tree};
            std::vector<int32_t> treelets(n_treelets);
            for (uint32_t k = 0; k < n_treelets; k++) treelets[k] = (int32_t)k;
            root = h.upper(treelets);
            if (h.failed || root < 0) return false;
        } else {
            tree.clear();
        }
    }
    if (split_method == PT_SPLIT_SAH && dev && dev->mode != PT_BVH_BUILD_HOST && n_items >= 2 && bld.max_prims >= 2 &&
        (dev->mode == PT_BVH_BUILD_DEVICE || n_items >= kDeviceMinPrims)) {
        float* raw = pinned_floats((size_t)n_items * 6);
        if (!raw) return false;
        parallel_for(n_items, [&](size_t p0, size_t p1) {
            for (size_t pi = p0; pi < p1; pi++) { std::memcpy(&raw[pi * 6], prims[pi].lo, 12); std::memcpy(&raw[pi * 6 + 3], prims[pi].hi, 12); }
        });
        tt[1] = now();
        const int rc = device_sah(dev->stream, raw, n_items, bld.max_prims, &order, &tree, &dev->err);
        tt[2] = now();
        if (rc < 0) return false;
        if (rc == 0) { on_device = true; dev->used = true; root = 0; }
        else { tree.clear(); order.clear(); }
    }
    if (!on_device) {
        init_items();
        bld.items = items.data();
        bld.scratch = scratch.data();
    }
    if (on_device) {
    } else if (split_method == PT_SPLIT_HLBVH) {
        std::vector<uint32_t> code;
        morton_sort(items, scratch, code);
        Hlbvh h{items.data(), code.data(), bld.max_prims, tree};
        std::vector<int32_t> treelets;
        const uint32_t top12 = 0x3ffc0000u;
        for (size_t start = 0, end = 1; end <= n_items; end++) {
            if (end == n_items || (code[start] & top12) != (code[end] & top12)) {
                treelets.push_back(h.emit(start, end, 29 - 12));
                start = end;
            }
        }
        root = h.upper(treelets);
        if (h.failed || root < 0) return false;
    } else {
        bld.build(tree, 0, n_items, 0);
    }

    tt[3] = now();
    std::memcpy(out->root_lo, tree[root].lo, 12);
    std::memcpy(out->root_hi, tree[root].hi, 12);
    // leaf records in final item order; PT_TRI_LAST closes each leaf
    out->tris.reserve((size_t)n_items + 1);
    out->tris.resize(n_items);
    out->rec_of_prim.resize(n_items);
    if (!on_device) {
        order.resize(n_items);
        for (uint32_t r = 0; r < n_items; r++) order[r] = items[r].prim;
    }
    parallel_for(n_items, [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; r++) {
            const uint32_t prim = order[r];
            PtTri& tr = out->tris[r];
            tr = prims[prim].rec;
            tr.prim = prim;
            tr.flags &= ~PT_TRI_LAST;
            tr.light1 = 0;            // filled by the caller once lights are numbered
            out->rec_of_prim[prim] = (uint32_t)r;
        }
    });
    {   // one zero pad record: the kernels fetch triangle records two at a time
        PtTri pad;
        std::memset(&pad, 0, sizeof(pad));
        pad.flags = PT_TRI_LAST;
        out->tris.push_back(pad);
    }
    {
        std::atomic<uint32_t> n_leaves{0}, max_leaf{0};
        parallel_for(tree.size(), [&](size_t k0, size_t k1) {
            uint32_t nl = 0, ml = 0;
            for (size_t k = k0; k < k1; k++) {
                const BNode& n = tree[k];
                if (n.count > 0) { out->tris[n.first + n.count - 1].flags |= PT_TRI_LAST; nl++; ml = std::max(ml, n.count); }
            }
            n_leaves += nl;
            uint32_t cur = max_leaf.load();
            while (ml > cur && !max_leaf.compare_exchange_weak(cur, ml)) {}
        });
        out->n_leaves = n_leaves.load();
        out->max_leaf = max_leaf.load();
    }
    if (tree[root].count > 0) {           // the whole scene is one leaf
        out->root_ref = leaf_reference(tree[root].first, tree[root].count);
        out->max_stack = 1;
        return true;
    }
    tt[4] = now();
    Collapser col{tree, *out};
    out->root_ref = col.run(root);
    out->max_stack = 3 * col.max_depth4 + 2;
    tt[5] = now();
    if (trace) std::fprintf(stderr, "[bvh] n=%u device=%d pack %.1f lbvh %.1f tree(host)/upper %.1f records %.1f collapse %.1f ms\n", n_items, (int)on_device,
                            tt[1] - tt[0], tt[2] - tt[1], tt[3] - (on_device ? tt[2] : tt[0]), tt[4] - tt[3], tt[5] - tt[4]);
    return true;
}


// Triangles (+ spheres spliced in before `before_triangle`) as one primitive list.
bool build(const float* P, const uint32_t* indices, const uint32_t* tri_flags, uint32_t n_tris, const SpherePrim* spheres, uint32_t n_spheres,
           int split_method, int max_node_prims, Result* out) {
    const uint64_t n_prims64 = (uint64_t)n_tris + n_spheres;
    if (n_prims64 >= PT_LEAF_FIRST_MASK - 16u) return false;
    std::vector<Prim> prims((size_t)n_prims64);
    uint32_t si = 0, pi = 0;
    for (uint32_t t = 0; t <= n_tris; t++) {
        while (si < n_spheres && spheres[si].before_triangle <= t) {
            if (spheres[si].before_triangle < t) return false;        // not ordered
            Prim& pr = prims[pi++];
            std::memcpy(pr.lo, spheres[si].lo, 12); std::memcpy(pr.hi, spheres[si].hi, 12);
            sphere_record(si, spheres[si].flags, &pr.rec);
            si++;
        }
        if (t == n_tris) break;
        triangle_prim(P, indices, t, tri_flags[t], &prims[pi++]);
    }
    if (si != n_spheres) return false;                                // before_triangle > n_tris
    return build_prims(prims.data(), (uint32_t)prims.size(), split_method, max_node_prims, out);
}
void sphere_record(uint32_t sphere_index, uint32_t flags, PtTri* rec) {
    std::memset(rec, 0, sizeof(*rec));
    std::memcpy(&rec->p0[0], &sphere_index, 4);
    rec->flags = (flags | PT_TRI_SPHERE) & ~PT_TRI_LAST;
}
void instance_record(uint32_t instance_index, PtTri* rec) {
    std::memset(rec, 0, sizeof(*rec));
    std::memcpy(&rec->p0[0], &instance_index, 4);
    rec->flags = PT_TRI_INSTANCE;
}
void triangle_prim(const float* P, const uint32_t* indices, uint32_t t, uint32_t flags, Prim* pr) {
    const float* p0 = P + 3 * (size_t)indices[3 * (size_t)t];
    const float* p1 = P + 3 * (size_t)indices[3 * (size_t)t + 1];
    const float* p2 = P + 3 * (size_t)indices[3 * (size_t)t + 2];
    for (int i = 0; i < 3; i++) {
        pr->lo[i] = std::fmin(std::fmin(p0[i], p1[i]), p2[i]);   // union3 (triangle.rs:189-200)
        pr->hi[i] = std::fmax(std::fmax(p0[i], p1[i]), p2[i]);
    }
    std::memset(&pr->rec, 0, sizeof(pr->rec));
    std::memcpy(pr->rec.p0, p0, 12); std::memcpy(pr->rec.p1, p1, 12); std::memcpy(pr->rec.p2, p2, 12);
    pr->rec.flags = flags & ~(PT_TRI_LAST | PT_TRI_SPHERE | PT_TRI_INSTANCE);
}

}  // namespace ptbvh
