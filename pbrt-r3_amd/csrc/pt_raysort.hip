// pt_raysort.hip -- the shadow rays of a traversal launch, ordered by where they start.
//
// Continuation rays reach k_trace in path (pixel) order and keep whatever coherence the image has; the next-event rays of a bounce start
// at that bounce's hit points, which in a scene of small triangles are scattered through space, and all head for the lights.  Ordered by
// the Morton cell of their origin (PT_SORT_CELL_BITS per axis inside the world bound) and the octant of their direction, neighbouring lanes walk
// the same nodes and leaves: lanes that touch the same line share the L1's tag lookup, which is what bounds k_trace (DESIGN.md section
// 4).  Only the work list is permuted -- path state stays where it is, results are written per path, so nothing downstream can tell.
// The sort is rocprim's radix sort on the 3 x PT_SORT_CELL_BITS + 3 key bits with the path id as the value.  Why a library sort stays on
// this path although pt_hlbvh.hip holds a hand-written radix sort: rocprim is a header-only part of ROCm that compiles INTO this
// library (no run-time dependency), its onesweep kernels sort a bounce's 30-140 M (key, id) pairs in 0.5 % of the frame, and the in-tree
// sort is built for the builder's needs -- a stable 6-bit LSD pass whose digit offsets come from ONE 1024-thread scan over
// 64 x n / 1024 counters, 8.7 M of them for a 140 M-ray list and three passes of that per bounce -- where what this list needs is a
// decoupled-lookback single-pass scan, i.e. rocprim's own design again.  Nothing about parity rides on it: only the work list is permuted.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>
#include "pt_device.h"
#include "pt_kernels.h"

size_t ptk_sort_rays_temp_bytes(uint32_t cap) {
    size_t bytes = 0;
    rocprim::double_buffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
    if (rocprim::radix_sort_pairs(nullptr, bytes, k, v, cap, 0, 3 * PT_SORT_CELL_BITS + 3, nullptr) != hipSuccess) return 0;
    return bytes;
}

// ids[0 .. n): path ids of the launch's shadow rays, keys[0 .. n) their sort keys (k_shade writes both, entry by entry).  On return *sorted
// points at the ordered list (either ids or ids_alt).
hipError_t ptk_sort_shadow_rays(hipStream_t st, uint32_t* ids, uint32_t* ids_alt, uint32_t* keys, uint32_t* keys_alt, void* temp, size_t temp_bytes, uint32_t n,
                                uint32_t** sorted) {
    *sorted = ids;
    if (n < 2) return hipSuccess;
    rocprim::double_buffer<uint32_t> k(keys, keys_alt), v(ids, ids_alt);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k, v, n, 0, 3 * PT_SORT_CELL_BITS + 3, st);
    if (e != hipSuccess) return e;
    *sorted = v.current();
    return hipGetLastError();
}

// The same sort with the input list left as it is: (keys, ids) -> (keys_out, ids_out).  For the continuation rays of a bounce, where the
// shading kernel may want the list in path order while the traversal kernel walks the ordered copy.
size_t ptk_sort_rays_keep_temp_bytes(uint32_t cap) {
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, cap, 0, 3 * PT_SORT_CELL_BITS + 3, nullptr) != hipSuccess)
        return 0;
    return bytes;
}
hipError_t ptk_sort_rays_keep(hipStream_t st, const uint32_t* ids, uint32_t* ids_out, const uint32_t* keys, uint32_t* keys_out, void* temp, size_t temp_bytes, uint32_t n) {
    if (n == 0) return hipSuccess;
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, ids, ids_out, n, 0, 3 * PT_SORT_CELL_BITS + 3, st);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}
