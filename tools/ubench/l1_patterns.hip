// Micro-benchmark: what the CU's vector L1 charges for the access patterns of the traversal, and what the alternatives would cost.
//   node7            every lane reads seven 16-byte rows of its own random 128-byte node line (the lean node visit of k_trace)
//   node7 + LDS top  the same, but a share of the visits (those whose node index falls in the first TOP nodes, as the top levels of
//                    a breadth-first numbered tree would) read the node from an LDS copy instead; blocks of 1024 lanes, one copy per block
//   tri private      every lane reads one random 48-byte record (3 x 16 B)
//   tri leaf4        four adjacent lanes read the four consecutive records of one random 4-triangle leaf, each lane its own record
//                    (what a pooled leaf round does today)
//   tri leaf4 coal   the same 192 bytes, but load j of lane k takes chunk 4 j + k: the four lanes cover 64 contiguous bytes per load
// build: hipcc --offload-arch=gfx950 -O3 -o l1_patterns l1_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 256
#define TOP 341

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// share16: sixteenths of the visits that go to the LDS copy
template <int BLOCK, int SHARE16>
__global__ void __launch_bounds__(BLOCK) k_node7(const char* nodes, uint32_t n_nodes, float* out, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) char s_top[SHARE16 > 0 ? TOP * 128 : 16];
    if constexpr (SHARE16 > 0) {
        for (uint32_t k = threadIdx.x; k < TOP * 8; k += BLOCK) reinterpret_cast<float4*>(s_top)[k] = reinterpret_cast<const float4*>(nodes)[k];
        __syncthreads();
    }
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t sx = (gid & 1) ? 48u : 0u, sy = (gid & 2) ? 48u : 0u, sz = (gid & 4) ? 48u : 0u;
    float acc = 0.0f;
    uint32_t idx = __umulhi(hash32(gid ^ seed), n_nodes);
    for (int it = 0; it < ITER; it++) {
        float4 nx, fx, ny, fy, nz, fz; uint4 ch;
        bool in_lds = false;
        if constexpr (SHARE16 > 0) in_lds = (hash32(idx + 77u * it) & 15u) < (uint32_t)SHARE16;
        if (in_lds) {
            const char* p = s_top + ((idx % TOP) << 7);
            nx = *(const float4*)(p + sx); fx = *(const float4*)(p + 48u - sx);
            ny = *(const float4*)(p + 16u + sy); fy = *(const float4*)(p + 64u - sy);
            nz = *(const float4*)(p + 32u + sz); fz = *(const float4*)(p + 80u - sz);
            ch = *(const uint4*)(p + 96u);
        } else {
            const char* p = nodes + ((size_t)idx << 7);
            nx = *(const float4*)(p + sx); fx = *(const float4*)(p + 48u - sx);
            ny = *(const float4*)(p + 16u + sy); fy = *(const float4*)(p + 64u - sy);
            nz = *(const float4*)(p + 32u + sz); fz = *(const float4*)(p + 80u - sz);
            ch = *(const uint4*)(p + 96u);
        }
        acc += nx.x + fx.y + ny.z + fy.w + nz.x + fz.y;
        idx = __umulhi(hash32(idx + ch.x + it), n_nodes);
    }
    if (acc == 1234.5f) out[gid] = acc;
}

// MODE 0: private random record; 1: leaf of four, own record; 2: leaf of four, 64 contiguous bytes per load
template <int MODE>
__global__ void __launch_bounds__(256) k_tri(const char* recs, uint32_t n_recs, float* out, uint32_t seed) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, k = threadIdx.x & 3u;
    float acc = 0.0f;
    uint32_t idx = __umulhi(hash32((MODE == 0 ? gid : gid >> 2) ^ seed), n_recs - 4u);
    for (int it = 0; it < ITER; it++) {
        float4 a, b, c;
        if constexpr (MODE == 0) {
            const char* p = recs + (size_t)idx * 48u;
            a = *(const float4*)p; b = *(const float4*)(p + 16); c = *(const float4*)(p + 32);
        } else if constexpr (MODE == 1) {
            const char* p = recs + (size_t)(idx + k) * 48u;
            a = *(const float4*)p; b = *(const float4*)(p + 16); c = *(const float4*)(p + 32);
        } else {
            const char* p = recs + (size_t)idx * 48u + k * 16u;
            a = *(const float4*)p; b = *(const float4*)(p + 64); c = *(const float4*)(p + 128);
        }
        acc += a.x + b.y + c.z;
        uint32_t nxt = hash32(idx + __float_as_uint(c.w) * 0u + it);
        if constexpr (MODE != 0) nxt = (uint32_t)__builtin_amdgcn_mov_dpp((int)nxt, 0x00, 0xf, 0xf, true);     // quad_perm [0,0,0,0]: the leaf is the group's
        idx = __umulhi(nxt, n_recs - 4u);
    }
    if (acc == 1234.5f) out[gid] = acc;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    double mb = argc > 1 ? atof(argv[1]) : 21.0;
    uint32_t n_nodes = (uint32_t)(mb * 1e6 / 128);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    char* nodes; float* out;
    (void)hipMalloc(&nodes, (size_t)n_nodes * 128);
    (void)hipMalloc(&out, (size_t)cus * 16 * 64 * 4);
    (void)hipMemset(nodes, 0, (size_t)n_nodes * 128);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::printf("table %.1f MB, %d CUs, %d iterations per lane\n", mb, cus, ITER);
    auto run = [&](const char* name, auto launch, double lanes, double req_per_iter) {
        launch(1u);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        launch(2u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double its = lanes * ITER;
        std::printf("%-44s %8.3f ms  %7.1f G lane-iterations/s  %7.1f G lane requests/s  (%s)\n", name, ms, its / ms / 1e6, its * req_per_iter / ms / 1e6, hipGetErrorString(hipGetLastError()));
    };
    const double lanes = (double)cus * 1024;
    run("node7, blocks of 256", [&](uint32_t s) { hipLaunchKernelGGL((k_node7<256, 0>), dim3(cus * 4), dim3(256), 0, 0, nodes, n_nodes, out, s); }, lanes, 7);
    run("node7, blocks of 1024", [&](uint32_t s) { hipLaunchKernelGGL((k_node7<1024, 0>), dim3(cus), dim3(1024), 0, 0, nodes, n_nodes, out, s); }, lanes, 7);
    run("node7, 2/16 of visits from LDS", [&](uint32_t s) { hipLaunchKernelGGL((k_node7<1024, 2>), dim3(cus), dim3(1024), 0, 0, nodes, n_nodes, out, s); }, lanes, 7);
    run("node7, 3/16 of visits from LDS", [&](uint32_t s) { hipLaunchKernelGGL((k_node7<1024, 3>), dim3(cus), dim3(1024), 0, 0, nodes, n_nodes, out, s); }, lanes, 7);
    run("node7, 4/16 of visits from LDS", [&](uint32_t s) { hipLaunchKernelGGL((k_node7<1024, 4>), dim3(cus), dim3(1024), 0, 0, nodes, n_nodes, out, s); }, lanes, 7);
    run("node7, 8/16 of visits from LDS", [&](uint32_t s) { hipLaunchKernelGGL((k_node7<1024, 8>), dim3(cus), dim3(1024), 0, 0, nodes, n_nodes, out, s); }, lanes, 7);
    run("node7, all visits from LDS", [&](uint32_t s) { hipLaunchKernelGGL((k_node7<1024, 16>), dim3(cus), dim3(1024), 0, 0, nodes, n_nodes, out, s); }, lanes, 7);
    const uint32_t n_recs = (uint32_t)(((size_t)n_nodes * 128) / 48);
    run("tri private (3 x 16 B of a random record)", [&](uint32_t s) { hipLaunchKernelGGL((k_tri<0>), dim3(cus * 4), dim3(256), 0, 0, nodes, n_recs, out, s); }, lanes, 3);
    run("tri leaf4, own record", [&](uint32_t s) { hipLaunchKernelGGL((k_tri<1>), dim3(cus * 4), dim3(256), 0, 0, nodes, n_recs, out, s); }, lanes, 3);
    run("tri leaf4, 64 contiguous bytes per load", [&](uint32_t s) { hipLaunchKernelGGL((k_tri<2>), dim3(cus * 4), dim3(256), 0, 0, nodes, n_recs, out, s); }, lanes, 3);
    return 0;
}
