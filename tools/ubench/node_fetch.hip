// Micro-benchmark: how fast can a wave fetch 64 different 128-byte BVH nodes (one per lane-ray)?
//   scatter   what k_trace's node visit does today: every lane issues 7 x dwordx4 + 1 x dword on its own 128-byte line, so one
//             wave-instruction touches up to 64 different lines (one L1 tag lookup per lane);
//   coop      the wave fetches the same 64 nodes cooperatively: instruction j, lane l loads 16-byte chunk (l & 7) of the node
//             of lane 8 j + (l >> 3) -- eight lanes cover one whole line -- stages it in LDS (ds_write_b128), and every owner
//             lane reads its node's rows back from LDS (ds_read_b128) with a chunk swizzle against bank conflicts;
//   coop_dma  the same with LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write.
// Node indices are pseudo-random over a table of `n_nodes` lines (argv[1] MB; default 21 MB = RT1M's node array).
// Reported: node visits / s over the chip, and the implied bytes / s.
// build: hipcc --offload-arch=gfx950 -O3 -o node_fetch node_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITER 512
#define BLOCK 256

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

extern "C" __global__ void __launch_bounds__(BLOCK) k_scatter(const char* nodes, uint32_t n_nodes, float* out, uint32_t seed) {
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t sx = (gid & 1) ? 48u : 0u, sy = (gid & 2) ? 48u : 0u, sz = (gid & 4) ? 48u : 0u;
    float acc = 0.0f;
    uint32_t idx = __umulhi(hash32(gid ^ seed), n_nodes);
    for (int it = 0; it < ITER; it++) {
        const uint32_t no = idx << 7;
        const float4 nx = *(const float4*)(nodes + (no + sx)), fx = *(const float4*)(nodes + (no + 48u - sx));
        const float4 ny = *(const float4*)(nodes + (no + 16u + sy)), fy = *(const float4*)(nodes + (no + 64u - sy));
        const float4 nz = *(const float4*)(nodes + (no + 32u + sz)), fz = *(const float4*)(nodes + (no + 80u - sz));
        const uint4 ch = *(const uint4*)(nodes + (no + 96u));
        const uint32_t lut = *(const uint32_t*)(nodes + (no + 116u));
        acc += nx.x + fx.y + ny.z + fy.w + nz.x + fz.y;
        idx = __umulhi(hash32(idx + ch.x + lut + it), n_nodes);       // the next node depends on this one, as in a traversal
    }
    if (acc == 1234.5f) out[gid] = acc;
}

// quad   no LDS: the four lanes of a quad fetch the node of each of their four rays in turn (two loads per ray: the four lanes
//        take four different 16-byte rows of the same line, near / far rows first as the OWNER's direction signs say), and
//        the owner picks the values up through DPP quad_perm operands
template <int K> __device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}
template <int K> __device__ __forceinline__ uint32_t quad_bcast_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, K * 0x55, 0xf, 0xf, true); }
extern "C" __global__ void __launch_bounds__(BLOCK) k_quad(const char* nodes, uint32_t n_nodes, float* out, uint32_t seed) {
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x, m = threadIdx.x & 3u;
    const uint32_t sgn = gid & 7u;
    float acc = 0.0f;
    uint32_t idx = __umulhi(hash32(gid ^ seed), n_nodes);
    for (int it = 0; it < ITER; it++) {
        float4 a[4], b[4];
#define QUAD_FETCH(K)                                                                                          \
        {                                                                                                      \
            const uint32_t oid = quad_bcast_u<K>(idx), os = quad_bcast_u<K>(sgn);                                \
            /* rows in the owner's near/far order: lane 0 near x, 1 near y, 2 near z, 3 far x | 0 far y, 1 far z, 2 children, 3 tail */ \
            const uint32_t sx = (os & 1u) ? 3u : 0u, sy = (os & 2u) ? 3u : 0u, sz = (os & 4u) ? 3u : 0u;        \
            const uint32_t c0 = m == 0 ? sx : (m == 1 ? 1u + sy : (m == 2 ? 2u + sz : 3u - sx));               \
            const uint32_t c1 = m == 0 ? 4u - sy : (m == 1 ? 5u - sz : (m == 2 ? 6u : 7u));                     \
            a[K] = *(const float4*)(nodes + ((oid << 7) + (c0 << 4)));                                          \
            b[K] = *(const float4*)(nodes + ((oid << 7) + (c1 << 4)));                                          \
        }
        QUAD_FETCH(0) QUAD_FETCH(1) QUAD_FETCH(2) QUAD_FETCH(3)
#undef QUAD_FETCH
        // owner m consumes a[m] / b[m] of its mates: rows arrive through DPP operands (modelled as broadcasts + select)
        float nxv, fyv; uint32_t chx, lut;
#define QUAD_TAKE(K)                                                                                           \
        {                                                                                                      \
            const float t0 = quad_bcast<0>(a[K].x) + quad_bcast<1>(a[K].y) + quad_bcast<2>(a[K].z) + quad_bcast<3>(a[K].w) + quad_bcast<0>(b[K].x) + quad_bcast<1>(b[K].y); \
            const uint32_t t1 = quad_bcast_u<2>(__float_as_uint(b[K].x)), t2 = quad_bcast_u<3>(__float_as_uint(b[K].y));   \
            if (m == K) { nxv = t0; chx = t1; lut = t2; }                                                          \
        }
        nxv = 0; fyv = 0; chx = 0; lut = 0;
        QUAD_TAKE(0) QUAD_TAKE(1) QUAD_TAKE(2) QUAD_TAKE(3)
#undef QUAD_TAKE
        acc += nxv + fyv;
        idx = __umulhi(hash32(idx + chx + lut + it), n_nodes);
    }
    if (acc == 1234.5f) out[gid] = acc;
}

template <bool DMA>
__device__ __forceinline__ void coop_body(const char* nodes, uint32_t n_nodes, float* out, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) char s_stage[4][64 * 128];     // per wave: 64 nodes
    __shared__ uint32_t s_idx[4][64];
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t i = lane >> 3, c = lane & 7;          // this lane loads chunk c^i ... of the node of owner 8 j + i
    const uint32_t own_i = lane & 7, own_j = lane >> 3;  // as an owner: my node sits in piece own_j, sub-node own_i
    const uint32_t sx = (gid & 1) ? 3u : 0u, sy = (gid & 2) ? 3u : 0u, sz = (gid & 4) ? 3u : 0u;   // chunk numbers: rows 0..5, children 6, tail 7
    // owner-side LDS byte addresses of my node's chunks (logical chunk q lives at physical chunk q ^ own_i)
    char* my = &s_stage[w][(own_j * 8 + own_i) * 128];
    auto chunk = [&](uint32_t q) -> const float4* { return reinterpret_cast<const float4*>(my - own_i * 128 + 0) + 0; };
    (void)chunk;
    const uint32_t base = (own_j * 8u) * 128u;          // piece base
    auto at = [&](uint32_t q) -> const char* { return &s_stage[w][base + ((own_i * 8u) + (q ^ own_i)) * 16u]; };
    const char* a_nx = at(0 + sx), *a_fx = at(3 - sx), *a_ny = at(1 + sy), *a_fy = at(4 - sy), *a_nz = at(2 + sz), *a_fz = at(5 - sz), *a_ch = at(6), *a_tl = at(7);
    float acc = 0.0f;
    uint32_t idx = __umulhi(hash32(gid ^ seed), n_nodes);
    for (int it = 0; it < ITER; it++) {
        // owners publish their node index, transposed so that each loader lane reads its 8 owners' indices as two b128
        s_idx[w][own_i * 8 + own_j] = idx;
        __builtin_amdgcn_wave_barrier();
        const uint4 ia = *reinterpret_cast<const uint4*>(&s_idx[w][i * 8]), ib = *reinterpret_cast<const uint4*>(&s_idx[w][i * 8 + 4]);
        const uint32_t ids[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
        const uint32_t coff = ((c ^ i) << 4);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const char* src = nodes + ((ids[j] << 7) + coff);
            char* dst = &s_stage[w][j * 1024];
            if constexpr (DMA) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else {
                const float4 v = *reinterpret_cast<const float4*>(src);
                *reinterpret_cast<float4*>(dst + lane * 16) = v;
            }
        }
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float4 nx = *(const float4*)a_nx, fx = *(const float4*)a_fx, ny = *(const float4*)a_ny, fy = *(const float4*)a_fy, nz = *(const float4*)a_nz, fz = *(const float4*)a_fz;
        const uint4 ch = *(const uint4*)a_ch;
        const uint32_t lut = *(const uint32_t*)(a_tl + 4);
        acc += nx.x + fx.y + ny.z + fy.w + nz.x + fz.y;
        idx = __umulhi(hash32(idx + ch.x + lut + it), n_nodes);
        __builtin_amdgcn_wave_barrier();
    }
    if (acc == 1234.5f) out[gid] = acc;
}
extern "C" __global__ void __launch_bounds__(BLOCK) k_coop(const char* nodes, uint32_t n_nodes, float* out, uint32_t seed) { coop_body<false>(nodes, n_nodes, out, seed); }
extern "C" __global__ void __launch_bounds__(BLOCK) k_coop_dma(const char* nodes, uint32_t n_nodes, float* out, uint32_t seed) { coop_body<true>(nodes, n_nodes, out, seed); }

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
    std::vector<uint32_t> h((size_t)n_nodes * 32);
    for (size_t k = 0; k < h.size(); k++) h[k] = (uint32_t)(k * 2654435761u) >> 8;
    (void)hipMemcpy(nodes, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    struct K { const char* name; void (*fn)(const char*, uint32_t, float*, uint32_t); };
    K ks[] = {{"scatter (today)", k_scatter}, {"coop via VGPR + ds_write", k_coop}, {"coop via LDS-DMA", k_coop_dma}, {"quad + DPP (no LDS)", k_quad}};
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::printf("table %.1f MB (%u nodes), %d CUs\n", mb, n_nodes, cus);
    for (int bpc : {2, 4}) {
        for (auto& k : ks) {
            int blocks = cus * bpc;
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(BLOCK), 0, 0, nodes, n_nodes, out, 1u);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(BLOCK), 0, 0, nodes, n_nodes, out, 2u);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            double visits = (double)blocks * BLOCK * ITER;
            std::printf("%d waves/SIMD  %-26s %8.3f ms  %7.1f G node visits/s  %6.2f TB/s of node lines  (%s)\n", bpc, k.name, ms, visits / ms / 1e6, visits * 128 / ms / 1e9,
                        hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
