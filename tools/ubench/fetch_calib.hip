// Calibration of the FETCH_SIZE counter for k_trace's access pattern: every lane gathers from a random 128-byte line of a table far
// larger than L2 + Infinity Cache (argv[1] MB, default 8192), so nearly every line touched has to cross the fabric once.  Three kernels
// with a KNOWN number of lines touched:
//   k_gather16    one 16-byte load per lane and line (the smallest request the traversal makes)
//   k_gather112   the seven 16-byte rows of a node visit (six plane rows + the child references) of one line
//   k_gather48    the three 16-byte loads of one 48-byte triangle record (records at a 48-byte stride: 1/4 of them straddle two lines)
// Run under `rocprofv3 --pmc FETCH_SIZE` and compare the counter (KiB) with the printed figures: lines x 128 B (what a 128-byte
// L2 line fill moves), lines x 64 B and the bytes the lanes consume.  build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITER 64
#define BLOCK 256

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

extern "C" __global__ void __launch_bounds__(BLOCK) k_gather16(const char* tab, uint32_t n_lines, float* out, uint32_t seed) {
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x;
    float acc = 0.0f;
    uint32_t idx = __umulhi(hash32(gid ^ seed), n_lines);
    for (int it = 0; it < ITER; it++) {
        const float4 a = *(const float4*)(tab + ((size_t)idx << 7) + ((gid & 7u) << 4));
        acc += a.x;
        idx = __umulhi(hash32(idx + __float_as_uint(a.y) + it), n_lines);
    }
    if (acc == 1234.5f) out[gid] = acc;
}

extern "C" __global__ void __launch_bounds__(BLOCK) k_gather112(const char* tab, uint32_t n_lines, float* out, uint32_t seed) {
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x;
    float acc = 0.0f;
    uint32_t idx = __umulhi(hash32(gid ^ seed), n_lines);
    for (int it = 0; it < ITER; it++) {
        const char* p = tab + ((size_t)idx << 7);
        float4 r[7];
#pragma unroll
        for (int k = 0; k < 7; k++) r[k] = *(const float4*)(p + 16 * k);
#pragma unroll
        for (int k = 0; k < 7; k++) acc += r[k].x;
        idx = __umulhi(hash32(idx + __float_as_uint(r[6].y) + it), n_lines);
    }
    if (acc == 1234.5f) out[gid] = acc;
}

extern "C" __global__ void __launch_bounds__(BLOCK) k_gather48(const char* tab, uint32_t n_lines, float* out, uint32_t seed) {
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t n_recs = (uint32_t)(((size_t)n_lines << 7) / 48u) - 1u;
    float acc = 0.0f;
    uint32_t idx = __umulhi(hash32(gid ^ seed), n_recs);
    for (int it = 0; it < ITER; it++) {
        const char* p = tab + (size_t)idx * 48u;
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 16), c = *(const float4*)(p + 32);
        acc += a.x + b.x + c.x;
        idx = __umulhi(hash32(idx + __float_as_uint(c.y) + it), n_recs);
    }
    if (acc == 1234.5f) out[gid] = acc;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    double mb = argc > 1 ? atof(argv[1]) : 8192.0;
    uint32_t n_lines = (uint32_t)(mb * 1048576.0 / 128);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    char* tab; float* out;
    if (hipMalloc(&tab, (size_t)n_lines * 128) != hipSuccess) { std::printf("no memory\n"); return 1; }
    (void)hipMalloc(&out, (size_t)cus * 8 * BLOCK * 4);
    (void)hipMemset(tab, 0x11, (size_t)n_lines * 128);
    (void)hipDeviceSynchronize();
    struct K { const char* name; void (*fn)(const char*, uint32_t, float*, uint32_t); double lines_per_fetch, used; };
    K ks[] = {{"k_gather16", k_gather16, 1.0, 16.0}, {"k_gather112", k_gather112, 1.0, 112.0}, {"k_gather48", k_gather48, 1.25, 48.0}};
    std::printf("table %.0f MiB (%u lines of 128 B), %d CUs\n", mb, n_lines, cus);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (auto& k : ks) {
        int blocks = cus * 8;
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(BLOCK), 0, 0, tab, n_lines, out, 7u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double fetches = (double)blocks * BLOCK * ITER, lines = fetches * k.lines_per_fetch;
        std::printf("%-12s %8.3f ms  gathers %.0f  lines touched %.0f  = %.1f KiB at 128 B/line, %.1f KiB at 64 B/line, %.1f KiB consumed by lanes  (%s)\n", k.name, ms,
                    fetches, lines, lines * 128 / 1024, lines * 64 / 1024, fetches * k.used / 1024, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
