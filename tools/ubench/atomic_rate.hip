// Micro-benchmark: what a returning atomicAdd on a shared 32-bit counter costs when every wave of the chip does one per loop iteration
// (the queue reservations of k_shade): lane 0 of each of 2048 waves adds to one of K counters `stride` bytes apart, waits for the old
// value and goes on.  Reports atomics / s over the chip and the time per atomic on one counter.
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 2048

__global__ void __launch_bounds__(256) k_atomics(uint32_t* ctr, uint32_t n_ctr, uint32_t stride_words, uint32_t per_iter, uint32_t* out) {
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    uint32_t acc = 0;
    if (lane == 0) {
        for (int it = 0; it < ITER; it++) {
            uint32_t r[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < per_iter; k++) r[k] = atomicAdd(&ctr[((wave + k * 7u) % n_ctr) * stride_words], 1u + (acc & 1u));
            acc += r[0] + r[1] + r[2] + r[3];
        }
        out[wave] = acc;
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 2, waves = blocks * 4;
    uint32_t* ctr; uint32_t* out;
    (void)hipMalloc(&ctr, 64u << 20);
    (void)hipMalloc(&out, waves * 4);
    (void)hipMemset(ctr, 0, 64u << 20);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::printf("%d waves (lane 0 of each), %d iterations\n", waves, ITER);
    struct C { uint32_t n_ctr, stride_bytes, per_iter; };
    const C cs[] = {{1, 128, 1}, {1, 128, 4}, {4, 128, 4}, {4, 4096, 4}, {4, 65536, 4}, {16, 128, 1}, {16, 4096, 1}, {64, 128, 1}, {64, 4096, 1}, {256, 4096, 1}, {2048, 128, 1}};
    for (const C& c : cs) {
        hipLaunchKernelGGL(k_atomics, dim3(blocks), dim3(256), 0, 0, ctr, c.n_ctr, c.stride_bytes / 4, c.per_iter, out);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_atomics, dim3(blocks), dim3(256), 0, 0, ctr, c.n_ctr, c.stride_bytes / 4, c.per_iter, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)waves * ITER * c.per_iter;
        std::printf("%4u counters %6u B apart, %u atomics per iteration: %8.3f ms  %7.1f M atomics/s  %6.1f ns per atomic per counter  %6.2f us per wave iteration  (%s)\n", c.n_ctr,
                    c.stride_bytes, c.per_iter, ms, n / ms / 1e3, ms * 1e6 / (n / c.n_ctr), ms * 1e3 / ITER, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
