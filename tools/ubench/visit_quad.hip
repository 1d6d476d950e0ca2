// Micro-benchmark: a COMPLETE 4-wide node visit -- fetch, four slab tests in the reference's un-fused arithmetic, front-to-back order,
// pushes onto an LDS stack, pop of the next entry -- organised two ways:
//   lane   one ray per lane (k_trace's node_step_lean): seven 16-byte loads on the lane's own 128-byte node, 4 slab tests per lane,
//          the order applied to SGPR hit masks, four EXEC-predicated pushes;
//   quad   one ray per lane QUAD, lane c owns child c (round 2's verdict, item 4; the reference's own shape, qbvh_x86.rs:26-67): the node
//          is four 32-byte child records {lo.xyz, ref | hi.xyz, order LUT}, each lane loads its child with two dwordx4 -- the quad's
//          eight loads fall in ONE line --, does ONE slab test, the hit bits of the quad come from one ballot, ORDER_TABLE
//          (qbvh_x86.rs:186-204) is a 128-entry LDS table indexed by (hit mask, node_idx) that hands every child its push slot, every
//          lane pushes its own child at top + slot, and all four lanes keep the ray's stack pointer.
//   pair   (round 4; round 3's review, item 8) one ray per lane PAIR: lane c owns children 2c and 2c + 1 of the same 32-byte child records -- four dwordx4 per
//          lane, the pair's eight loads fall in one line: 4 tag look-ups per ray-visit instead of 7 --, TWO slab tests per lane, the order look-up and the
//          stack bookkeeping paid twice per ray (the quad form: four times), 32 rays per wave instruction.
// Both walk a synthetic table of random boxes with random rays, so the instruction mix and the memory pattern are the real ones
// while the "tree" never ends (the next node is a hash of the popped reference).  Reported: ray-visits per second over the chip.
// What the numbers are for: the quad form cuts the L1 tag look-ups per visit (the resource that bounds k_trace) by ~3.5x, but it
// spends a wave instruction on 16 rays instead of 64, so everything that is not the slab arithmetic costs 4x per ray.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o visit_quad visit_quad.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define ITER 256
#define BLOCK 256
#define SLOTS 16
#define EMPTY 0xffffffffu

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float v_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float v_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float v_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float v_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

struct Ray { float ox, oy, oz, ix, iy, iz, tmin, tmax; uint32_t oct; };
__device__ __forceinline__ Ray make_ray(uint32_t id) {
    Ray r;
    const uint32_t h0 = hash32(id * 3u + 1u), h1 = hash32(id * 3u + 2u), h2 = hash32(id * 3u + 3u);
    r.ox = (h0 & 0xffff) * (1.0f / 65536.0f); r.oy = (h1 & 0xffff) * (1.0f / 65536.0f); r.oz = (h2 & 0xffff) * (1.0f / 65536.0f);
    float dx = ((h0 >> 16) + 1u) * (1.0f / 32768.0f) - 1.0f, dy = ((h1 >> 16) + 1u) * (1.0f / 32768.0f) - 1.0f, dz = ((h2 >> 16) + 1u) * (1.0f / 32768.0f) - 1.0f;
    if (dx == 0.0f) dx = 0.5f; if (dy == 0.0f) dy = 0.5f; if (dz == 0.0f) dz = 0.5f;
    r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
    r.oct = (dx < 0.0f ? 1u : 0u) | (dy < 0.0f ? 2u : 0u) | (dz < 0.0f ? 4u : 0u);
    r.tmin = 0.0f; r.tmax = 4.0f;
    return r;
}

// ---- lane form: PtNode layout (bmin x/y/z rows, bmax x/y/z rows, child refs; split axes in bits 26..27 of refs 0 / 1 / 3)
extern "C" __global__ void __launch_bounds__(BLOCK, 4) k_lane(const char* nodes, uint32_t n_nodes, uint32_t* out, uint32_t seed) {
    __shared__ uint32_t s_stack[SLOTS * BLOCK];
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x;
    const Ray r = make_ray(gid ^ seed);
    const uint32_t ox = (r.oct & 1u) ? 3u : 0u, oy = (r.oct & 2u) ? 3u : 0u, oz = (r.oct & 4u) ? 3u : 0u;
    const uint32_t o_nx = ox << 4, o_fx = (3u - ox) << 4, o_ny = (1u + oy) << 4, o_fy = (4u - oy) << 4, o_nz = (2u + oz) << 4, o_fz = (5u - oz) << 4;
    uint32_t sp = 0, top = __umulhi(hash32(gid ^ seed), n_nodes), acc = 0;
    for (int it = 0; it < ITER; it++) {
        const uint32_t ref = top;
        const uint32_t no = __umulhi(ref, n_nodes) << 7;
        if (sp > 0) { sp--; top = s_stack[sp * BLOCK + threadIdx.x]; } else top = EMPTY;
        const float4 nx = *(const float4*)(nodes + (no + o_nx)), fx = *(const float4*)(nodes + (no + o_fx));
        const float4 ny = *(const float4*)(nodes + (no + o_ny)), fy = *(const float4*)(nodes + (no + o_fy));
        const float4 nz = *(const float4*)(nodes + (no + o_nz)), fz = *(const float4*)(nodes + (no + o_fz));
        const uint4 ch = *(const uint4*)(nodes + (no + 96u));
#define SLAB(C) (v_min3(v_min(r.tmax, (fx.C - r.ox) * r.ix), (fy.C - r.oy) * r.iy, (fz.C - r.oz) * r.iz) >= v_max3(v_max(r.tmin, (nx.C - r.ox) * r.ix), (ny.C - r.oy) * r.iy, (nz.C - r.oz) * r.iz))
        const bool h0 = SLAB(x), h1 = SLAB(y), h2 = SLAB(z), h3 = SLAB(w);
#undef SLAB
        const bool T = __builtin_amdgcn_ubfe(r.oct, __builtin_amdgcn_ubfe(ch.x, 26, 2), 1) != 0u;
        const bool L = __builtin_amdgcn_ubfe(r.oct, __builtin_amdgcn_ubfe(ch.y, 26, 2), 1) != 0u;
        const bool R = __builtin_amdgcn_ubfe(r.oct, __builtin_amdgcn_ubfe(ch.w, 26, 2), 1) != 0u;
        const uint32_t l0 = L ? ch.x : ch.y, l1 = L ? ch.y : ch.x, r0 = R ? ch.z : ch.w, r1 = R ? ch.w : ch.z;
        const uint32_t c0 = T ? l0 : r0, c1 = T ? l1 : r1, c2 = T ? r0 : l0, c3 = T ? r1 : l1;
        const unsigned long long H0 = __ballot(h0), H1 = __ballot(h1), H2 = __ballot(h2), H3 = __ballot(h3), Tm = __ballot(T), Lm = __ballot(L), Rm = __ballot(R);
        const unsigned long long yl = Lm & (H0 ^ H1), el0 = H1 ^ yl, el1 = H0 ^ yl;
        const unsigned long long yr = Rm & (H2 ^ H3), er0 = H3 ^ yr, er1 = H2 ^ yr;
        const unsigned long long y0 = Tm & (el0 ^ er0), e0 = er0 ^ y0, e2 = el0 ^ y0;
        const unsigned long long y1 = Tm & (el1 ^ er1), e1 = er1 ^ y1, e3 = el1 ^ y1;
        const unsigned long long me = 1ull << (threadIdx.x & 63u);
        // pushes (the kernel runs them with EXEC = the condition; a bounded ring here so that the walk never ends or overflows)
        if (e0 & me) { if (top != EMPTY) { s_stack[sp * BLOCK + threadIdx.x] = top; sp = (sp + 1) & (SLOTS - 1); } top = c0; }
        if (e1 & me) { if (top != EMPTY) { s_stack[sp * BLOCK + threadIdx.x] = top; sp = (sp + 1) & (SLOTS - 1); } top = c1; }
        if (e2 & me) { if (top != EMPTY) { s_stack[sp * BLOCK + threadIdx.x] = top; sp = (sp + 1) & (SLOTS - 1); } top = c2; }
        if (e3 & me) { if (top != EMPTY) { s_stack[sp * BLOCK + threadIdx.x] = top; sp = (sp + 1) & (SLOTS - 1); } top = c3; }
        acc += (uint32_t)__popcll((e0 | e1 | e2 | e3) & me);
        if (top == EMPTY) top = hash32(ref + it);           // the walk goes on: a fresh pseudo-random node
        top = hash32(top) | 0u;
    }
    if (acc == 0xdeadbeefu) out[gid] = acc;
}

// ---- quad form: node = 4 x {lo.x lo.y lo.z ref | hi.x hi.y hi.z lut}; lut = node_idx (3 bits) per ray octant, 24 bits
extern "C" __global__ void __launch_bounds__(BLOCK, 4) k_quad(const char* nodes, uint32_t n_nodes, uint32_t* out, uint32_t seed) {
    __shared__ uint32_t s_stack[SLOTS * (BLOCK / 4)];      // one stack per quad (ray)
    __shared__ uint32_t s_order[128];                      // (hit mask << 3 | node_idx) -> 2-bit push slot per child (bits 2c..2c+1), count in bits 8..10
    const uint32_t c = threadIdx.x & 3u, q = threadIdx.x >> 2;
    if (threadIdx.x < 128) {
        // ORDER_TABLE in closed form: pops visit {0,1} before {2,3} iff T = 0 ... what matters here is that it is a table look-up
        const uint32_t m = threadIdx.x >> 3, idx = threadIdx.x & 7u;
        const uint32_t T = (idx >> 2) & 1u, L = (idx >> 1) & 1u, R = idx & 1u;
        const uint32_t l0 = L ? 0u : 1u, l1 = L ? 1u : 0u, r0 = R ? 2u : 3u, r1 = R ? 3u : 2u;
        const uint32_t ord[4] = {T ? l0 : r0, T ? l1 : r1, T ? r0 : l0, T ? r1 : l1};      // push order
        uint32_t e = 0, n = 0;
        for (int k = 0; k < 4; k++) if (m & (1u << ord[k])) { e |= n << (2u * ord[k]); n++; }
        s_order[threadIdx.x] = e | (n << 8);
    }
    __syncthreads();
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x;
    const Ray r = make_ray((gid >> 2) ^ seed);             // the four lanes of a quad carry the same ray
    const uint32_t oct3 = r.oct * 3u, c2 = c * 2u, coff = c << 5;
    const uint32_t qshift = (threadIdx.x & 63u) & ~3u;
    uint32_t sp = 0, top = __umulhi(hash32((gid >> 2) ^ seed), n_nodes), acc = 0;
    for (int it = 0; it < ITER; it++) {
        const uint32_t ref = top;
        const uint32_t no = (__umulhi(ref, n_nodes) << 7) + coff;
        const float4 lo = *(const float4*)(nodes + no), hi = *(const float4*)(nodes + (no + 16u));
        // one slab test: both planes of every axis, near = min, far = max (the direction sign picks the same pair)
        const float ax = (lo.x - r.ox) * r.ix, bx = (hi.x - r.ox) * r.ix, ay = (lo.y - r.oy) * r.iy, by = (hi.y - r.oy) * r.iy, az = (lo.z - r.oz) * r.iz, bz = (hi.z - r.oz) * r.iz;
        const float tn = v_max(v_max3(r.tmin, v_min(ax, bx), v_min(ay, by)), v_min(az, bz));
        const float tf = v_min(v_min3(r.tmax, v_max(ax, bx), v_max(ay, by)), v_max(az, bz));
        const uint32_t cref = __float_as_uint(lo.w);
        const bool hit = tf >= tn && cref != EMPTY;
        const unsigned long long H = __ballot(hit);
        const uint32_t m4 = (uint32_t)(H >> qshift) & 15u;
        const uint32_t nidx = __builtin_amdgcn_ubfe(__float_as_uint(hi.w), oct3, 3);
        const uint32_t e = s_order[(m4 << 3) | nidx];
        const uint32_t slot = __builtin_amdgcn_ubfe(e, c2, 2), cnt = e >> 8;
        // pop, then every hit child goes to stack[sp + slot]; all four lanes keep sp
        if (sp > 0) sp--;
        if (hit) s_stack[((sp + slot) & (SLOTS - 1)) * (BLOCK / 4) + q] = cref;
        sp = (sp + cnt) & (SLOTS - 1);
        __builtin_amdgcn_wave_barrier();
        top = sp > 0 ? s_stack[((sp - 1) & (SLOTS - 1)) * (BLOCK / 4) + q] : EMPTY;     // the new top, by all four lanes (one broadcast read)
        acc += cnt;
        if (top == EMPTY || cnt == 0) top = hash32(ref + it);
        top = hash32(top);
    }
    if (acc == 0xdeadbeefu) out[gid] = acc;
}


// ---- pair form: the quad form's node (4 x {lo.x lo.y lo.z ref | hi.x hi.y hi.z lut}), lane c of a pair owns children 2c and 2c + 1
extern "C" __global__ void __launch_bounds__(BLOCK, 4) k_pair(const char* nodes, uint32_t n_nodes, uint32_t* out, uint32_t seed) {
    __shared__ uint32_t s_stack[SLOTS * (BLOCK / 2)];      // one stack per pair (ray)
    __shared__ uint32_t s_order[128];
    const uint32_t c = threadIdx.x & 1u, q = threadIdx.x >> 1;
    if (threadIdx.x < 128) {
        const uint32_t m = threadIdx.x >> 3, idx = threadIdx.x & 7u;
        const uint32_t T = (idx >> 2) & 1u, L = (idx >> 1) & 1u, R = idx & 1u;
        const uint32_t l0 = L ? 0u : 1u, l1 = L ? 1u : 0u, r0 = R ? 2u : 3u, r1 = R ? 3u : 2u;
        const uint32_t ord[4] = {T ? l0 : r0, T ? l1 : r1, T ? r0 : l0, T ? r1 : l1};
        uint32_t e = 0, n = 0;
        for (int k = 0; k < 4; k++) if (m & (1u << ord[k])) { e |= n << (2u * ord[k]); n++; }
        s_order[threadIdx.x] = e | (n << 8);
    }
    __syncthreads();
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x;
    const Ray r = make_ray((gid >> 1) ^ seed);             // the two lanes of a pair carry the same ray
    const uint32_t oct3 = r.oct * 3u, ca = c * 4u, cb = c * 4u + 2u, coff = c << 6;
    const uint32_t pshift = (threadIdx.x & 63u) & ~1u;
    uint32_t sp = 0, top = __umulhi(hash32((gid >> 1) ^ seed), n_nodes), acc = 0;
    for (int it = 0; it < ITER; it++) {
        const uint32_t ref = top;
        const uint32_t no = (__umulhi(ref, n_nodes) << 7) + coff;
        const float4 lo0 = *(const float4*)(nodes + no), hi0 = *(const float4*)(nodes + (no + 16u)), lo1 = *(const float4*)(nodes + (no + 32u)), hi1 = *(const float4*)(nodes + (no + 48u));
#define PSLAB(lo, hi, tn, tf)                                                                                                                                        \
        const float ax##tn = (lo.x - r.ox) * r.ix, bx##tn = (hi.x - r.ox) * r.ix, ay##tn = (lo.y - r.oy) * r.iy, by##tn = (hi.y - r.oy) * r.iy, az##tn = (lo.z - r.oz) * r.iz, \
                    bz##tn = (hi.z - r.oz) * r.iz;                                                                                                                   \
        const float tn = v_max(v_max3(r.tmin, v_min(ax##tn, bx##tn), v_min(ay##tn, by##tn)), v_min(az##tn, bz##tn));                                                   \
        const float tf = v_min(v_min3(r.tmax, v_max(ax##tn, bx##tn), v_max(ay##tn, by##tn)), v_max(az##tn, bz##tn));
        PSLAB(lo0, hi0, tn0, tf0)
        PSLAB(lo1, hi1, tn1, tf1)
#undef PSLAB
        const uint32_t ref_a = __float_as_uint(lo0.w), ref_b = __float_as_uint(lo1.w);
        const bool hit_a = tf0 >= tn0 && ref_a != EMPTY, hit_b = tf1 >= tn1 && ref_b != EMPTY;
        const unsigned long long Ha = __ballot(hit_a), Hb = __ballot(hit_b);
        const uint32_t a2 = (uint32_t)(Ha >> pshift) & 3u, b2 = (uint32_t)(Hb >> pshift) & 3u;      // bit 0: lane 0's child (0 / 1), bit 1: lane 1's (2 / 3)
        const uint32_t m4 = (a2 & 1u) | ((b2 & 1u) << 1) | ((a2 >> 1) << 2) | ((b2 >> 1) << 3);
        const uint32_t nidx = __builtin_amdgcn_ubfe(__float_as_uint(hi0.w), oct3, 3);
        const uint32_t e = s_order[(m4 << 3) | nidx];
        const uint32_t slot_a = __builtin_amdgcn_ubfe(e, ca, 2), slot_b = __builtin_amdgcn_ubfe(e, cb, 2), cnt = e >> 8;
        if (sp > 0) sp--;
        if (hit_a) s_stack[((sp + slot_a) & (SLOTS - 1)) * (BLOCK / 2) + q] = ref_a;
        if (hit_b) s_stack[((sp + slot_b) & (SLOTS - 1)) * (BLOCK / 2) + q] = ref_b;
        sp = (sp + cnt) & (SLOTS - 1);
        __builtin_amdgcn_wave_barrier();
        top = sp > 0 ? s_stack[((sp - 1) & (SLOTS - 1)) * (BLOCK / 2) + q] : EMPTY;
        acc += cnt;
        if (top == EMPTY || cnt == 0) top = hash32(ref + it);
        top = hash32(top);
    }
    if (acc == 0xdeadbeefu) out[gid] = acc;
}

// ---- coop form (round 4): one ray per lane as in `lane`, but the two lanes of a PAIR fetch each other's node together.  Node A belongs to the even
// lane, node B to the odd one; in loads 1-3 both lanes read from A (the owner its three near rows, the partner A's three far rows, addressed with the
// OWNER's sign-dependent offsets), in loads 4-6 both read from B; the child references are fetched by each lane for itself.  Six of the seven loads
// of a visit then see two lanes per 128-byte line: 4 tag look-ups per ray-visit instead of 7, with 64 rays per wave.  Afterwards the odd lanes swap
// registers X1-3 <-> X4-6 (v_swap_b32 under EXEC = odd lanes), so that every lane finds its near rows in X1-3 and its far rows in the PARTNER's X4-6, which
// the slab arithmetic reads through DPP (quad_perm [1,0,3,2]) -- a source-operand modifier, no instruction of its own when the compiler folds it.
#define DPP_SWAP_PAIR(x) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), 0xB1, 0xF, 0xF, false))
extern "C" __global__ void __launch_bounds__(BLOCK, 4) k_coop(const char* nodes, uint32_t n_nodes, uint32_t* out, uint32_t seed) {
    __shared__ uint32_t s_stack[SLOTS * BLOCK];
    const uint32_t gid = blockIdx.x * BLOCK + threadIdx.x;
    const Ray r = make_ray(gid ^ seed);
    const bool odd = (threadIdx.x & 1u) != 0u;
    const uint32_t ox = (r.oct & 1u) ? 3u : 0u, oy = (r.oct & 2u) ? 3u : 0u, oz = (r.oct & 4u) ? 3u : 0u;
    const uint32_t o_nx = ox << 4, o_fx = (3u - ox) << 4, o_ny = (1u + oy) << 4, o_fy = (4u - oy) << 4, o_nz = (2u + oz) << 4, o_fz = (5u - oz) << 4;
    // the partner's far offsets (per-ray constants: refreshed when a lane takes a new ray)
    const uint32_t p_fx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o_fx, 0xB1, 0xF, 0xF, false), p_fy = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o_fy, 0xB1, 0xF, 0xF, false),
                   p_fz = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o_fz, 0xB1, 0xF, 0xF, false);
    const uint32_t cAx = odd ? p_fx : o_nx, cAy = odd ? p_fy : o_ny, cAz = odd ? p_fz : o_nz;          // what this lane reads of node A (the even lane's)
    const uint32_t cBx = odd ? o_nx : p_fx, cBy = odd ? o_ny : p_fy, cBz = odd ? o_nz : p_fz;          // ... of node B (the odd lane's)
    const unsigned long long odd_mask = 0xAAAAAAAAAAAAAAAAull;
    uint32_t sp = 0, top = __umulhi(hash32(gid ^ seed), n_nodes), acc = 0;
    for (int it = 0; it < ITER; it++) {
        const uint32_t ref = top;
        const uint32_t no = __umulhi(ref, n_nodes) << 7;
        if (sp > 0) { sp--; top = s_stack[sp * BLOCK + threadIdx.x]; } else top = EMPTY;
        const uint32_t no_p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)no, 0xB1, 0xF, 0xF, false);
        const uint32_t bA = odd ? no_p : no, bB = odd ? no : no_p;
        float4 x1 = *(const float4*)(nodes + (bA + cAx)), x2 = *(const float4*)(nodes + (bA + cAy)), x3 = *(const float4*)(nodes + (bA + cAz));
        float4 x4 = *(const float4*)(nodes + (bB + cBx)), x5 = *(const float4*)(nodes + (bB + cBy)), x6 = *(const float4*)(nodes + (bB + cBz));
        const uint4 ch = *(const uint4*)(nodes + (no + 96u));
        {   // odd lanes: X1-3 <-> X4-6
            unsigned long long sv;
            asm volatile("s_mov_b64 %[sv], exec\n\ts_and_b64 exec, exec, %[om]\n\t"
                         "v_swap_b32 %0, %12\n\tv_swap_b32 %1, %13\n\tv_swap_b32 %2, %14\n\tv_swap_b32 %3, %15\n\t"
                         "v_swap_b32 %4, %16\n\tv_swap_b32 %5, %17\n\tv_swap_b32 %6, %18\n\tv_swap_b32 %7, %19\n\t"
                         "v_swap_b32 %8, %20\n\tv_swap_b32 %9, %21\n\tv_swap_b32 %10, %22\n\tv_swap_b32 %11, %23\n\t"
                         "s_mov_b64 exec, %[sv]\n\ts_nop 1"
                         : "+v"(x1.x), "+v"(x1.y), "+v"(x1.z), "+v"(x1.w), "+v"(x2.x), "+v"(x2.y), "+v"(x2.z), "+v"(x2.w), "+v"(x3.x), "+v"(x3.y), "+v"(x3.z), "+v"(x3.w),
                           "+v"(x4.x), "+v"(x4.y), "+v"(x4.z), "+v"(x4.w), "+v"(x5.x), "+v"(x5.y), "+v"(x5.z), "+v"(x5.w), "+v"(x6.x), "+v"(x6.y), "+v"(x6.z), "+v"(x6.w), [sv] "=&s"(sv)
                         : [om] "s"(odd_mask));
        }
#define CSLAB(C) (v_min3(v_min(r.tmax, (DPP_SWAP_PAIR(x4.C) - r.ox) * r.ix), (DPP_SWAP_PAIR(x5.C) - r.oy) * r.iy, (DPP_SWAP_PAIR(x6.C) - r.oz) * r.iz) >= \
                  v_max3(v_max(r.tmin, (x1.C - r.ox) * r.ix), (x2.C - r.oy) * r.iy, (x3.C - r.oz) * r.iz))
        const bool h0 = CSLAB(x), h1 = CSLAB(y), h2 = CSLAB(z), h3 = CSLAB(w);
#undef CSLAB
        const bool T = __builtin_amdgcn_ubfe(r.oct, __builtin_amdgcn_ubfe(ch.x, 26, 2), 1) != 0u;
        const bool L = __builtin_amdgcn_ubfe(r.oct, __builtin_amdgcn_ubfe(ch.y, 26, 2), 1) != 0u;
        const bool R = __builtin_amdgcn_ubfe(r.oct, __builtin_amdgcn_ubfe(ch.w, 26, 2), 1) != 0u;
        const uint32_t l0 = L ? ch.x : ch.y, l1 = L ? ch.y : ch.x, r0 = R ? ch.z : ch.w, r1 = R ? ch.w : ch.z;
        const uint32_t c0 = T ? l0 : r0, c1 = T ? l1 : r1, c2 = T ? r0 : l0, c3 = T ? r1 : l1;
        const unsigned long long H0 = __ballot(h0), H1 = __ballot(h1), H2 = __ballot(h2), H3 = __ballot(h3), Tm = __ballot(T), Lm = __ballot(L), Rm = __ballot(R);
        const unsigned long long yl = Lm & (H0 ^ H1), el0 = H1 ^ yl, el1 = H0 ^ yl;
        const unsigned long long yr = Rm & (H2 ^ H3), er0 = H3 ^ yr, er1 = H2 ^ yr;
        const unsigned long long y0 = Tm & (el0 ^ er0), e0 = er0 ^ y0, e2 = el0 ^ y0;
        const unsigned long long y1 = Tm & (el1 ^ er1), e1 = er1 ^ y1, e3 = el1 ^ y1;
        const unsigned long long me = 1ull << (threadIdx.x & 63u);
        if (e0 & me) { if (top != EMPTY) { s_stack[sp * BLOCK + threadIdx.x] = top; sp = (sp + 1) & (SLOTS - 1); } top = c0; }
        if (e1 & me) { if (top != EMPTY) { s_stack[sp * BLOCK + threadIdx.x] = top; sp = (sp + 1) & (SLOTS - 1); } top = c1; }
        if (e2 & me) { if (top != EMPTY) { s_stack[sp * BLOCK + threadIdx.x] = top; sp = (sp + 1) & (SLOTS - 1); } top = c2; }
        if (e3 & me) { if (top != EMPTY) { s_stack[sp * BLOCK + threadIdx.x] = top; sp = (sp + 1) & (SLOTS - 1); } top = c3; }
        acc += (uint32_t)__popcll((e0 | e1 | e2 | e3) & me);
        if (top == EMPTY) top = hash32(ref + it);
        top = hash32(top) | 0u;
    }
    if (acc == 0xdeadbeefu) out[gid] = acc;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    double mb = argc > 1 ? atof(argv[1]) : 21.0;
    uint32_t n_nodes = (uint32_t)(mb * 1e6 / 128);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    char *nodes_lane, *nodes_quad; uint32_t* out;
    (void)hipMalloc(&nodes_lane, (size_t)n_nodes * 128);
    (void)hipMalloc(&nodes_quad, (size_t)n_nodes * 128);
    (void)hipMalloc(&out, (size_t)cus * 8 * BLOCK * 4);
    // random child boxes inside the unit cube, about a quarter of the cube's extent per axis; references = random words below 2^26
    std::vector<float> hl((size_t)n_nodes * 32), hq((size_t)n_nodes * 32);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (uint32_t n = 0; n < n_nodes; n++) {
        for (int k = 0; k < 4; k++) {
            float lo[3], hi[3];
            for (int a = 0; a < 3; a++) { const float c0 = rnd(), e = 0.05f + 0.3f * rnd(); lo[a] = c0 - e; hi[a] = c0 + e; }
            uint32_t ref = ((uint32_t)(rnd() * 67108864.0f)) & 0x3ffffffu;
            if (k == 0 || k == 1 || k == 3) ref |= ((uint32_t)(rnd() * 3.0f) % 3u) << 26;          // split axes ride in refs 0 / 1 / 3
            float fref; std::memcpy(&fref, &ref, 4);
            for (int a = 0; a < 3; a++) { hl[(size_t)n * 32 + a * 4 + k] = lo[a]; hl[(size_t)n * 32 + 12 + a * 4 + k] = hi[a]; }
            hl[(size_t)n * 32 + 24 + k] = fref;
            uint32_t lut = (uint32_t)(rnd() * 16777216.0f);
            float flut; std::memcpy(&flut, &lut, 4);
            float* qd = &hq[(size_t)n * 32 + k * 8];
            qd[0] = lo[0]; qd[1] = lo[1]; qd[2] = lo[2]; qd[3] = fref; qd[4] = hi[0]; qd[5] = hi[1]; qd[6] = hi[2]; qd[7] = flut;
        }
    }
    (void)hipMemcpy(nodes_lane, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(nodes_quad, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::printf("table %.1f MB (%u nodes), %d CUs, %d visits per ray\n", mb, n_nodes, cus, ITER);
    for (int bpc : {2, 4}) {
        for (int form = 0; form < 4; form++) {
            const int blocks = cus * bpc;
            auto launch = [&](uint32_t seed) {
                if (form == 0) hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(BLOCK), 0, 0, nodes_lane, n_nodes, out, seed);
                else if (form == 1) hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(BLOCK), 0, 0, nodes_quad, n_nodes, out, seed);
                else if (form == 2) hipLaunchKernelGGL(k_pair, dim3(blocks), dim3(BLOCK), 0, 0, nodes_quad, n_nodes, out, seed);
                else hipLaunchKernelGGL(k_coop, dim3(blocks), dim3(BLOCK), 0, 0, nodes_lane, n_nodes, out, seed);
            };
            launch(1u);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            launch(2u);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double rays = (double)blocks * BLOCK / ((form == 0 || form == 3) ? 1 : (form == 1 ? 4 : 2));
            std::printf("%d waves/SIMD  %-28s %8.3f ms  %7.1f G ray-visits/s   (%.0f rays per wave-instruction)  (%s)\n", bpc,
                        form == 0 ? "lane: one ray per lane" : (form == 1 ? "quad: one ray per lane quad" : (form == 2 ? "pair: one ray per lane pair" : "coop: lane form, paired fetch")), ms, rays * ITER / ms / 1e6,
                        (form == 0 || form == 3) ? 64.0 : (form == 1 ? 16.0 : 32.0), hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
