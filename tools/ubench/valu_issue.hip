// Micro-benchmark: VALU issue cost of the instruction forms k_trace's node visit could be built from (gfx950).
// Each kernel runs ITER iterations of 16 independent instructions of one form on every wave; the grid fills the chip at
// W waves per SIMD.  Reported: SIMD cycles per wave-instruction = elapsed * clock / (ITER * 16 * waves_per_simd).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run: ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

typedef float f2 __attribute__((ext_vector_type(2)));

#define KERNEL(name, DECL, OP)                                                         \
    extern "C" __global__ void __launch_bounds__(256) name(float* out, float a, float b) { \
        DECL                                                                           \
        for (int it = 0; it < ITER; it++) { R16(OP) }                                  \
        float s = 0;                                                                   \
        _Pragma("unroll") for (int k = 0; k < 16; k++) s += fin(x[k]);                 \
        if (s == 12345.678f) out[threadIdx.x] = s;                                     \
    }
__device__ inline float fin(float v) { return v; }
__device__ inline float fin(f2 v) { return v.x + v.y; }
__device__ inline float fin(unsigned v) { return (float)v; }

#define DECL_F float x[16]; for (int k = 0; k < 16; k++) x[k] = a + k + threadIdx.x;
#define DECL_F2 f2 x[16]; for (int k = 0; k < 16; k++) { x[k].x = a + k + threadIdx.x; x[k].y = b + k; } f2 bb; bb.x = b; bb.y = a;
#define DECL_U unsigned x[16]; for (int k = 0; k < 16; k++) x[k] = (unsigned)(a + k + threadIdx.x); unsigned ub = (unsigned)b; unsigned long long sm = __ballot(threadIdx.x & 1); unsigned lds_addr = threadIdx.x * 4; __shared__ unsigned s_lds[4096]; s_lds[threadIdx.x] = ub; (void)sm; (void)lds_addr;

#define OP_MUL(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(b));
#define OP_SUB(k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[k]) : "v"(b));
#define OP_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(b));
#define OP_MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(b), "v"(a));
#define OP_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(bb));
#define OP_PKADD(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(bb));
#define OP_PKADD_SEL(k) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(x[k]) : "v"(bb));
#define OP_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(bb));
#define OP_CND(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(ub));
#define OP_BFE(k) asm volatile("v_bfe_u32 %0, %0, %1, 3" : "+v"(x[k]) : "v"(ub));
#define OP_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 7, %1" : "+v"(x[k]) : "v"(ub));
#define OP_PERM(k) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(ub));
#define OP_ADD64(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(y[k]) : "v"(yb));
#define OP_CMP(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[k]), "v"(b) : "vcc");
#define OP_MOV(k) asm volatile("v_mov_b32 %0, %1" : "+v"(x[k]) : "v"(ub));
#define OP_CND64(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(ub), "s"(sm));
#define OP_CNDLIT(k) asm volatile("v_cndmask_b32_e64 %0, 1, 2, vcc" : "=v"(x[k]) : : );
#define OP_BFI(k) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x[k]) : "v"(ub));
#define OP_AND(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[k]) : "v"(ub));
#define OP_XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[k]) : "v"(ub));
#define OP_MAXF(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[k]) : "v"(b));
#define OP_ADDU(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(ub));
#define OP_ADDC(k) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[k]) : "v"(ub) : "vcc");
#define OP_ADDCS(k) asm volatile("v_addc_co_u32_e64 %0, s[20:21], %0, %1, vcc" : "+v"(x[k]) : "v"(ub) : "s20", "s21");
#define OP_CMP64(k) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" :: "v"(x[k]), "v"(b) : "s20", "s21");
#define OP_CMPCND(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[k]) : "v"(b), "v"(a) : "vcc");
#define OP_LSHLOR(k) asm volatile("v_lshl_or_b32 %0, %0, 10, %1" : "+v"(x[k]) : "v"(ub));
#define OP_DSW(k) asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr), "v"(x[k]));
#define OP_DSR(k) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(8)" : "=v"(x[k]) : "v"(lds_addr));
#define OP_MBCNT(k) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x[k]) : "v"(ub));
#define OP_SNOP(k) asm volatile("s_nop 0");
#define OP_SAND(k) asm volatile("s_and_b64 s[20:21], s[20:21], exec" ::: "s20", "s21");
#define OP_SBCNT(k) asm volatile("s_bcnt1_i32_b64 s20, s[22:23]" ::: "s20");
#define OP_BPERM(k) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x[k]) : "v"(ub));

KERNEL(k_mul, DECL_F, OP_MUL)
KERNEL(k_sub, DECL_F, OP_SUB)
KERNEL(k_fma, DECL_F, OP_FMA)
KERNEL(k_max3, DECL_F, OP_MAX3)
KERNEL(k_pk_mul, DECL_F2, OP_PKMUL)
KERNEL(k_pk_add, DECL_F2, OP_PKADD)
KERNEL(k_pk_add_sel, DECL_F2, OP_PKADD_SEL)
KERNEL(k_pk_fma, DECL_F2, OP_PKFMA)
KERNEL(k_cndmask, DECL_U, OP_CND)
KERNEL(k_bfe, DECL_U, OP_BFE)
KERNEL(k_lshl_add, DECL_U, OP_LSHLADD)
KERNEL(k_perm, DECL_U, OP_PERM)
KERNEL(k_cmp, DECL_F, OP_CMP)
KERNEL(k_mov, DECL_U, OP_MOV)
KERNEL(k_bpermute, DECL_U, OP_BPERM)
KERNEL(k_cnd64, DECL_U, OP_CND64)
KERNEL(k_cndlit, DECL_U, OP_CNDLIT)
KERNEL(k_bfi, DECL_U, OP_BFI)
KERNEL(k_and, DECL_U, OP_AND)
KERNEL(k_xor, DECL_U, OP_XOR)
KERNEL(k_maxf, DECL_F, OP_MAXF)
KERNEL(k_addu, DECL_U, OP_ADDU)
KERNEL(k_addc, DECL_U, OP_ADDC)
KERNEL(k_addcs, DECL_U, OP_ADDCS)
KERNEL(k_cmp64, DECL_F, OP_CMP64)
KERNEL(k_cmpcnd, DECL_F, OP_CMPCND)
KERNEL(k_lshlor, DECL_U, OP_LSHLOR)
KERNEL(k_dsw, DECL_U, OP_DSW)
KERNEL(k_dsr, DECL_U, OP_DSR)
KERNEL(k_mbcnt, DECL_U, OP_MBCNT)
KERNEL(k_snop, DECL_U, OP_SNOP)

extern "C" __global__ void __launch_bounds__(256) k_add64(float* out, float a, float b) {
    unsigned long long y[16];
    for (int k = 0; k < 16; k++) y[k] = (unsigned long long)(a + k + threadIdx.x);
    unsigned long long yb = (unsigned long long)b;
    for (int it = 0; it < ITER; it++) { R16(OP_ADD64) }
    unsigned long long s = 0;
    for (int k = 0; k < 16; k++) s += y[k];
    if (s == 12345678ull) out[threadIdx.x] = (float)s;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    double clk = p.clockRate * 1e3;      // Hz
    float* out;
    hipMalloc(&out, 4096);
    struct K { const char* name; void (*fn)(float*, float, float); };
    K ks[] = {{"v_mul_f32", k_mul}, {"v_sub_f32", k_sub}, {"v_fma_f32", k_fma}, {"v_max3_f32", k_max3}, {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add},
              {"v_pk_add_f32 op_sel neg", k_pk_add_sel}, {"v_pk_fma_f32", k_pk_fma}, {"v_cndmask_b32", k_cndmask}, {"v_bfe_u32", k_bfe}, {"v_lshl_add_u32", k_lshl_add},
              {"v_perm_b32", k_perm}, {"v_cmp_lt_f32", k_cmp}, {"v_mov_b32", k_mov}, {"v_lshl_add_u64", k_add64}, {"ds_bpermute_b32+wait", k_bpermute},
              {"v_cndmask_b32_e64 (sgpr mask)", k_cnd64}, {"v_cndmask_b32_e64 1,2,vcc", k_cndlit}, {"v_bfi_b32", k_bfi}, {"v_and_b32", k_and}, {"v_xor_b32", k_xor},
              {"v_max_f32", k_maxf}, {"v_add_u32", k_addu}, {"v_addc_co_u32 vcc", k_addc}, {"v_addc_co_u32_e64 sgpr", k_addcs}, {"v_cmp_lt_f32_e64 sgpr", k_cmp64},
              {"v_cmp+s_nop 1+v_cndmask (3 instr)", k_cmpcnd}, {"v_lshl_or_b32", k_lshlor}, {"ds_write_b32", k_dsw}, {"ds_read_b32 (no wait)", k_dsr},
              {"v_mbcnt_lo", k_mbcnt}, {"s_nop 0", k_snop}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::printf("device: %s, %d CUs, clockRate %.0f MHz (nominal; DVFS lowers it under load)\n", p.name, cus, clk / 1e6);
    for (int wps : {1, 4}) {
        std::printf("-- %d wave(s) per SIMD\n", wps);
        for (auto& k : ks) {
            int blocks = cus * wps;       // 256 threads = 4 waves = one per SIMD
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 2.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 2.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double cyc = ms * 1e-3 * clk / ((double)ITER * 16 * wps);
            std::printf("%-28s %8.3f ms  %6.2f SIMD cycles per wave-instruction (at nominal clock)\n", k.name, ms, cyc);
        }
    }
    return 0;
}
