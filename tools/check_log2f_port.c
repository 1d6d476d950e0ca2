/* Checks the log2f restatement used on the device (pt_device_math.h: pt_log2f) against the host libm, over every positive float,
 * with and without the FMA contraction glibc's x86-64 multiarch build uses (both agree after the final rounding).
 *   gcc -O2 -ffp-contract=off -mfma -o /tmp/chk tools/check_log2f_port.c -lm && /tmp/chk      (about 30 s; expect 0 mismatches) */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static inline uint32_t fu(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float uf(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static const double T[16][2] = {
 { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 }, { 0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2 },
 { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 }, { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3 },
 { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 }, { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 },
 { 0x1p+0, 0x0p+0 }, { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 }, { 0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3 },
 { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 }, { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 },
 { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 } };
static const double A[4] = { -0x1.712b6f70a7e4dp-2, 0x1.ecabf496832ep-2, -0x1.715479ffae3dep-1, 0x1.715475f35c8b8p0 };
static int USE_FMA = 1;
float my_log2f(float x){
    uint32_t ix = fu(x);
    if (ix == 0x3f800000) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -INFINITY;
        if (ix == 0x7f800000) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return NAN;
        ix = fu(x * 0x1p23f); ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    int i = (tmp >> 19) % 16;
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)tmp >> 23;
    double invc = T[i][0], logc = T[i][1];
    double z = (double)uf(iz);
    double r, y0, r2, y, p;
    if (USE_FMA) {
        r = __builtin_fma(z, invc, -1.0); y0 = logc + (double)k; r2 = r * r;
        y = __builtin_fma(A[1], r, A[2]); y = __builtin_fma(A[0], r2, y); p = __builtin_fma(A[3], r, y0); y = __builtin_fma(y, r2, p);
    } else {
        r = z * invc - 1; y0 = logc + (double)k; r2 = r * r;
        y = A[1] * r + A[2]; y = A[0] * r2 + y; p = A[3] * r + y0; y = y * r2 + p;
    }
    return (float)y;
}
int main(){
    for (USE_FMA = 1; USE_FMA >= 0; USE_FMA--) {
        uint64_t bad=0; uint32_t first=0;
        for (uint64_t u=1; u<0x7f800000u; u++){ float x=uf((uint32_t)u); if (fu(my_log2f(x))!=fu(log2f(x))){ if(!bad) first=(uint32_t)u; bad++; } }
        printf("fma=%d mismatches %llu first %08x\n", USE_FMA, (unsigned long long)bad, first);
    }
    return 0;
}
