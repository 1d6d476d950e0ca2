/* Checks the logf restatement used on the device (pt_device_math.h: pt_logf; roughness_to_alpha's `ln`,
 * core/distribution/trowbridge_reitz.rs:113-121) against the host libm over every positive float, with and without the FMA
 * contraction glibc's x86-64 multiarch build uses.
 *   gcc -O2 -ffp-contract=off -mfma -o /tmp/chk tools/check_logf_port.c -lm && /tmp/chk      (about 30 s; expect 0 mismatches) */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static inline uint32_t fu(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float uf(uint32_t u){float f;memcpy(&f,&u,4);return f;}
/* glibc sysdeps/ieee754/flt-32/e_logf_data.c (ARM optimized-routines logf): N = 16 intervals, {1/c, log(c)}, degree-3 polynomial */
static const double T[16][2] = {
 { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 }, { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 }, { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 },
 { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 }, { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 }, { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
 { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 }, { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 }, { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 },
 { 0x1p+0, 0x0p+0 }, { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 }, { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
 { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 }, { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },
 { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 } };
static const double A[3] = { -0x1.00ea348b88334p-2, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2 };
static const double LN2 = 0x1.62e42fefa39efp-1;
static int USE_FMA = 1;
float my_logf(float x){
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
    int k = (int32_t)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000u);
    double invc = T[i][0], logc = T[i][1];
    double z = (double)uf(iz);
    double r, y0, r2, y;
    if (USE_FMA) {
        r = __builtin_fma(z, invc, -1.0); y0 = __builtin_fma((double)k, LN2, logc); r2 = r * r;
        y = __builtin_fma(A[1], r, A[2]); y = __builtin_fma(A[0], r2, y); y = __builtin_fma(y, r2, y0 + r);
    } else {
        r = z * invc - 1; y0 = logc + (double)k * LN2; r2 = r * r;
        y = A[1] * r + A[2]; y = A[0] * r2 + y; y = y * r2 + (y0 + r);
    }
    return (float)y;
}
int main(){
    for (USE_FMA = 1; USE_FMA >= 0; USE_FMA--) {
        uint64_t bad=0; uint32_t first=0;
        for (uint64_t u=1; u<0x7f800000u; u++){ float x=uf((uint32_t)u); if (fu(my_logf(x))!=fu(logf(x))){ if(!bad) first=(uint32_t)u; bad++; } }
        printf("fma=%d mismatches %llu first %08x\n", USE_FMA, (unsigned long long)bad, first);
    }
    return 0;
}
