// cvt_check.hip -- are gfx950's packed conversions (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, what
// __builtin_convertvector emits) bit-identical to the software round-to-nearest-even of hip_bfloat16(float) and
// to __float2half over ALL 2^32 fp32 inputs?  Counts mismatches among non-NaN inputs and, separately, NaN inputs
// whose result is not a NaN / whose bits differ.  Exit code 0 iff every non-NaN input agrees and every NaN stays a NaN.
#include <hip/hip_runtime.h>
#include <hip/hip_bfloat16.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__global__ void check(unsigned long long *cnt) {
    const unsigned long long n = 1ull << 32;
    unsigned long long bad_b = 0, bad_h = 0, nan_b = 0, nan_h = 0, nanbits_b = 0, nanbits_h = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned u = (unsigned)i;
        const float v = __uint_as_float(u);
        const f2 vv = {v, v};
        const b2 rb = __builtin_convertvector(vv, b2);
        const h2 rh = __builtin_convertvector(vv, h2);
        unsigned hb, hh;
        __builtin_memcpy(&hb, &rb, 4);
        __builtin_memcpy(&hh, &rh, 4);
        const hip_bfloat16 sb = hip_bfloat16(v);
        const __half sh = __float2half(v);
        unsigned short sbb, shb;
        __builtin_memcpy(&sbb, &sb, 2);
        __builtin_memcpy(&shb, &sh, 2);
        const bool isnan_in = (u & 0x7fffffffu) > 0x7f800000u;
        const unsigned short b_lo = hb & 0xffff, b_hi = hb >> 16, h_lo = hh & 0xffff, h_hi = hh >> 16;
        if (!isnan_in) {
            if (b_lo != sbb || b_hi != sbb) ++bad_b;
            if (h_lo != shb || h_hi != shb) ++bad_h;
        } else {
            if ((b_lo & 0x7fff) <= 0x7f80) ++nan_b;            // a NaN that did not stay a NaN
            if ((h_lo & 0x7fff) <= 0x7c00) ++nan_h;
            if (b_lo != sbb) ++nanbits_b;
            if (h_lo != shb) ++nanbits_h;
        }
    }
    atomicAdd(&cnt[0], bad_b); atomicAdd(&cnt[1], bad_h); atomicAdd(&cnt[2], nan_b); atomicAdd(&cnt[3], nan_h);
    atomicAdd(&cnt[4], nanbits_b); atomicAdd(&cnt[5], nanbits_h);
}

int main() {
    unsigned long long *d, h[6] = {0, 0, 0, 0, 0, 0};
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("non-NaN inputs whose packed HW conversion differs from the software one: bf16 %llu, fp16 %llu\n", h[0], h[1]);
    printf("NaN inputs that did not stay NaN: bf16 %llu, fp16 %llu\n", h[2], h[3]);
    printf("NaN inputs whose NaN bits differ (payload / quiet bit): bf16 %llu, fp16 %llu (of 16777214)\n", h[4], h[5]);
    return (h[0] | h[1] | h[2] | h[3]) ? 1 : 0;
}
