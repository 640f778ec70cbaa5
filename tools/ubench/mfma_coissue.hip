// Microbenchmark (round 4, VERDICT r3 #3c): do v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate) and
// v_fmac_f32 issued by SIBLING waves of one SIMD overlap?  If the matrix pipe runs beside the vector pipe at
// >= 1.6x the combined rate, a band-matrix third of the correlation backward's channels could move to it.
//   mode 0: every wave FMA only      mode 1: every wave MFMA only
//   mode 2: waves 0-3 FMA, waves 4-7 MFMA (one of each per SIMD)      mode 3: every wave alternates 16 FMA / 1 MFMA
// Reports ns per (16 FMA + 1 MFMA)-unit and the FLOP rates.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x = seed * 0.5f, y = seed * 0.25f;
    const bool do_fma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool do_mfma = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (do_fma) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        }
        if (do_mfma) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c3, 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + c0[0] + c1[1] + c2[2] + c3[3];
}
template <int MODE> float run(float *out, int iters, int blocks) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float *out; (void)hipMalloc(&out, 2048 * 512 * 4);
    const int iters = 20000;
    for (int blocks : {256, 512}) {   // 2 / 4 waves per SIMD
        const float t0 = run<0>(out, iters, blocks), t1 = run<1>(out, iters, blocks), t2 = run<2>(out, iters, blocks), t3 = run<3>(out, iters, blocks);
        const double waves = double(blocks) * 8;
        const double fma_flop = waves * iters * 64.0 * 64 * 2, mfma_flop = waves * iters * 4.0 * (16 * 16 * 4 * 2);
        printf("%d workgroups of 8 waves (%d waves/SIMD):\n", blocks, blocks / 128);
        printf("  FMA only            %.3f ms  %.1f TFLOP/s\n", t0, fma_flop / t0 * 1e-9);
        printf("  MFMA 16x16x4 only   %.3f ms  %.1f TFLOP/s\n", t1, mfma_flop / t1 * 1e-9);
        printf("  half FMA half MFMA  %.3f ms  FMA part %.1f + MFMA part %.1f = %.1f TFLOP/s  (%.2fx FMA-only)\n", t2,
               0.5 * fma_flop / t2 * 1e-9, 0.5 * mfma_flop / t2 * 1e-9, (0.5 * fma_flop + 0.5 * mfma_flop) / t2 * 1e-9,
               (0.5 * fma_flop + 0.5 * mfma_flop) / t2 / (fma_flop / t0));
        printf("  every wave both     %.3f ms  %.1f TFLOP/s  (%.2fx FMA-only; sum of the separate times %.3f ms)\n", t3,
               (fma_flop + mfma_flop) / t3 * 1e-9, (fma_flop + mfma_flop) / t3 / (fma_flop / t0), t0 + t1);
    }
    return 0;
}
