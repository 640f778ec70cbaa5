// Pins the operand layout of v_mfma_f32_16x16x32_{f16,bf16} (gfx950) that corr_mfma.hip relies on:
//   A (16 x 32): lane l holds row m = l % 16, k = 8 * (l / 16) .. + 7  (8 consecutive k)
//   B (32 x 16): lane l holds col n = l % 16, k = 8 * (l / 16) .. + 7
//   D (16 x 16): lane l holds col n = l % 16, rows m = 4 * (l / 16) + j, j = 0..3
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 tools/ubench/mfma_layout.hip -o /tmp/mfma && /tmp/mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename E, typename V>
__global__ void k(const E *A, const E *B, float *D) {
    const int l = threadIdx.x;
    V a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = A[(l % 16) * 32 + 8 * (l / 16) + i];
        b[i] = B[(8 * (l / 16) + i) * 16 + l % 16];
    }
    f4 c = {0, 0, 0, 0};
    if constexpr (sizeof(E) == 2 && __is_same(E, _Float16)) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(4 * (l / 16) + j) * 16 + l % 16] = c[j];
}

template <typename E, typename V> double run(const char *name) {
    E hA[16 * 32], hB[32 * 16];
    float ref[256] = {0}, out[256];
    srand(7);
    for (int i = 0; i < 512; ++i) { hA[i] = (E)(float)((rand() % 17) - 8); hB[i] = (E)(float)((rand() % 13) - 6); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int kk = 0; kk < 32; ++kk)
        ref[m * 16 + n] += (float)hA[m * 32 + kk] * (float)hB[kk * 16 + n];
    E *dA, *dB; float *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(out));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<E, V>), dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(out, dD, sizeof(out), hipMemcpyDeviceToHost);
    double e = 0;
    for (int i = 0; i < 256; ++i) e = fmax(e, fabs(out[i] - ref[i]));
    printf("%s: max |D - ref| = %g (integers: must be 0)\n", name, e);
    return e;
}
int main() {
    const double e = run<_Float16, h8>("v_mfma_f32_16x16x32_f16") + run<__bf16, b8>("v_mfma_f32_16x16x32_bf16");
    return e == 0 ? 0 : 1;
}
