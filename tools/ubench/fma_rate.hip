// Microbenchmark: f32 FMA issue rate on gfx950, scalar v_fma_f32 vs packed v_pk_fma_f32,
// as a function of waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 fma_rate.hip -o fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int CH>
__global__ void k_fma(float *out, float a, float b, int iters) {
    float acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
__global__ void k_pk(float *out, float a, float b, int iters) {
    float2v acc[CH];
    float2v av = {a, a * 1.0001f}, bv = {b, b * 0.9999f};
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = float2v{threadIdx.x * 1e-3f + i, 1.0f * i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    float *out; hipMalloc(&out, 256 * 2048 * sizeof(float) * 4);
    const int iters = 65536;
    constexpr int CH = 16;
    printf("waves/SIMD  fma: FMA/clk/CU(@2.4GHz)   pk_fma: FMA/clk/CU\n");
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 256;                 // 4 waves = 1 per SIMD
        const int blocks = 256 * wps;            // wps blocks per CU
        float t1 = time_ms([&] { hipLaunchKernelGGL(k_fma<CH>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); });
        float t2 = time_ms([&] { hipLaunchKernelGGL(k_pk<CH>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f, iters); });
        const double fma1 = double(blocks) * threads * CH * iters;       // FMAs
        const double fma2 = fma1 * 2;
        printf("%d  %8.3f ms %7.1f    %8.3f ms %7.1f\n", wps, t1, fma1 / (t1 * 1e-3) / 2.4e9 / 256, t2,
               fma2 / (t2 * 1e-3) / 2.4e9 / 256);
    }
    return 0;
}
