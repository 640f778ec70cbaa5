// Microbenchmark: the floor of a short kernel inside a hipGraph of back-to-back launches on gfx950,
// by grid shape, LDS allocation and number of independent 16-byte loads per lane before one store.
// Build: hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NL>
__global__ void k_load(const f4 *__restrict__ src, f4 *__restrict__ dst, int mask) {
    extern __shared__ float smem[];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NL > 0) {
        f4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = src[(gid + i * 8192) & mask];
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i];
        dst[gid & mask] = acc;
    }
}

// load -> LDS -> barrier -> read a neighbour's value -> store (one dependent round trip through LDS)
__global__ void k_lds(const f4 *__restrict__ src, f4 *__restrict__ dst, int mask) {
    extern __shared__ float smem[];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    f4 v = src[gid & mask];
    reinterpret_cast<f4 *>(smem)[threadIdx.x] = v;
    __syncthreads();
    f4 w = reinterpret_cast<f4 *>(smem)[(threadIdx.x + 64) % blockDim.x];
    dst[gid & mask] = v + w;
}

template <typename F>
float graph_us(F launch, hipStream_t s, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int r = 0; r < reps; ++r) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 7; ++it) {
        hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best * 1e3f / reps;
}

int main() {
    const int n = 1 << 22;   // 64 MiB of float4
    f4 *src, *dst; hipMalloc(&src, n * sizeof(f4)); hipMalloc(&dst, n * sizeof(f4));
    hipMemset(src, 0, n * sizeof(f4));
    hipStream_t s; hipStreamCreate(&s);
    const int mask = (1 << 18) - 1;   // 4 MiB footprint: L2 / MALL resident after the first launch
    struct Cfg { int wgs, threads, lds; };
    const Cfg cfgs[] = {{128, 640, 0}, {128, 640, 61440}, {256, 640, 61440}, {576, 64, 0}, {1152, 64, 0},
                        {2304, 64, 0}, {192, 768, 0}, {256, 256, 0}, {512, 256, 0}, {1024, 256, 0}, {512, 512, 32768}};
    printf("%5s %7s %6s | %7s %7s %7s %7s %7s %7s   (us per launch, 20 launches per graph, min of 7)\n", "wgs", "threads", "lds",
           "empty", "1 load", "8 ld", "16 ld", "32 ld", "ld-lds");
    for (const Cfg &c : cfgs) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_load<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_load<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_load<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_load<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_load<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        const int lds_min = c.threads * 16;
        float t0 = graph_us([&] { hipLaunchKernelGGL(k_load<0>, dim3(c.wgs), dim3(c.threads), c.lds, s, src, dst, mask); }, s, 20);
        float t1 = graph_us([&] { hipLaunchKernelGGL(k_load<1>, dim3(c.wgs), dim3(c.threads), c.lds, s, src, dst, mask); }, s, 20);
        float t8 = graph_us([&] { hipLaunchKernelGGL(k_load<8>, dim3(c.wgs), dim3(c.threads), c.lds, s, src, dst, mask); }, s, 20);
        float t16 = graph_us([&] { hipLaunchKernelGGL(k_load<16>, dim3(c.wgs), dim3(c.threads), c.lds, s, src, dst, mask); }, s, 20);
        float t32 = graph_us([&] { hipLaunchKernelGGL(k_load<32>, dim3(c.wgs), dim3(c.threads), c.lds, s, src, dst, mask); }, s, 20);
        float tl = graph_us([&] { hipLaunchKernelGGL(k_lds, dim3(c.wgs), dim3(c.threads), c.lds > lds_min ? c.lds : lds_min, s, src, dst, mask); }, s, 20);
        printf("%5d %7d %6d | %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f\n", c.wgs, c.threads, c.lds, t0, t1, t8, t16, t32, tl);
    }
    return 0;
}
