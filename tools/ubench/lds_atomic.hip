// Microbenchmark: LDS float atomic add (ds_add_f32) vs plain ds_write_b32 throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void ki(float *out, int iters, int stride, int wide) {
    __shared__ unsigned long long acc[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    int idx = (threadIdx.x * stride) & 4095;
    unsigned long long v = threadIdx.x + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = (idx + u * 257) & 4095;
            if (wide) atomicAdd(&acc[a], v);
            else atomicAdd(reinterpret_cast<unsigned int *>(&acc[a]), (unsigned int)v);
        }
        idx = (idx + 64) & 4095;
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc[threadIdx.x];
}
template <int MODE>
__global__ void k(float *out, int iters, int stride) {
    __shared__ float acc[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    int idx = (threadIdx.x * stride) & 8191;
    float v = threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = (idx + u * 257) & 8191;
            if (MODE == 0) atomicAdd(&acc[a], v);
            else if (MODE == 1) acc[a] = v + u;
            else v += acc[a];
        }
        idx = (idx + 64) & 8191;
        if (MODE == 1) asm volatile("" ::: "memory");
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x] + v;
}
template <int MODE> float run(float *out, int iters, int stride, int blocks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float *out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 2000, blocks = 512;   // 2 blocks per CU
    for (int stride : {1, 2, 33}) {
        float t0 = run<0>(out, iters, stride, blocks), t1 = run<1>(out, iters, stride, blocks), t2 = run<2>(out, iters, stride, blocks);
        double ops = double(blocks) * 256 * iters * 8;
        printf("stride %2d: ds_add_f32 %.3f ms (%.1f lanes/clk/CU)  ds_write %.3f ms (%.1f)  ds_read %.3f ms (%.1f)\n", stride,
               t0, ops / (t0 * 1e-3) / 2.4e9 / 256, t1, ops / (t1 * 1e-3) / 2.4e9 / 256, t2, ops / (t2 * 1e-3) / 2.4e9 / 256);
    }
    for (int wide : {0, 1}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(ki, dim3(blocks), dim3(256), 0, 0, out, iters, 1, wide);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(ki, dim3(blocks), dim3(256), 0, 0, out, iters, 1, wide);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double ops = double(blocks) * 256 * iters * 8;
        printf("ds_add_u%d: %.3f ms (%.1f lanes/clk/CU)\n", wide ? 64 : 32, ms, ops / (ms * 1e-3) / 2.4e9 / 256);
    }
    return 0;
}
