// Microbenchmark (round 4): does a ds_add_u64 cost less when part of its lanes are exec-masked?
// Decides whether combining neighbouring lanes' taps before the atomic (fewer ACTIVE lanes, same
// instruction count) can shorten the warp backward's add phase.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, int iters, unsigned long long mask_lo_hi, int dummy) {
    __shared__ unsigned long long acc[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = (mask_lo_hi >> lane) & 1ull;
    int idx = threadIdx.x & 4095;
    unsigned long long v = threadIdx.x + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = (idx + u * 257) & 4095;
            if (on) atomicAdd(&acc[a], v);
        }
        idx = (idx + 64) & 4095;
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc[threadIdx.x] + dummy;
}
int main() {
    float *out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 2000, blocks = 512;
    struct { const char *name; unsigned long long m; } cases[] = {
        {"all 64 lanes", ~0ull}, {"even lanes (32)", 0x5555555555555555ull}, {"lanes 0-31", 0xffffffffull},
        {"every 4th (16)", 0x1111111111111111ull}, {"lanes 0-15", 0xffffull}, {"one lane", 1ull}, {"every 8th (8)", 0x0101010101010101ull}};
    for (auto &c : cases) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, c.m, 0);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, c.m, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double instr = double(blocks) * 4 * iters * 8;   // wave-instructions
        printf("%-18s %.3f ms  %.1f cycles per wave-instruction per CU (2.4 GHz)\n", c.name, ms,
               ms * 1e-3 * 2.4e9 / (instr / 256));
    }
    return 0;
}
