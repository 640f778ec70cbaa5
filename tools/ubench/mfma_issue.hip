// Microbenchmark: is the correlation backward's inner loop cheaper as 4x4x1 fp32 MFMA rank-1
// updates (16 blocks of 4 channels x 4 pixels per instruction, 256 FMAs) than as v_pk_fma_f32
// (128 FMAs per instruction)?  Both loops read their window operand from LDS the way the real
// kernels would and keep the gradOutput operand in registers.
//   A: per "step" = 4 channels x 64 pixels:  9 rows x (3 ds_read_b128 + 12 mfma)      [108 G regs]
//   B: per "step" = 1 channel  x 128 pixels: 9 rows x (5 ds_read_b64 + 8 pk_fma + 2 fma) [162 G regs]
// Reports ns per (pixel*channel) at a given waves/SIMD.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_mfma(float *out, const float *in, int steps) {
    __shared__ __attribute__((aligned(16))) float win[4 * 16 * 80];
    for (int i = threadIdx.x; i < 4 * 16 * 80; i += 256) win[i] = in[i % 1024];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[9][12];
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int t = 0; t < 12; ++t) g[r][t] = in[(lane * 7 + r * 12 + t) & 1023];
    float4v acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    // lane = (block b = lane/4, t = lane%4): channel t of the chunk, block's 12-float segment
    const float *base = win + (lane & 3) * (16 * 80) + wave * 80 + (lane >> 2) * 4;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const float4v a0 = *reinterpret_cast<const float4v *>(base + r * 80);
            const float4v a1 = *reinterpret_cast<const float4v *>(base + r * 80 + 4);
            const float4v a2 = *reinterpret_cast<const float4v *>(base + r * 80 + 8);
            const float a[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
            for (int t = 0; t < 12; ++t)
                acc[r % 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t], g[r][t], acc[r % 3], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    const float4v t = acc[0] + acc[1] + acc[2];
    out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y + t.z + t.w;
}

template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_pk(float *out, const float *in, int steps) {
    __shared__ __attribute__((aligned(16))) float win[2 * 16 * 72];
    for (int i = threadIdx.x; i < 2 * 16 * 72; i += 256) win[i] = in[i % 1024];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2v g0p[9][4], g1p[9][4];
    float g0s[9], g1s[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g0p[r][j] = float2v{in[(lane + r * 9 + j) & 1023], in[(lane * 3 + r + j) & 1023]};
            g1p[r][j] = float2v{in[(lane * 5 + r * 9 + j) & 1023], in[(lane * 11 + r + j) & 1023]};
        }
        g0s[r] = in[(lane + r) & 1023]; g1s[r] = in[(lane * 2 + r) & 1023];
    }
    float acc0 = 0.f;
    const float *wbase = win + (wave * 2 + lane / 32) * 72 + 2 * (lane % 32);
    typedef const volatile __attribute__((address_space(3))) float2v *lp;
    for (int s = 0; s < steps; ++s) {
        const float *wp = wbase + (s & 1) * (16 * 72);
        float2v a0[3] = {{0, 0}, {0, 0}, {0, 0}}, a1[3] = {{0, 0}, {0, 0}, {0, 0}};
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            float2v w[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) w[q] = *(lp)(wp + r * 72 + 2 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0[r % 3] = __builtin_elementwise_fma(g0p[r][j], w[j], a0[r % 3]);
                a1[r % 3] = __builtin_elementwise_fma(g1p[r][j], w[j + 1], a1[r % 3]);
            }
            s0 = fmaf(g0s[r], w[4].x, s0);
            s1 = fmaf(g1s[r], w[0].y, s1);
        }
        const float2v t0 = a0[0] + a0[1] + a0[2], t1 = a1[0] + a1[1] + a1[2];
        acc0 += t0.x + t0.y + s0 + t1.x + t1.y + s1;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0;
}

template <typename F> float time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    float *out, *in; hipMalloc(&out, 256 * 4096 * 4); hipMalloc(&in, 4096);
    hipMemset(in, 0, 4096);
    const int steps = 4096;
    printf("waves/SIMD   mfma4x4x1: ns per 1e3 px*ch    pk_fma: ns per 1e3 px*ch   (per CU, lower is better)\n");
#define RUN(W) { \
    float t1 = time_ms([&] { hipLaunchKernelGGL(k_mfma<W>, dim3(256 * W), dim3(256), 0, 0, out, in, steps); }); \
    float t2 = time_ms([&] { hipLaunchKernelGGL(k_pk<(W > 2 ? 2 : W)>, dim3(256 * (W > 2 ? 2 : W)), dim3(256), 0, 0, out, in, steps); }); \
    double pc1 = double(W) * 4 * steps * 4 * 64;      /* px*ch per CU */ \
    double pc2 = double(W > 2 ? 2 : W) * 4 * steps * 128; \
    printf("%d (pk at %d)  %8.3f ms %8.2f      %8.3f ms %8.2f\n", W, W > 2 ? 2 : W, t1, t1 * 1e6 / pc1 * 1e3, t2, t2 * 1e6 / pc2 * 1e3); }
    RUN(1) RUN(2) RUN(3)
    return 0;
}
