// What does a pure streaming kernel take for the byte counts of the correlation launches?  Reads R MB, writes
// W MB (16-byte accesses, non-temporal stores), 20 launches per hipGraph: on the same buffers (hot: Infinity
// Cache resident) and walking through > 256 MiB of buffers (cold).  The practical ceiling the roofline fractions
// of DESIGN.md can be read against.  Build: hipcc -O3 --offload-arch=gfx950 stream_sol.hip -o stream_sol
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_stream(const f4 *__restrict__ src, f4 *__restrict__ dst, long nr, long nw) {
    const long t = blockIdx.x * 256L + threadIdx.x, n = gridDim.x * 256L;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = t; i < nr; i += n) acc += src[i];
    for (long i = t; i < nw; i += n) __builtin_nontemporal_store(acc, dst + i);
}
// each thread: 4 loads in flight, then 5 stores (the shape of a correlation forward tile: 2 in, 81/32 out)
template <int NT>
__global__ __launch_bounds__(256) void k_stream4(const f4 *__restrict__ src, f4 *__restrict__ dst, long nr, long nw) {
    const long t = blockIdx.x * 256L + threadIdx.x, n = gridDim.x * 256L;
    f4 a[4] = {};
    for (long i = t; i < nr; i += 4 * n) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * n < nr) a[u] += src[i + u * n];
    }
    f4 acc = a[0] + a[1] + a[2] + a[3];
    for (long i = t; i < nw; i += n) {
        if constexpr (NT) __builtin_nontemporal_store(acc, dst + i);
        else dst[i] = acc;
    }
}

template <typename F>
float graph_us(F launch, hipStream_t s, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int r = 0; r < reps; ++r) launch(r);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 7; ++it) {
        (void)hipEventRecord(a, s); (void)hipGraphLaunch(ge, s); (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return best * 1e3f / reps;
}

int main() {
    const long slot = 96L << 20;          // bytes per rotating slot (read side and write side each); every case fits one slot
    const int nslot = 6;                  // 6 x 96 MiB x 2: nothing survives in the 256 MiB Infinity Cache
    f4 *src, *dst;
    (void)hipMalloc(&src, slot * nslot); (void)hipMalloc(&dst, slot * nslot);
    (void)hipMemset(src, 0, slot * nslot);
    hipStream_t s; (void)hipStreamCreate(&s);
    struct Case { const char *name; double rmb, wmb; };
    const Case cases[] = {{"corr_fwd L3 (33.5 in, 42.5 out)", 33.55, 42.47}, {"corr_bwd L3 (76.0 in, 33.5 out)", 76.0, 33.55},
                          {"corr_fwd L2 (16.8 in, 10.6 out)", 16.8, 10.6}, {"warp_bwd L3 (35 in, 17.8 out)", 35.0, 17.8},
                          {"copy 32 + 32", 32, 32}, {"read 64", 64, 0.001}, {"write 64", 0.001, 64}};
    printf("%-34s %6s | %8s %8s | %8s %8s   us per launch (TB/s), 20 launches per graph, min of 7\n", "bytes (MB)", "wgs", "hot", "", "cold", "");
    for (const Case &c : cases) {
        const long nr = (long)(c.rmb * 1e6 / 16), nw = (long)(c.wmb * 1e6 / 16);
        for (int wgs : {1024, 2048, 4096}) {
            float hot = graph_us([&](int) { hipLaunchKernelGGL(k_stream4<1>, dim3(wgs), dim3(256), 0, s, src, dst, nr, nw); }, s, 20);
            float cold = graph_us([&](int r) { hipLaunchKernelGGL(k_stream4<1>, dim3(wgs), dim3(256), 0, s, src + (r % nslot) * (slot / 16), dst + (r % nslot) * (slot / 16), nr, nw); }, s, 20);
            float hotp = graph_us([&](int) { hipLaunchKernelGGL(k_stream4<0>, dim3(wgs), dim3(256), 0, s, src, dst, nr, nw); }, s, 20);
            float coldp = graph_us([&](int r) { hipLaunchKernelGGL(k_stream4<0>, dim3(wgs), dim3(256), 0, s, src + (r % nslot) * (slot / 16), dst + (r % nslot) * (slot / 16), nr, nw); }, s, 20);
            const double tb = (c.rmb + c.wmb) * 1e6;
            printf("%-34s %6d | %8.2f (%4.2f) | %8.2f (%4.2f) | plain stores: %8.2f %8.2f\n", c.name, wgs, hot, tb / hot / 1e6, cold, tb / cold / 1e6, hotp, coldp);
        }
    }
    return 0;
}
