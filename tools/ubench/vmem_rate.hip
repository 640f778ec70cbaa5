// Microbenchmark (round 5): what a wave-level vector-memory LOAD instruction costs on gfx950 by width, when the data is
// L2-resident and every lane reads its own consecutive element of a plane row (the gradOutput pattern of the 16-bit
// correlation backward: 81 planes, lane = pixel), and whether buffer loads at 2-byte-aligned addresses work for the
// dword / two-dword widths (they do if the queue runs in unaligned mode).
// Build: hipcc -O3 --offload-arch=gfx950 vmem_rate.hip -o vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// NL loads per lane of WIDTH bytes each; plane p, element (row, lane); `shift` = byte offset added to every address
template <int WIDTH, int NL>
__global__ __launch_bounds__(256) void k(const unsigned short *__restrict__ src, unsigned *__restrict__ dst, int plane_bytes,
                                         int rows, int shift) {
    const __amdgpu_buffer_rsrc_t r = rsrc(src, plane_bytes * NL);
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6)) % rows, lane = threadIdx.x & 63;
    const int voff = wave * 64 * WIDTH + lane * WIDTH + shift;
    unsigned acc = 0;
    if constexpr (WIDTH == 2) {
        unsigned short v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b16(r, voff, i * plane_bytes, 0);
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i];
    } else if constexpr (WIDTH == 4) {
        unsigned v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, i * plane_bytes, 0);
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i];
    } else if constexpr (WIDTH == 8) {
        u2v v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b64(r, voff, i * plane_bytes, 0);
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i].x + v[i].y;
    } else {
        u4v v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, i * plane_bytes, 0);
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    dst[blockIdx.x * 256 + threadIdx.x] = acc;
}

// few active lanes: only lanes with (lane & 15) < nact issue the loads (the rest of the wave is masked off by EXEC)
template <int NL>
__global__ __launch_bounds__(256) void k_masked(const unsigned short *__restrict__ src, unsigned *__restrict__ dst, int plane_bytes,
                                                int rows, int nact) {
    const __amdgpu_buffer_rsrc_t r = rsrc(src, plane_bytes * NL);
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6)) % rows, lane = threadIdx.x & 63;
    const int voff = wave * 256 + lane * 4;
    unsigned acc = 0;
    if ((lane & 15) < nact) {
        unsigned v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, i * plane_bytes, 0);
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i];
    }
    dst[blockIdx.x * 256 + threadIdx.x] = acc;
}

// correctness of unaligned loads: out[i] = the dword / two dwords at byte offset 2 * i
__global__ void k_unaligned(const unsigned short *__restrict__ src, unsigned *__restrict__ o32, u2v *__restrict__ o64, int n) {
    const __amdgpu_buffer_rsrc_t r = rsrc(src, n * 2);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 4 <= n) {
        o32[i] = __builtin_amdgcn_raw_buffer_load_b32(r, 2 * i, 0, 0);
        o64[i] = __builtin_amdgcn_raw_buffer_load_b64(r, 2 * i, 0, 0);
    }
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

template <int WIDTH, int NL>
void run(const unsigned short *src, unsigned *dst, hipStream_t s, int shift, const char *what) {
    const int rows = 64;                              // 64 rows of 64 lanes: a plane of 64 * 64 * WIDTH bytes
    const int plane_bytes = rows * 64 * WIDTH + 256;
    const int blocks = 8192;                          // 32 workgroups of 4 waves per CU
    const float us = graph_us([&] { hipLaunchKernelGGL((k<WIDTH, NL>), dim3(blocks), dim3(256), 0, s, src, dst, plane_bytes, rows, shift); }, s, 10);
    const double instr_per_cu = double(blocks) * 4 * NL / 256.0;
    printf("%-28s width %2d B x %2d loads/lane: %7.1f us  -> %5.1f ns = %5.1f cycles (2.4 GHz) per wave-instruction per CU, %6.2f TB/s from L2\n",
           what, WIDTH, NL, us, us * 1e3 / instr_per_cu, us * 1e3 / instr_per_cu * 2.4,
           double(blocks) * 256 * NL * WIDTH / us / 1e6);
}

int main() {
    const size_t bytes = 64u << 20;
    unsigned short *src; unsigned *dst;
    hipMalloc(&src, bytes); hipMalloc(&dst, 8192 * 256 * 4);
    std::vector<unsigned short> h(bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = static_cast<unsigned short>(i * 2654435761u >> 13);
    hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    // ---- unaligned correctness ----
    {
        const int n = 1 << 16;
        unsigned *o32; u2v *o64; hipMalloc(&o32, n * 4); hipMalloc(&o64, n * 8);
        hipMemset(o32, 0, n * 4); hipMemset(o64, 0, n * 8);
        hipLaunchKernelGGL(k_unaligned, dim3(n / 256), dim3(256), 0, s, src, o32, o64, n);
        hipStreamSynchronize(s);
        std::vector<unsigned> a(n); std::vector<u2v> b(n);
        hipMemcpy(a.data(), o32, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o64, n * 8, hipMemcpyDeviceToHost);
        long bad32 = 0, bad64 = 0;
        for (int i = 0; i + 4 <= n; ++i) {
            const unsigned w0 = h[i] | (unsigned(h[i + 1]) << 16), w1 = h[i + 2] | (unsigned(h[i + 3]) << 16);
            if (a[i] != w0) ++bad32;
            if (b[i].x != w0 || b[i].y != w1) ++bad64;
        }
        printf("buffer loads at 2-byte-aligned offsets: b32 %s (%ld wrong of %d), b64 %s (%ld wrong)\n", bad32 ? "WRONG" : "correct", bad32, n,
               bad64 ? "WRONG" : "correct", bad64);
    }
    run<2, 81>(src, dst, s, 0, "halfword per lane");
    run<4, 81>(src, dst, s, 0, "dword per lane");
    run<4, 41>(src, dst, s, 0, "dword per lane");
    run<4, 41>(src, dst, s, 2, "dword per lane, +2 bytes");
    run<8, 21>(src, dst, s, 0, "two dwords per lane");
    run<8, 21>(src, dst, s, 2, "two dwords per lane, +2 B");
    run<8, 21>(src, dst, s, 4, "two dwords per lane, +4 B");
    run<16, 11>(src, dst, s, 0, "four dwords per lane");
    for (int nact : {16, 8, 2, 1}) {
        const int rows = 64, plane_bytes = rows * 256 + 256, blocks = 8192;
        const float us = graph_us([&] { hipLaunchKernelGGL((k_masked<41>), dim3(blocks), dim3(256), 0, s, src, dst, plane_bytes, rows, nact); }, s, 10);
        const double ipc = double(blocks) * 4 * 41 / 256.0;
        printf("dword per lane, %2d of every 16 lanes active x 41 loads: %7.1f us -> %5.1f cycles per wave-instruction per CU\n", nact, us,
               us * 1e3 / ipc * 2.4);
    }
    return 0;
}
