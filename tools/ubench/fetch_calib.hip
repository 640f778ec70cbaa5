// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths this package uses.
// MI355X_MICROARCH.md (HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
// stream (16 B per lane); "other access widths are uncalibrated".  The warp kernels gather with
// 4-byte (and unaligned 8-byte) loads, so their "x2" correction needs its own calibration:
// each kernel below reads a known 1 GiB once (larger than the 256 MiB Infinity Cache).
//   hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read_x4(const float4 *p, float *out, size_t n4) {     // 16 B per lane, coalesced
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.f) out[0] = acc;
}
__global__ void read_x1(const float *p, float *out, size_t n) {        // 4 B per lane, coalesced
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += p[i];
    if (acc == 12345.f) out[0] = acc;
}
struct __attribute__((packed, aligned(4))) f2u { float a, b; };
__global__ void read_x2u(const float *p, float *out, size_t n) {       // 8 B per lane at 4-byte alignment (+1)
    float acc = 0.f;
    for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 2 + 1; i + 1 < n;
         i += (size_t)gridDim.x * blockDim.x * 2) {
        const f2u v = *reinterpret_cast<const f2u *>(p + i);
        acc += v.a + v.b;
    }
    if (acc == 12345.f) out[0] = acc;
}
__global__ void read_rows(const float *p, float *out, size_t n, int W) {   // bilinear-like: 2 rows x (64 + jitter)
    float acc = 0.f;
    const size_t rows = n / W;
    for (size_t r = blockIdx.x; r + 1 < rows; r += gridDim.x) {
        const int x = threadIdx.x + ((threadIdx.x * 7) & 3);      // gaps and repeats inside a row segment
        if (x + 1 < W) acc += p[r * W + x] + p[r * W + x + 1] + p[(r + 1) * W + x] + p[(r + 1) * W + x + 1];
    }
    if (acc == 12345.f) out[0] = acc;
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 4;
    float *p, *out;
    hipMalloc(&p, bytes); hipMalloc(&out, 4);
    hipMemset(p, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read_x4, dim3(2048), dim3(256), 0, 0, (const float4 *)p, out, n / 4);
        hipLaunchKernelGGL(read_x1, dim3(2048), dim3(256), 0, 0, p, out, n);
        hipLaunchKernelGGL(read_x2u, dim3(2048), dim3(256), 0, 0, p, out, n);
        hipLaunchKernelGGL(read_rows, dim3(4096), dim3(64), 0, 0, p, out, n, 256);
        hipDeviceSynchronize();
    }
    printf("each kernel read %zu bytes once (read_rows: %zu rows of 1 KiB, each row twice through L1/L2)\n", bytes, n / 256);
    return 0;
}
