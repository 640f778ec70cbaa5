// Round-3 probes for the strip-DPP correlation kernels (gfx950):
//   1. semantics of wave_shr:1 / wave_shl:1 / row_shr:1 DPP with bound_ctrl on v_fmac_f32
//   2. issue rate of v_fmac_f32 (plain, DPP, mixed 16+20 as in the kernel) and v_pk_fma_f32 vs waves per SIMD
//   3. buffer_load_dwordx4 from a dword-aligned (not 16-byte aligned) offset, per-dword range check
//   4. the same through LDS-DMA (buffer_load_dwordx4 ... lds)
//   5. L2-resident streaming read rate, 16 B per lane, direct to registers
//   6. (argument "unaligned_lds") ds_read_b128 from a dword-aligned LDS address -- may fault, run last
// Build: hipcc -O3 --offload-arch=gfx950 dpp_probe.hip -o dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *lds_void_ptr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// ---- 1. semantics -------------------------------------------------------------------------
__global__ void k_sem(const float *x, const float *g, float *out) {
    const int l = threadIdx.x;
    float xv = x[l], gv = g[l];
    float a0 = 100.f, a1 = 100.f, a2 = 100.f, a3 = 100.f;
    asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0) : "v"(xv), "v"(gv));
    asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a1) : "v"(xv), "v"(gv));
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a2) : "v"(xv), "v"(gv));
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a3) : "v"(xv), "v"(gv));
    out[l] = a0; out[64 + l] = a1; out[128 + l] = a2; out[192 + l] = a3;
}

// ---- 2. rates -------------------------------------------------------------------------------
// MODE 0: 36 plain v_fmac  1: 36 wave_shr DPP  2: 16 plain + 10 shr + 10 shl (the kernel's mix)
// MODE 3: 18 v_pk_fma_f32 (= 36 FMAs)          4: 36 row_shr DPP
template <int MODE>
__global__ void k_rate(float *out, const float *in, int iters, unsigned long long *clk) {
    float acc[16];
    float2v accp[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) accp[i] = float2v{threadIdx.x * 1e-3f + i, 1.0f * i};
    float x[4], g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = in[threadIdx.x * 4 + i]; g[i] = in[1024 + threadIdx.x * 4 + i]; }
    float2v xp = {x[0], x[1]}, gp = {g[0], g[1]};
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 36; ++k) {
            const int a = k % 16, xi = k % 4, gi = (k / 4) % 4;
            if (MODE == 0) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[a]) : "v"(x[xi]), "v"(g[gi]));
            if (MODE == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc[a]) : "v"(x[xi]), "v"(g[gi]));
            if (MODE == 4) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc[a]) : "v"(x[xi]), "v"(g[gi]));
            if (MODE == 2) {
                if (k < 16) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[a]) : "v"(x[xi]), "v"(g[gi]));
                else if (k < 26) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc[a]) : "v"(x[xi]), "v"(g[gi]));
                else asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc[a]) : "v"(x[xi]), "v"(g[gi]));
            }
            if (MODE == 3 && k < 18) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(accp[k % 8]) : "v"(xp), "v"(gp));
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += accp[i].x + accp[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- 3 / 4. unaligned buffer loads, per-dword range check ---------------------------------
__global__ void k_buf(const float *src, float *out, int nbytes, int shift) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int l = threadIdx.x;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, nbytes, 0x00020000);
    const int voff = (4 * l + shift) * 4;    // negative for lane 0 when shift < 0: wraps to a huge unsigned offset
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef int i4 __attribute__((ext_vector_type(4)));
    i4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    f4 vf = __builtin_bit_cast(f4, v);
    out[4 * l + 0] = vf.x; out[4 * l + 1] = vf.y; out[4 * l + 2] = vf.z; out[4 * l + 3] = vf.w;
    // through LDS-DMA
    smem[4 * l] = -7.f; smem[4 * l + 1] = -7.f; smem[4 * l + 2] = -7.f; smem[4 * l + 3] = -7.f;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = 0; q < 4; ++q) out[256 + 4 * l + q] = smem[4 * l + q];
}

// ---- 5. L2-resident streaming reads ---------------------------------------------------------
// every workgroup sweeps a region shared by the workgroups of its XCD (blockIdx % 8), `bytes_per_xcd` long
template <int NIF>
__global__ void k_l2(const float4 *src, float *out, int bytes_per_xcd, int sweeps) {
    const int xcd = blockIdx.x % 8;
    const float4 *base = src + (size_t)xcd * (bytes_per_xcd / 16);
    const int n16 = bytes_per_xcd / 16;
    const int nth = blockDim.x * (gridDim.x / 8);
    int idx = (blockIdx.x / 8) * blockDim.x + threadIdx.x;
    float4 acc = {0, 0, 0, 0};
    const int steps = sweeps * (n16 / nth) / NIF;
    for (int s = 0; s < steps; ++s) {
        float4 v[NIF];
#pragma unroll
        for (int q = 0; q < NIF; ++q) {
            { typedef float f4n __attribute__((ext_vector_type(4))); f4n t_ = __builtin_nontemporal_load(reinterpret_cast<const f4n *>(base + idx)); v[q] = make_float4(t_.x, t_.y, t_.z, t_.w); }   // nt load
            idx += nth; if (idx >= n16) idx -= n16;
        }
#pragma unroll
        for (int q = 0; q < NIF; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int NIF>
__global__ void k_l2_plain(const float4 *src, float *out, int bytes_per_xcd, int sweeps) {
    const int xcd = blockIdx.x % 8;
    const float4 *base = src + (size_t)xcd * (bytes_per_xcd / 16);
    const int n16 = bytes_per_xcd / 16;
    const int nth = blockDim.x * (gridDim.x / 8);
    int idx = (blockIdx.x / 8) * blockDim.x + threadIdx.x;
    float4 acc = {0, 0, 0, 0};
    const int steps = sweeps * (n16 / nth) / NIF;
    for (int s = 0; s < steps; ++s) {
        float4 v[NIF];
#pragma unroll
        for (int q = 0; q < NIF; ++q) {
            v[q] = base[idx];
            idx += nth; if (idx >= n16) idx -= n16;
        }
#pragma unroll
        for (int q = 0; q < NIF; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// ---- 6. unaligned ds_read_b128 -----------------------------------------------------------------
__global__ void k_ulds(float *out, int shift) {
    __shared__ __attribute__((aligned(16))) float sm[64 * 4 + 16];
    const int l = threadIdx.x;
    for (int q = 0; q < 4; ++q) sm[4 * l + q] = 4 * l + q;
    if (l < 16) sm[256 + l] = 256 + l;
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(sm) + (4 * l + shift) * 4;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    out[4 * l] = v.x; out[4 * l + 1] = v.y; out[4 * l + 2] = v.z; out[4 * l + 3] = v.w;
}

template <typename F> float time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

int main(int argc, char **argv) {
    const bool ulds = argc > 1 && !strcmp(argv[1], "unaligned_lds");
    float *out, *in; unsigned long long *clk;
    CK(hipMalloc(&out, 256 * 2048 * sizeof(float) * 4));
    CK(hipMalloc(&in, 1 << 20));
    CK(hipMalloc(&clk, 64));
    std::vector<float> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0f + (i % 97) * 0.001f;
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));

    if (ulds) {
        for (int shift : {0, 1, 2, 3}) {
            hipLaunchKernelGGL(k_ulds, dim3(1), dim3(64), 0, 0, out, shift);
            hipError_t e = hipDeviceSynchronize();
            std::vector<float> o(256);
            hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) if (o[4 * l + q] != 4 * l + q + shift) ++bad;
            printf("unaligned ds_read_b128 shift %d: err=%d bad=%d  lane1: %g %g %g %g\n", shift, (int)e, bad, o[4], o[5], o[6], o[7]);
        }
        return 0;
    }

    // 1. semantics
    {
        std::vector<float> x(64), g(64), o(256);
        for (int l = 0; l < 64; ++l) { x[l] = l + 1; g[l] = 1.f; }
        float *dx, *dg; CK(hipMalloc(&dx, 256)); CK(hipMalloc(&dg, 256));
        hipMemcpy(dx, x.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dg, g.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dx, dg, out);
        CK(hipDeviceSynchronize());
        hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
        const char *names[4] = {"wave_shr:1", "wave_shl:1", "row_shr:1", "row_shl:1"};
        for (int m = 0; m < 4; ++m) {
            printf("%-10s (acc = 100 + src*1; x[l] = l+1): lanes 0,1,15,16,17,31,32,47,48,62,63 ->", names[m]);
            for (int l : {0, 1, 15, 16, 17, 31, 32, 47, 48, 62, 63}) printf(" %g", o[64 * m + l] - 100.f);
            printf("\n");
        }
    }

    // 2. rates
    {
        const int iters = 20000;
        printf("\nFMA issue rate (36 FMAs per iteration, 16 accumulators), FMA/clk/CU by the in-kernel clock\n");
        printf("waves/SIMD | plain v_fmac | wave_shr dpp | mix 16+10+10 | v_pk_fma | row_shr dpp | clock GHz\n");
        for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
            const int threads = 256, blocks = 256 * wps;
            double r[5], ghz = 0;
            for (int m = 0; m < 5; ++m) {
                auto launch = [&] {
                    if (m == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(threads), 0, 0, out, in, iters, clk);
                    if (m == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(threads), 0, 0, out, in, iters, clk);
                    if (m == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(threads), 0, 0, out, in, iters, clk);
                    if (m == 3) hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(threads), 0, 0, out, in, iters, clk);
                    if (m == 4) hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(threads), 0, 0, out, in, iters, clk);
                };
                float ms = time_ms(launch, 3);
                unsigned long long c[2];
                hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
                const double clock = double(c[0]) / (double(c[1]) / 100e6);   // Hz
                ghz = clock / 1e9;
                const double fma = double(blocks) * threads * 36.0 * iters;
                r[m] = fma / (ms * 1e-3) / clock / 256;
            }
            printf("%d | %7.1f | %7.1f | %7.1f | %7.1f | %7.1f | %.2f\n", wps, r[0], r[1], r[2], r[3], r[4], ghz);
        }
    }

    // 3 / 4. unaligned buffer loads
    {
        float *src; CK(hipMalloc(&src, 4096));
        std::vector<float> s(1024);
        for (int i = 0; i < 1024; ++i) s[i] = i;
        hipMemcpy(src, s.data(), 4096, hipMemcpyHostToDevice);
        for (int shift : {0, -3, -1, 2, 3}) {
            hipLaunchKernelGGL(k_buf, dim3(1), dim3(64), 1024, 0, src, out, 1024 /* 256 floats = the 64 strips */, shift);
            hipError_t e = hipDeviceSynchronize();
            std::vector<float> o(512);
            hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost);
            int bad_r = 0, bad_l = 0;
            for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
                const int e_ = 4 * l + q + shift;
                const float want = (e_ >= 0 && e_ < 256) ? e_ : 0.f;
                if (o[4 * l + q] != want) ++bad_r;
                if (o[256 + 4 * l + q] != want) ++bad_l;
            }
            printf("buffer_load_dwordx4 shift %+d floats: err=%d  to-VGPR bad=%d  LDS-DMA bad=%d | lane0 reg %g %g %g %g lds %g %g %g %g | lane63 reg %g %g %g %g lds %g %g %g %g\n",
                   shift, (int)e, bad_r, bad_l, o[0], o[1], o[2], o[3], o[256], o[257], o[258], o[259],
                   o[252], o[253], o[254], o[255], o[508], o[509], o[510], o[511]);
        }
    }

    // 5. L2 streaming
    {
        float4 *big; CK(hipMalloc(&big, 64 << 20));
        hipMemset(big, 0, 64 << 20);
        printf("\nL2-resident streaming reads (16 B/lane), TB/s; region per XCD | waves/CU | loads in flight | nt | plain\n");
        for (int mb : {1, 2}) for (int wpc : {8, 16, 32}) {
            const int threads = 256, blocks = 256 * wpc / 4;
            const int bytes = mb << 20, sweeps = 64;
            const int nth = threads * (blocks / 8);
            for (int nif : {2, 4, 8}) {
                auto l1 = [&] {
                    if (nif == 2) hipLaunchKernelGGL(k_l2<2>, dim3(blocks), dim3(threads), 0, 0, big, out, bytes, sweeps);
                    if (nif == 4) hipLaunchKernelGGL(k_l2<4>, dim3(blocks), dim3(threads), 0, 0, big, out, bytes, sweeps);
                    if (nif == 8) hipLaunchKernelGGL(k_l2<8>, dim3(blocks), dim3(threads), 0, 0, big, out, bytes, sweeps);
                };
                auto l2 = [&] {
                    if (nif == 2) hipLaunchKernelGGL(k_l2_plain<2>, dim3(blocks), dim3(threads), 0, 0, big, out, bytes, sweeps);
                    if (nif == 4) hipLaunchKernelGGL(k_l2_plain<4>, dim3(blocks), dim3(threads), 0, 0, big, out, bytes, sweeps);
                    if (nif == 8) hipLaunchKernelGGL(k_l2_plain<8>, dim3(blocks), dim3(threads), 0, 0, big, out, bytes, sweeps);
                };
                const double total = double(sweeps) * (bytes / 16 / nth) / nif * nif * 16.0 * nth * 8;  // bytes read chip-wide
                float t1 = time_ms(l1, 3), t2 = time_ms(l2, 3);
                printf("%d MiB | %2d | %d | %6.2f | %6.2f\n", mb, wpc, nif, total / (t1 * 1e-3) / 1e12, total / (t2 * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
