// corr_d4_common.h -- what the tuned d = 4 correlation kernels share: corr_d4.hip (forward) and corr_d4_bwd.hip (the tile
// backward kernels: the fallbacks for maps the strip / coarse kernels do not cover).  Internal to those two translation units.
#pragma once
#include <atomic>
#include <type_traits>

#include "common.h"

namespace cerb {
namespace {

constexpr int kD = 4;            // max displacement
constexpr int kND = 2 * kD + 1;  // 9 displacements per axis
constexpr int kP = 4;            // pixels per lane in forward (one float4)

__host__ __device__ constexpr int pad_to_residue(int x, int res) {
    return x + ((res - x % 64) + 64) % 64;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// ---- global-memory element types ------------------------------------------------------
// fp16 / bf16 are STORAGE formats: values are widened when staged into LDS / registers,
// every product and sum is fp32 (the reference accumulates fp16 in fp16),
// results are rounded once on the way out.  LDS always holds fp32.
template <typename T> struct Gmem;
template <> struct Gmem<float> {
    static __device__ __forceinline__ float load1(const float *p) { return *p; }
    static __device__ __forceinline__ float2 load2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
    static __device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void store1(float *p, float v) { *p = v; }
    static __device__ __forceinline__ void store2(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
    static __device__ __forceinline__ void store4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    // streaming store (written once, not re-read by the kernel): leaves L2 to the halo lines
    static __device__ __forceinline__ void stream2(float *p, float a, float b) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(f2v{a, b}, reinterpret_cast<f2v *>(p));
    }
};
template <> struct Gmem<__half> {
    static __device__ __forceinline__ void stream2(__half *p, float a, float b) { store2(p, a, b); }
    static __device__ __forceinline__ float load1(const __half *p) { return __half2float(*p); }
    static __device__ __forceinline__ float2 load2(const __half *p) { return __half22float2(*reinterpret_cast<const __half2 *>(p)); }
    static __device__ __forceinline__ float4 load4(const __half *p) {
        const uint2 raw = *reinterpret_cast<const uint2 *>(p);
        const float2 lo = __half22float2(*reinterpret_cast<const __half2 *>(&raw.x));
        const float2 hi = __half22float2(*reinterpret_cast<const __half2 *>(&raw.y));
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    static __device__ __forceinline__ void store1(__half *p, float v) { *p = __float2half(v); }
    static __device__ __forceinline__ void store2(__half *p, float a, float b) { *reinterpret_cast<__half2 *>(p) = __floats2half2_rn(a, b); }
    static __device__ __forceinline__ void store4(__half *p, float4 v) {
        uint2 raw;
        *reinterpret_cast<__half2 *>(&raw.x) = __floats2half2_rn(v.x, v.y);
        *reinterpret_cast<__half2 *>(&raw.y) = __floats2half2_rn(v.z, v.w);
        *reinterpret_cast<uint2 *>(p) = raw;
    }
};
template <> struct Gmem<hip_bfloat16> {
    static __device__ __forceinline__ void stream2(hip_bfloat16 *p, float a, float b) { store2(p, a, b); }
    static __device__ __forceinline__ float widen(unsigned short b) { return __uint_as_float(static_cast<unsigned int>(b) << 16); }
    static __device__ __forceinline__ unsigned short narrow(float v) {
        const __bf16 h = static_cast<__bf16>(v);  // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
        unsigned short b;
        __builtin_memcpy(&b, &h, 2);
        return b;
    }
    static __device__ __forceinline__ float load1(const hip_bfloat16 *p) { return widen(*reinterpret_cast<const unsigned short *>(p)); }
    static __device__ __forceinline__ float2 load2(const hip_bfloat16 *p) {
        const unsigned int raw = *reinterpret_cast<const unsigned int *>(p);
        return make_float2(__uint_as_float(raw << 16), __uint_as_float(raw & 0xFFFF0000u));
    }
    static __device__ __forceinline__ float4 load4(const hip_bfloat16 *p) {
        const uint2 raw = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xFFFF0000u),
                           __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xFFFF0000u));
    }
    static __device__ __forceinline__ void store1(hip_bfloat16 *p, float v) { *reinterpret_cast<unsigned short *>(p) = narrow(v); }
    // two floats -> packed bf16 pair: gfx950's v_cvt_pk_bf16_f32 (round to nearest even, NaN
    // preserved), one instruction instead of the ~10 of the software rounding per value
    static __device__ __forceinline__ unsigned int narrow2(float a, float b) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned int, __builtin_convertvector(f2v{a, b}, bf2v));
    }
    static __device__ __forceinline__ void store2(hip_bfloat16 *p, float a, float b) {
        *reinterpret_cast<unsigned int *>(p) = narrow2(a, b);
    }
    static __device__ __forceinline__ void store4(hip_bfloat16 *p, float4 v) {
        uint2 raw;
        raw.x = narrow2(v.x, v.y);
        raw.y = narrow2(v.z, v.w);
        *reinterpret_cast<uint2 *>(p) = raw;
    }
};

typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v pkfma(float2v a, float2v b, float2v c) {
    return __builtin_elementwise_fma(a, b, c);  // v_pk_fma_f32
}
__device__ __forceinline__ float2v ld2v(const float *p) { return *reinterpret_cast<const float2v *>(p); }
// volatile: keeps hipcc's load/store optimizer from fusing neighbouring 8-byte LDS reads
// into ds_read2_b64, which runs at HALF the LDS rate of ds_read_b64 on gfx950
// (MI355X_MICROARCH.md LDS table: 8 vs 2 cycles per wave-instruction for 2x/1x 512 B)
typedef const volatile __attribute__((address_space(3))) float2v *lds_f2_volatile_ptr;
__device__ __forceinline__ float2v ld2v_nomerge(const float *p) {
    return *(lds_f2_volatile_ptr)(p);  // explicit LDS address space: stays a ds_read_b64
}

// 16 bytes of zeros in device memory: the source of every halo / padding slot, so
// that staging loads are UNCONDITIONAL.  (A load under a branch makes hipcc's
// waitcnt pass fall back to vmcnt(0) at the next use, which serialises the
// prefetch pipeline -- seen in the ISA of the first version.)
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef __attribute__((address_space(3))) void *lds_void_ptr;
typedef const __attribute__((address_space(1))) void *gbl_void_ptr;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


bool fast_config(const CorrGeom &g, int dtype) {
    return (dtype == CERB_F32 || dtype == CERB_F16 || dtype == CERB_BF16) && g.pad == kD &&
           g.maxd == kD && g.ksize == 1 && g.s1 == 1 && g.s2 == 1 &&
           static_cast<int64_t>(g.C) * g.H * g.W < (1ll << 30) &&
           static_cast<int64_t>(kND * kND) * g.H * g.W < (1ll << 30);  // 32-bit offsets
}

// The LDS-DMA kernels address a batch item through 32-bit buffer offsets and use 2^31 as the
// "out of range" offset: a batch item (and its 81-plane gradOutput) must stay below 2 GiB.
// Larger items keep the register-staged kernels (64-bit pointers, up to 2^30 elements).
bool dma_ok(const CorrGeom &g) {
    return static_cast<int64_t>(g.C) * g.H * g.W < (1ll << 29) &&
           static_cast<int64_t>(kND * kND) * g.H * g.W < (1ll << 29);
}

// a 4-element group must be naturally aligned: 16 B (fp32) or 8 B (16-bit storage)
bool aligned_group(const void *p, int dtype) {
    return (reinterpret_cast<uintptr_t>(p) & (dtype == CERB_F32 ? 15 : 7)) == 0;
}


}  // namespace
}  // namespace cerb
