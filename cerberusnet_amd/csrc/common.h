// common.h -- shared device/host helpers for libcerberus_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>

#include "../../include/cerberus_hip.h"

namespace cerb {

// ---- storage type -> arithmetic type ---------------------------------------
// fp16/bf16 are storage formats only: every product is formed and accumulated
// in fp32 (the reference accumulates fp16 in fp16; we do better).
template <typename T> struct Acc { using type = float; };
template <> struct Acc<double> { using type = double; };

template <typename T> __device__ __forceinline__ typename Acc<T>::type ld(const T *p) {
    return static_cast<typename Acc<T>::type>(*p);
}
template <> __device__ __forceinline__ float ld<__half>(const __half *p) { return __half2float(*p); }
template <> __device__ __forceinline__ float ld<hip_bfloat16>(const hip_bfloat16 *p) {
    return static_cast<float>(*p);
}
template <typename T, typename A> __device__ __forceinline__ void st(T *p, A v) { *p = static_cast<T>(v); }
template <> __device__ __forceinline__ void st<__half, float>(__half *p, float v) { *p = __float2half(v); }
// gfx950's v_cvt_pk_bf16_f32 (one instruction) instead of the ~8 of hip_bfloat16(float)'s software rounding: identical
// bits for every non-NaN input, NaN stays NaN with a different payload (tools/ubench/cvt_check.hip: all 2^32 inputs)
template <> __device__ __forceinline__ void st<hip_bfloat16, float>(hip_bfloat16 *p, float v) {
    const __bf16 b = static_cast<__bf16>(v);
    __builtin_memcpy(p, &b, 2);
}

// ---- correlation geometry (correlation_cuda.cpp:6-14) -----------------------
struct CorrGeom {
    int B, C, H, W;
    int pad, ksize, maxd, s1, s2;
    int krad, drad, dsize;
    int oC, oH, oW;
};

static inline int ceil_div_float(int a, int b) {
    // the reference rounds through float: ceil((float)a / (float)b)
    float q = static_cast<float>(a) / static_cast<float>(b);
    int r = static_cast<int>(q);
    return (static_cast<float>(r) < q) ? r + 1 : r;
}

static inline int corr_geom_init(CorrGeom &g, int B, int C, int H, int W, int pad, int ksize,
                                 int maxd, int s1, int s2) {
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || pad < 0 || ksize <= 0 || maxd < 0 || s1 <= 0 ||
        s2 <= 0)
        return CERB_EINVAL;
    g.B = B; g.C = C; g.H = H; g.W = W;
    g.pad = pad; g.ksize = ksize; g.maxd = maxd; g.s1 = s1; g.s2 = s2;
    g.krad = (ksize - 1) / 2;
    const int border = g.krad + maxd;
    g.drad = maxd / s2;
    g.dsize = 2 * g.drad + 1;
    g.oC = g.dsize * g.dsize;
    g.oH = ceil_div_float(H + 2 * pad - 2 * border, s1);
    g.oW = ceil_div_float(W + 2 * pad - 2 * border, s1);
    if (g.oH <= 0 || g.oW <= 0) return CERB_EINVAL;
    return CERB_OK;
}

// process-wide knobs (api.hip): resolved to an index at compile time, one relaxed atomic
// load per use (names only matter to cerberus_set_option / cerberus_get_option)
enum OptId {
    OPT_CORR_FORCE_GENERIC = 0,
    OPT_CORR_FWD_VARIANT,
    OPT_CORR_BWD_VARIANT,
    OPT_CORR_BWD_CSLICE,
    OPT_CORR_NO_MFMA,
    OPT_WARP_PAIR_TAPS,
    OPT_WARP_TILE_RANGES,
    OPT_WARP_TILE_H,
    OPT_WARP_FORCE_SCATTER,
    OPT_WARP_STAGED,
    OPT_WARP_STAGGER,
    OPT_WARP_FEWC,
    OPT_WARP_PAIR16,
#ifdef CERB_ABLATE
    OPT_DEBUG_ABLATE,   // timing-ablation mask: exists in -DCERB_ABLATE builds only
#endif
    OPT_COUNT
};
int option(OptId id);
// ablation mask of the correlation kernels: the constant 0 in the product build
static inline int debug_mask() {
#ifdef CERB_ABLATE
    return option(OPT_DEBUG_ABLATE);
#else
    return 0;
#endif
}
void note_kernel(int which, const char *name);

// XCD-aware work-item order (speed only, never correctness).  The dispatcher deals
// workgroups round-robin over the 8 XCDs, each with a private 4 MiB L2, so blocks b and b+8
// share an L2 but b and b+1 do not.  Remap so that every XCD walks ONE contiguous range of
// work items: spatial neighbours (which share halo lines) then hit in the same L2 instead
// of going back to the fabric.  Bijective for any block count.
__device__ __forceinline__ int xcd_chunk(int bid, int nblocks) {
    constexpr int kXcd = 8;
    const int x = bid % kXcd, idx = bid / kXcd;
    const int q = nblocks / kXcd, rem = nblocks % kXcd;
    return x * q + min(x, rem) + idx;
}

#if defined(__HIP_DEVICE_COMPILE__)
// Buffer resource over [p, p+bytes) with the pointer pinned to SGPRs (64-bit address
// arithmetic runs on the VALU; a resource left in VGPRs costs a waterfall loop per use).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *p, int bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>((static_cast<uint64_t>(hi) << 32) | lo), 0,
        __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#endif

// LDS above 64 KiB needs an explicit opt-in, once per kernel AND device (function attributes
// are per device): one bit per device ordinal in an atomic mask owned by the call site.  The
// bit is set only AFTER hipFuncSetAttribute has succeeded, so a concurrent caller either sees
// the bit (attribute in place) or sets the attribute again itself (harmless), and a failed
// call is retried by the next launch.
template <typename Kern>
int ensure_lds(Kern kern, size_t bytes, std::atomic<uint64_t> *done) {
    if (bytes <= 64 * 1024) return CERB_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t bit = 1ull << (dev & 63);
    if (done->load(std::memory_order_acquire) & bit) return CERB_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(bytes));
    if (e != hipSuccess) return static_cast<int>(e);
    done->fetch_or(bit, std::memory_order_release);
    return CERB_OK;
}

// ---- launchers implemented in the kernel translation units ------------------
// all return hipError_t (as int) of the launch, or a negative CERB_E* code.
int corr_generic_forward(const void *in1, const void *in2, void *out, const CorrGeom &g,
                         float slope, int64_t out_bstride, int dtype, hipStream_t s);
int corr_generic_backward(const void *in1, const void *in2, const void *gout, void *gin1,
                          void *gin2, const CorrGeom &g, int dtype, hipStream_t s);

// fast path: pad == d == 4, k == 1, s1 == s2 == 1, fp32 (corr_d4.hip).
// Returns CERB_EUNSUPPORTED when the shape/dtype is not covered -> caller falls
// back to the generic kernels (still HIP; there is no CPU path anywhere).
int corr_d4_forward(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope,
                    int64_t out_bstride, int dtype, hipStream_t s);
int corr_d4_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
                     const CorrGeom &g, int dtype, hipStream_t s);

// 16-bit storage on the matrix cores (corr_mfma.hip); preconditions checked by corr_d4.hip
int corr_mfma_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
                       const CorrGeom &g, int dtype, hipStream_t s);

// fp32 backward for image rows that fit one wavefront (corr_strip.hip); CERB_EUNSUPPORTED otherwise
int corr_strip_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
                        const CorrGeom &g, hipStream_t s);

// forward for the coarse levels, W = 16 / 32 / 64, fp32 / fp16 / bf16 storage (corr_coarse.hip); CERB_EUNSUPPORTED otherwise
int corr_coarse_forward(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope,
                        int64_t obs, int dtype, hipStream_t s);

int corr_coarse_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2, const CorrGeom &g,
                         hipStream_t s);

int corr_mfma_forward(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope,
                      int64_t obs, int dtype, hipStream_t s);

// fp32 forward as a persistent, cross-item pipelined grid (corr_fwd_pipe.hip; -DCERB_EXPERIMENTS builds only: measured
// slower than the tile kernels); CERB_EUNSUPPORTED unless C % 8 == 0
int corr_fwd_pipe(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope, int64_t obs,
                  hipStream_t s);

int64_t warp_context_bytes(int B, int H, int W);
int64_t warp_backward_workspace_bytes(int B, int C, int H, int W);
int warp_forward(const void *image, const void *flow, void *out, void *ctx, int64_t ctx_size,
                 int B, int C, int H, int W, int pad_mode, int interp, int dtype, int flow_dtype,
                 hipStream_t s);
int warp_backward(const void *image, const void *flow, const void *gout, void *gimage,
                  void *gflow, const void *ctx, int64_t ctx_size, void *workspace,
                  int64_t workspace_bytes, int B, int C, int H, int W, int pad_mode, int interp,
                  int dtype, int flow_dtype, hipStream_t s);

// warp16.hip: 16-bit storage, two pixels per lane and two channels per LDS dword; CERB_EUNSUPPORTED when it does not apply
int warp16_forward(const void *image, const void *flow, void *out, void *ctx, int B, int C, int H, int W, int pad_mode,
                   int dtype, int flow_dtype, int crange_opt, hipStream_t s);

// warp_corr.hip: f2 -- the warp fused into the correlation forward (d = 4)
int64_t warp_corr_workspace_bytes(int B, int C, int H, int W);
int warp_corr_forward(const void *f1, const void *f2, const void *flow, void *out, void *workspace, int64_t workspace_bytes, int B,
                      int C, int H, int W, int pad_mode, float slope, int64_t out_bstride, int dtype, int flow_dtype, hipStream_t s);

// upsample.hip: flow * factor -> bilinear x factor, align_corners = true (forward) / its adjoint
int flow_upsample(bool forward, const void *src, void *dst, int64_t planes, int H, int W, int factor,
                  int dtype, hipStream_t s);
int area_resize(const void *src, void *dst, int64_t planes, int H, int W, int oH, int oW, int dtype,
                hipStream_t s);
// every scale of an image pyramid in one pass over the source (integer ratios 4..64); CERB_EUNSUPPORTED otherwise
int area_pyramid(const void *src, void *const *dsts, const int *out_h, const int *out_w, int n, int64_t planes, int H,
                 int W, int dtype, hipStream_t s);

// corr_grad_prep.hip: dense gradOutput (LeakyReLU derivative applied from the stored volume's sign) from a
// batch-strided one; `fwd` may be null (copy only)
int corr_grad_prep(const void *gout, int64_t g_stride, const void *fwd, int64_t f_stride, void *dst, int B, int64_t count,
                   float slope, int dtype, hipStream_t s);

static inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CERB_OK : static_cast<int>(e);
}

}  // namespace cerb
