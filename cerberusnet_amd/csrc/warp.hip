// warp.hip -- fused flow_warp forward / backward.
//
// Replaces the whole body of the reference's flow_warp
// (/root/reference/nnet_training/loss_functions/UnFlowLoss.py:83-94):
//   mesh_grid (:11-20, built on the CPU and copied to the device every call)
//   + flow -> norm_grid (:22-32) -> F.grid_sample(align_corners=False) (:92-93)
// and the autograd chain behind it, in ONE kernel per direction: no (B,H,W,2)
// grid tensor, no host mesh, no elementwise kernels.
//
// The coordinate arithmetic reproduces the reference's fp32 rounding sequence
// exactly (x + f, *2, /(W-1), -1, then ATen's unnormalise), with contraction
// disabled, because at x ~ 256 one fp32 ulp of coordinate (3e-5 px) is already
// above the 1e-5 parity budget (hard part 2).  Quirk Q2 is
// reproduced: the grid is normalised by (W-1) but sampled with
// align_corners=False, so zero flow is not the identity.
//
// Work decomposition: a workgroup is 64 consecutive pixels (the lanes of a
// wave: coalesced flow loads, output stores and near-coalesced taps) times
// kCg channel groups (one wave each).  The flow gradient is a sum over
// channels: each wave reduces its own channels in registers, the kCg partials
// meet in LDS, wave 0 stores -- deterministic, no atomics.  The image gradient
// is a data-dependent scatter: fp32/fp64 hardware atomics, as ATen does.
#include "common.h"

namespace cerb {
namespace {

constexpr int kCg = 4;           // channel groups (waves) per workgroup
constexpr int kPix = 64;         // pixels per workgroup = one wavefront

template <typename A> struct Coord {
    A pos;   // source index after unnormalise + padding
    A mult;  // d(pos)/d(flow component), 0 where clamped
};

#pragma clang fp contract(off)
template <typename A>
__device__ __forceinline__ Coord<A> source_coord(int pix, A flow, int size, int pad_mode) {
    // norm_grid: 2.0 * v / (size - 1) - 1.0   (UnFlowLoss.py:30-31)
    const A v = static_cast<A>(pix) + flow;
    const A t = A(2.0) * v;
    const A u = t / static_cast<A>(size - 1);
    const A g = u - A(1.0);
    // ATen grid_sampler_unnormalize, align_corners = false
    A p = ((g + A(1.0)) * static_cast<A>(size) - A(1.0)) / A(2.0);
    // d(p)/d(flow) = (size/2) * (1/(size-1)) * 2, in autograd's order
    A m = static_cast<A>(size) / A(2.0);
    if (pad_mode == CERB_PAD_BORDER) {
        // clip_coordinates_set_grad: gradient is 0 AT and beyond both limits
        const A hi = static_cast<A>(size - 1);
        if (p <= A(0)) { p = A(0); m = A(0); }
        else if (p >= hi) { p = hi; m = A(0); }
    } else if (pad_mode == CERB_PAD_REFLECTION) {
        // reflect_coordinates(p, -1, 2*size-1) then clip (forward only)
        const A mn = A(-0.5), span = static_cast<A>(size);
        A a = fabs(p - mn);
        A extra = fmod(a, span);
        const long long flips = static_cast<long long>(floor(a / span));
        p = (flips % 2 == 0) ? extra + mn : span - extra + mn;
        const A hi = static_cast<A>(size - 1);
        p = p < A(0) ? A(0) : (p > hi ? hi : p);
    }
    return {p, m};
}

template <typename T>
__global__ __launch_bounds__(kPix * kCg) void warp_fwd_kernel(
    const T *__restrict__ image, const T *__restrict__ flow, T *__restrict__ out, int C, int H,
    int W, int pad_mode, int interp) {
    using A = typename Acc<T>::type;
    const int64_t plane = static_cast<int64_t>(H) * W;
    const int lane = threadIdx.x & (kPix - 1);
    const int cg = threadIdx.x / kPix;
    const int64_t p = static_cast<int64_t>(blockIdx.x) * kPix + lane;
    const int b = blockIdx.y;
    if (p >= plane) return;
    const int y = static_cast<int>(p / W), x = static_cast<int>(p % W);
    const T *fl = flow + static_cast<int64_t>(b) * 2 * plane + p;
    const Coord<A> cx = source_coord<A>(x, ld(fl), W, pad_mode);
    const Coord<A> cy = source_coord<A>(y, ld(fl + plane), H, pad_mode);
    const T *img = image + static_cast<int64_t>(b) * C * plane;
    T *dst = out + static_cast<int64_t>(b) * C * plane + p;
    if (interp == CERB_INTERP_NEAREST) {
        const A xn = nearbyint(cx.pos), yn = nearbyint(cy.pos);
        const bool ok = xn >= A(0) && xn < static_cast<A>(W) && yn >= A(0) && yn < static_cast<A>(H);
        const int64_t off = ok ? static_cast<int64_t>(yn) * W + static_cast<int64_t>(xn) : 0;
        for (int c = cg; c < C; c += kCg)
            st(dst + c * plane, ok ? ld(img + c * plane + off) : A(0));
        return;
    }
    const A x0f = floor(cx.pos), y0f = floor(cy.pos);
    const A x1f = x0f + A(1), y1f = y0f + A(1);
    const A wnw = (x1f - cx.pos) * (y1f - cy.pos);
    const A wne = (cx.pos - x0f) * (y1f - cy.pos);
    const A wsw = (x1f - cx.pos) * (cy.pos - y0f);
    const A wse = (cx.pos - x0f) * (cy.pos - y0f);
    const int x0 = static_cast<int>(x0f), y0 = static_cast<int>(y0f);
    const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
    const int64_t o00 = static_cast<int64_t>(y0) * W + x0;
    for (int c = cg; c < C; c += kCg) {
        const T *q = img + c * plane + o00;
        A acc = 0;
        if (oky0 && okx0) acc += ld(q) * wnw;
        if (oky0 && okx1) acc += ld(q + 1) * wne;
        if (oky1 && okx0) acc += ld(q + W) * wsw;
        if (oky1 && okx1) acc += ld(q + W + 1) * wse;
        st(dst + c * plane, acc);
    }
}

// ---- atomics ----------------------------------------------------------------
__device__ __forceinline__ void atomic_accumulate(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_accumulate(double *p, double v) { unsafeAtomicAdd(p, v); }
// 16-bit storage: CAS on the containing dword (AMP path only; not on the fp32 headline path)
template <typename H16> __device__ __forceinline__ void atomic_accumulate_16(H16 *p, float v) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    unsigned int *word = reinterpret_cast<unsigned int *>(addr & ~uintptr_t(3));
    const bool upper = addr & 2;
    unsigned int seen = *word, assumed;
    do {
        assumed = seen;
        unsigned short bits = upper ? static_cast<unsigned short>(assumed >> 16)
                                    : static_cast<unsigned short>(assumed & 0xFFFFu);
        H16 cur;
        __builtin_memcpy(&cur, &bits, 2);
        H16 next;
        st(&next, ld(&cur) + v);
        unsigned short nb;
        __builtin_memcpy(&nb, &next, 2);
        const unsigned int repl = upper ? ((assumed & 0x0000FFFFu) | (static_cast<unsigned int>(nb) << 16))
                                        : ((assumed & 0xFFFF0000u) | nb);
        seen = atomicCAS(word, assumed, repl);
    } while (seen != assumed);
}
__device__ __forceinline__ void atomic_accumulate(__half *p, float v) { atomic_accumulate_16(p, v); }
__device__ __forceinline__ void atomic_accumulate(hip_bfloat16 *p, float v) { atomic_accumulate_16(p, v); }

template <typename T>
__global__ __launch_bounds__(kPix * kCg) void warp_bwd_kernel(
    const T *__restrict__ image, const T *__restrict__ flow, const T *__restrict__ gout,
    T *__restrict__ gimage, T *__restrict__ gflow, int C, int H, int W, int pad_mode) {
    using A = typename Acc<T>::type;
    __shared__ A part[kCg][2][kPix];
    const int64_t plane = static_cast<int64_t>(H) * W;
    const int lane = threadIdx.x & (kPix - 1);
    const int cg = threadIdx.x / kPix;
    const int64_t p = static_cast<int64_t>(blockIdx.x) * kPix + lane;
    const int b = blockIdx.y;
    const bool live = p < plane;
    A gix = 0, giy = 0;
    Coord<A> cx{0, 0}, cy{0, 0};
    if (live) {
        const int y = static_cast<int>(p / W), x = static_cast<int>(p % W);
        const T *fl = flow + static_cast<int64_t>(b) * 2 * plane + p;
        cx = source_coord<A>(x, ld(fl), W, pad_mode);
        cy = source_coord<A>(y, ld(fl + plane), H, pad_mode);
        const A x0f = floor(cx.pos), y0f = floor(cy.pos);
        const A x1f = x0f + A(1), y1f = y0f + A(1);
        const A wnw = (x1f - cx.pos) * (y1f - cy.pos);
        const A wne = (cx.pos - x0f) * (y1f - cy.pos);
        const A wsw = (x1f - cx.pos) * (cy.pos - y0f);
        const A wse = (cx.pos - x0f) * (cy.pos - y0f);
        const int x0 = static_cast<int>(x0f), y0 = static_cast<int>(y0f);
        const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
        const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
        const int64_t o00 = static_cast<int64_t>(y0) * W + x0;
        const int64_t base = static_cast<int64_t>(b) * C * plane;
        for (int c = cg; c < C; c += kCg) {
            const A g = ld(gout + base + c * plane + p);
            const int64_t q = base + c * plane + o00;
            if (gimage) {
                if (oky0 && okx0) atomic_accumulate(gimage + q, wnw * g);
                if (oky0 && okx1) atomic_accumulate(gimage + q + 1, wne * g);
                if (oky1 && okx0) atomic_accumulate(gimage + q + W, wsw * g);
                if (oky1 && okx1) atomic_accumulate(gimage + q + W + 1, wse * g);
            }
            if (gflow) {
                const A vnw = (oky0 && okx0) ? ld(image + q) : A(0);
                const A vne = (oky0 && okx1) ? ld(image + q + 1) : A(0);
                const A vsw = (oky1 && okx0) ? ld(image + q + W) : A(0);
                const A vse = (oky1 && okx1) ? ld(image + q + W + 1) : A(0);
                gix += (-vnw * (y1f - cy.pos) + vne * (y1f - cy.pos) - vsw * (cy.pos - y0f) +
                        vse * (cy.pos - y0f)) * g;
                giy += (-vnw * (x1f - cx.pos) - vne * (cx.pos - x0f) + vsw * (x1f - cx.pos) +
                        vse * (cx.pos - x0f)) * g;
            }
        }
    }
    if (!gflow) return;
    part[cg][0][lane] = gix;
    part[cg][1][lane] = giy;
    __syncthreads();
    if (cg == 0 && live) {
        A sx = 0, sy = 0;
#pragma unroll
        for (int k = 0; k < kCg; ++k) { sx += part[k][0][lane]; sy += part[k][1][lane]; }
        // autograd order: grad_grid = mult * sum ; through norm_grid: / (size-1) then * 2.0
        T *gf = gflow + static_cast<int64_t>(b) * 2 * plane + p;
        st(gf, cx.mult * sx / static_cast<A>(W - 1) * A(2.0));
        st(gf + plane, cy.mult * sy / static_cast<A>(H - 1) * A(2.0));
    }
}

size_t dtype_size(int dtype) {
    switch (dtype) {
        case CERB_F32: return 4;
        case CERB_F16: case CERB_BF16: return 2;
        case CERB_F64: return 8;
    }
    return 0;
}

}  // namespace

#define CERB_DISPATCH(dtype, ...)                                   \
    switch (dtype) {                                                \
        case CERB_F32:  { using T = float;        __VA_ARGS__; break; } \
        case CERB_F16:  { using T = __half;       __VA_ARGS__; break; } \
        case CERB_BF16: { using T = hip_bfloat16; __VA_ARGS__; break; } \
        case CERB_F64:  { using T = double;       __VA_ARGS__; break; } \
        default: return CERB_EDTYPE;                                \
    }

int warp_forward(const void *image, const void *flow, void *out, int B, int C, int H, int W,
                 int pad_mode, int interp, int dtype, hipStream_t s) {
    const int64_t plane = static_cast<int64_t>(H) * W;
    if (B == 0) return CERB_OK;
    const dim3 grid(static_cast<unsigned>((plane + kPix - 1) / kPix), B);
    CERB_DISPATCH(dtype, hipLaunchKernelGGL(warp_fwd_kernel<T>, grid, dim3(kPix * kCg), 0, s,
                                            static_cast<const T *>(image),
                                            static_cast<const T *>(flow), static_cast<T *>(out), C,
                                            H, W, pad_mode, interp));
    return launch_status();
}

int warp_backward(const void *image, const void *flow, const void *gout, void *gimage,
                  void *gflow, int B, int C, int H, int W, int pad_mode, int interp, int dtype,
                  hipStream_t s) {
    const int64_t plane = static_cast<int64_t>(H) * W;
    if (B == 0) return CERB_OK;
    const size_t esz = dtype_size(dtype);
    if (!esz) return CERB_EDTYPE;
    if (interp == CERB_INTERP_NEAREST || pad_mode == CERB_PAD_REFLECTION) {
        // nearest: grad_flow is identically zero and grad_image is a pure scatter;
        // no reference caller differentiates through either.
        return CERB_EUNSUPPORTED;
    }
    if (gimage) {
        hipError_t e = hipMemsetAsync(gimage, 0, static_cast<size_t>(B) * C * plane * esz, s);
        if (e != hipSuccess) return static_cast<int>(e);
    }
    const dim3 grid(static_cast<unsigned>((plane + kPix - 1) / kPix), B);
    CERB_DISPATCH(dtype, hipLaunchKernelGGL(warp_bwd_kernel<T>, grid, dim3(kPix * kCg), 0, s,
                                            static_cast<const T *>(image),
                                            static_cast<const T *>(flow),
                                            static_cast<const T *>(gout), static_cast<T *>(gimage),
                                            static_cast<T *>(gflow), C, H, W, pad_mode));
    return launch_status();
}

}  // namespace cerb
