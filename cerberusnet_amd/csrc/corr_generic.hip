// corr_generic.hip -- correlation forward/backward for ARBITRARY
// (pad, kernel, max_displacement, stride1, stride2) and every dtype.
//
// This is the coverage path (the reference op accepts any hyper-parameters,
// correlation_cuda.cpp:3-43); the tuned path for the configuration every model
// uses (pad=d=4, k=1, s=1) lives in corr_d4.hip.  Unlike the reference there is
// no NHWC transposed copy (correlation_cuda_kernel.cu:13-27, 266-293): the
// kernels read NCHW directly, with x on the lanes so every load/store is
// W-contiguous, and express the zero padding by predication.
#include "common.h"

namespace cerb {
namespace {

constexpr int kThreads = 256;

template <typename T>
__global__ __launch_bounds__(kThreads) void corr_fwd_generic_kernel(
    const T *__restrict__ in1, const T *__restrict__ in2, T *__restrict__ out, CorrGeom g,
    float slope, int64_t out_bstride) {
    using A = typename Acc<T>::type;
    const int64_t plane_o = static_cast<int64_t>(g.oH) * g.oW;
    const int64_t total = static_cast<int64_t>(g.B) * g.oC * plane_o;
    const int64_t plane_i = static_cast<int64_t>(g.H) * g.W;
    const A nelems = static_cast<A>(g.ksize * g.ksize * g.C);
    for (int64_t idx = blockIdx.x * static_cast<int64_t>(kThreads) + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * kThreads) {
        const int ox = static_cast<int>(idx % g.oW);
        const int oy = static_cast<int>((idx / g.oW) % g.oH);
        const int tc = static_cast<int>((idx / plane_o) % g.oC);
        const int b = static_cast<int>(idx / (plane_o * g.oC));
        const int ti = tc % g.dsize - g.drad;  // horizontal displacement (fast index, Q1)
        const int tj = tc / g.dsize - g.drad;  // vertical displacement (slow index)
        // centre positions in the UNPADDED frame (.cu:36-37 minus pad)
        const int y1 = oy * g.s1 + g.maxd - g.pad;
        const int x1 = ox * g.s1 + g.maxd - g.pad;
        const int y2 = y1 + tj * g.s2;
        const int x2 = x1 + ti * g.s2;
        const T *p1 = in1 + static_cast<int64_t>(b) * g.C * plane_i;
        const T *p2 = in2 + static_cast<int64_t>(b) * g.C * plane_i;
        A acc = 0;
        for (int j = -g.krad; j <= g.krad; ++j) {
            const int ya = y1 + j, yb = y2 + j;
            if (ya < 0 || ya >= g.H || yb < 0 || yb >= g.H) continue;  // zero padding
            for (int i = -g.krad; i <= g.krad; ++i) {
                const int xa = x1 + i, xb = x2 + i;
                if (xa < 0 || xa >= g.W || xb < 0 || xb >= g.W) continue;
                const T *q1 = p1 + static_cast<int64_t>(ya) * g.W + xa;
                const T *q2 = p2 + static_cast<int64_t>(yb) * g.W + xb;
                for (int ch = 0; ch < g.C; ++ch) acc += ld(q1 + ch * plane_i) * ld(q2 + ch * plane_i);
            }
        }
        A v = acc / nelems;
        v = v > A(0) ? v : v * static_cast<A>(slope);
        const int64_t bs = out_bstride ? out_bstride : g.oC * plane_o;
        st(out + b * bs + tc * plane_o + static_cast<int64_t>(oy) * g.oW + ox, v);
    }
}

// which = 0 -> gradInput1 (.cu:97-172), 1 -> gradInput2 (.cu:174-242); stride1 == 1.
template <typename T>
__global__ __launch_bounds__(kThreads) void corr_bwd_generic_kernel(
    const T *__restrict__ in1, const T *__restrict__ in2, const T *__restrict__ gout,
    T *__restrict__ gin1, T *__restrict__ gin2, CorrGeom g) {
    using A = typename Acc<T>::type;
    const int64_t plane_o = static_cast<int64_t>(g.oH) * g.oW;
    const int64_t plane_i = static_cast<int64_t>(g.H) * g.W;
    const int64_t per_side = static_cast<int64_t>(g.B) * g.C * plane_i;
    const A nelems = static_cast<A>(g.ksize * g.ksize * g.C);
    for (int64_t idx = blockIdx.x * static_cast<int64_t>(kThreads) + threadIdx.x;
         idx < 2 * per_side; idx += static_cast<int64_t>(gridDim.x) * kThreads) {
        const int which = idx >= per_side;
        const int64_t e = which ? idx - per_side : idx;
        const int bx = static_cast<int>(e % g.W);
        const int by = static_cast<int>((e / g.W) % g.H);
        const int c = static_cast<int>((e / plane_i) % g.C);
        const int b = static_cast<int>(e / (plane_i * g.C));
        const int y = by + g.pad, x = bx + g.pad;  // padded frame, .cu:106-107 with s1 = 1
        const T *go_b = gout + static_cast<int64_t>(b) * g.oC * plane_o;
        const T *src = (which ? in1 : in2) + (static_cast<int64_t>(b) * g.C + c) * plane_i;
        A acc = 0;
        if (!which) {
            int xmin = x - g.krad - g.maxd, xmax = x + g.krad - g.maxd;
            int ymin = y - g.krad - g.maxd, ymax = y + g.krad - g.maxd;
            const bool empty = xmax < 0 || ymax < 0 || xmin >= g.oW || ymin >= g.oH;
            if (!empty) {
                xmin = max(0, xmin); xmax = min(g.oW - 1, xmax);
                ymin = max(0, ymin); ymax = min(g.oH - 1, ymax);
                for (int tc = 0; tc < g.oC; ++tc) {
                    const int i2 = (tc % g.dsize - g.drad) * g.s2;
                    const int j2 = (tc / g.dsize - g.drad) * g.s2;
                    const int yy = y + j2 - g.pad, xx = x + i2 - g.pad;
                    if (yy < 0 || yy >= g.H || xx < 0 || xx >= g.W) continue;  // padded zero
                    const A v2 = ld(src + static_cast<int64_t>(yy) * g.W + xx);
                    const T *go = go_b + tc * plane_o;
                    for (int j = ymin; j <= ymax; ++j)
                        for (int i = xmin; i <= xmax; ++i)
                            acc += ld(go + static_cast<int64_t>(j) * g.oW + i) * v2;
                }
            }
            st(gin1 + e, acc / nelems);
        } else {
            for (int tc = 0; tc < g.oC; ++tc) {
                const int i2 = (tc % g.dsize - g.drad) * g.s2;
                const int j2 = (tc / g.dsize - g.drad) * g.s2;
                int xmin = x - g.krad - g.maxd - i2, xmax = x + g.krad - g.maxd - i2;
                int ymin = y - g.krad - g.maxd - j2, ymax = y + g.krad - g.maxd - j2;
                if (xmax < 0 || ymax < 0 || xmin >= g.oW || ymin >= g.oH) continue;
                xmin = max(0, xmin); xmax = min(g.oW - 1, xmax);
                ymin = max(0, ymin); ymax = min(g.oH - 1, ymax);
                const int yy = y - j2 - g.pad, xx = x - i2 - g.pad;
                if (yy < 0 || yy >= g.H || xx < 0 || xx >= g.W) continue;
                const A v1 = ld(src + static_cast<int64_t>(yy) * g.W + xx);
                const T *go = go_b + tc * plane_o;
                for (int j = ymin; j <= ymax; ++j)
                    for (int i = xmin; i <= xmax; ++i)
                        acc += ld(go + static_cast<int64_t>(j) * g.oW + i) * v1;
            }
            st(gin2 + e, acc / nelems);
        }
    }
}

inline int grid_for(int64_t total) {
    int64_t blocks = (total + kThreads - 1) / kThreads;
    const int64_t cap = 256 * 32;  // 256 CUs x 32 blocks; grid-stride beyond that
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return static_cast<int>(blocks);
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

int corr_generic_forward(const void *in1, const void *in2, void *out, const CorrGeom &g,
                         float slope, int64_t out_bstride, int dtype, hipStream_t s) {
    const int64_t total = static_cast<int64_t>(g.B) * g.oC * g.oH * g.oW;
    if (total == 0) return CERB_OK;
    CERB_DISPATCH(dtype, hipLaunchKernelGGL(corr_fwd_generic_kernel<T>, dim3(grid_for(total)),
                                            dim3(kThreads), 0, s, static_cast<const T *>(in1),
                                            static_cast<const T *>(in2), static_cast<T *>(out), g,
                                            slope, out_bstride));
    return launch_status();
}

int corr_generic_backward(const void *in1, const void *in2, const void *gout, void *gin1,
                          void *gin2, const CorrGeom &g, int dtype, hipStream_t s) {
    const int64_t total = 2 * static_cast<int64_t>(g.B) * g.C * g.H * g.W;
    if (total == 0) return CERB_OK;
    CERB_DISPATCH(dtype, hipLaunchKernelGGL(corr_bwd_generic_kernel<T>, dim3(grid_for(total)),
                                            dim3(kThreads), 0, s, static_cast<const T *>(in1),
                                            static_cast<const T *>(in2),
                                            static_cast<const T *>(gout), static_cast<T *>(gin1),
                                            static_cast<T *>(gin2), g));
    return launch_status();
}

}  // namespace cerb
