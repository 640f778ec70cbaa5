// upsample.hip -- the flow pyramid's upsampling step, fused.
//
// Reference (/root/reference/nnet_training/nnet_models/pwcnet_sfd.py):
//   :176      flow = F.interpolate(flow * 2, scale_factor=2, mode='bilinear', align_corners=True)
//   :199-201  flows = [F.interpolate(flow * 4, scale_factor=4, mode='bilinear', align_corners=True) ...]
// i.e. an elementwise multiply + ATen upsample_bilinear2d (2 launches forward; backward: ATen's
// atomicAdd scatter + a multiply).  Here: one launch each way; the backward is a deterministic
// GATHER (every input element sums the output elements whose taps touch it, in a fixed order)
// instead of float atomics.  SURVEY.md section 8(f)-3.
//
// Arithmetic follows ATen's upsample_bilinear2d with align_corners = true:
//   r = (in - 1) / (out - 1) (float, 0 when out == 1);  src = r * dst;  i0 = (int)src;
//   i1 = i0 + (i0 < in - 1);  l1 = src - i0;  l0 = 1 - l1;
//   out = l0y * (l0x * v00 + l1x * v01) + l1y * (l0x * v10 + l1x * v11)
// The factor (2 or 4) multiplies the result: scaling by a power of two commutes exactly with
// every rounding above, so factor * interp(flow) == interp(factor * flow) bit for bit.
#include "common.h"

namespace cerb {
namespace {

struct Tap { int i0, i1; float l0, l1; };
// contraction off: ATen rounds the source coordinate r * dst ONCE and derives both the index and
// the weight from that rounded value; a fused r * dst - i0 is more accurate but moves the
// weights by up to half an ulp of the coordinate (4e-6 at x ~ 64: 7e-6 of the output range
// against torch, measured)
#pragma clang fp contract(off)
__device__ __forceinline__ Tap tap_of(int dst, float r, int in) {
    const float src = r * static_cast<float>(dst);
    const int i0 = min(static_cast<int>(src), in - 1);
    const int i1 = i0 + (i0 < in - 1 ? 1 : 0);
    const float l1 = src - static_cast<float>(i0);
    return {i0, i1, 1.0f - l1, l1};
}
__host__ __device__ inline float ratio(int in, int out) {
    return out > 1 ? static_cast<float>(in - 1) / static_cast<float>(out - 1) : 0.0f;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                           int64_t planes, int H, int W, int oH, int oW,
                                                           float factor) {
    const float ry = ratio(H, oH), rx = ratio(W, oW);
    const int64_t total = planes * oH * oW;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += static_cast<int64_t>(gridDim.x) * 256) {
        const int ox = static_cast<int>(idx % oW);
        const int oy = static_cast<int>((idx / oW) % oH);
        const int64_t pl = idx / (static_cast<int64_t>(oW) * oH);
        const Tap ty = tap_of(oy, ry, H), tx = tap_of(ox, rx, W);
        const T *p = in + pl * H * W;
        const float v00 = ld(p + ty.i0 * W + tx.i0), v01 = ld(p + ty.i0 * W + tx.i1);
        const float v10 = ld(p + ty.i1 * W + tx.i0), v11 = ld(p + ty.i1 * W + tx.i1);
        const float v = ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
        st(out + idx, v * factor);
    }
}

// first output index whose source coordinate r * dst can reach input index i - 1 (exclusive
// lower bound handled by the caller's exact re-test), conservative by one
__device__ __forceinline__ int first_dst(int i, float r, int out) {
    if (r <= 0.f) return 0;
    return max(0, static_cast<int>(floorf(static_cast<float>(i - 1) / r)) - 1);
}
__device__ __forceinline__ int last_dst(int i, float r, int out) {
    if (r <= 0.f) return out - 1;
    return min(out - 1, static_cast<int>(ceilf(static_cast<float>(i + 1) / r)) + 1);
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T *__restrict__ gout, T *__restrict__ gin,
                                                           int64_t planes, int H, int W, int oH, int oW,
                                                           float factor) {
    const float ry = ratio(H, oH), rx = ratio(W, oW);
    const int64_t total = planes * H * W;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += static_cast<int64_t>(gridDim.x) * 256) {
        const int x = static_cast<int>(idx % W);
        const int y = static_cast<int>((idx / W) % H);
        const int64_t pl = idx / (static_cast<int64_t>(W) * H);
        const T *g = gout + pl * oH * oW;
        float acc = 0.f;
        // candidate output rows / columns: every dst whose taps (i0, i1) contain this index;
        // the exact taps are recomputed, so the conservative bounds only cost a few iterations
        const int y_lo = first_dst(y, ry, oH), y_hi = last_dst(y, ry, oH);
        const int x_lo = first_dst(x, rx, oW), x_hi = last_dst(x, rx, oW);
        for (int oy = y_lo; oy <= y_hi; ++oy) {
            const Tap ty = tap_of(oy, ry, H);
            const float wy = (ty.i0 == y ? ty.l0 : 0.f) + (ty.i1 == y ? ty.l1 : 0.f);
            if (ty.i0 != y && ty.i1 != y) continue;
            float row = 0.f;
            for (int ox = x_lo; ox <= x_hi; ++ox) {
                const Tap tx = tap_of(ox, rx, W);
                if (tx.i0 != x && tx.i1 != x) continue;
                const float wx = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
                row += wx * ld(g + oy * oW + ox);
            }
            acc += wy * row;
        }
        st(gin + idx, acc * factor);
    }
}

template <typename T>
int launch(bool fwd, const void *a, void *b, int64_t planes, int H, int W, int oH, int oW, float factor,
           hipStream_t s) {
    const int64_t total = planes * (fwd ? static_cast<int64_t>(oH) * oW : static_cast<int64_t>(H) * W);
    const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((total + 255) / 256, 8192));
    if (fwd)
        hipLaunchKernelGGL(upsample_fwd_kernel<T>, dim3(blocks), dim3(256), 0, s, static_cast<const T *>(a),
                           static_cast<T *>(b), planes, H, W, oH, oW, factor);
    else
        hipLaunchKernelGGL(upsample_bwd_kernel<T>, dim3(blocks), dim3(256), 0, s, static_cast<const T *>(a),
                           static_cast<T *>(b), planes, H, W, oH, oW, factor);
    return launch_status();
}

// ---- 'area' resize of the photometric loss's target images -------------------------------
// F.interpolate(image, (h, w), mode='area') (UnFlowLoss.py:279-280) = adaptive average pooling:
// output (oy, ox) is the mean of rows [floor(oy H / h), ceil((oy + 1) H / h)) x the same in x.
// ATen's arithmetic: an fp32 sum in row-major order, then / kh / kw -- reproduced exactly
// (bit-identical to torch's CPU kernel on fp32).  One thread per output element; neighbouring
// threads read neighbouring windows, so rows are read coalesced.
template <typename T>
__global__ __launch_bounds__(256) void area_resize_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                          int64_t planes, int H, int W, int oH, int oW) {
    const int64_t total = planes * oH * oW;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += static_cast<int64_t>(gridDim.x) * 256) {
        const int ox = static_cast<int>(idx % oW);
        const int oy = static_cast<int>((idx / oW) % oH);
        const int64_t pl = idx / (static_cast<int64_t>(oW) * oH);
        const int y0 = static_cast<int>(static_cast<int64_t>(oy) * H / oH);
        const int y1 = static_cast<int>((static_cast<int64_t>(oy + 1) * H + oH - 1) / oH);
        const int x0 = static_cast<int>(static_cast<int64_t>(ox) * W / oW);
        const int x1 = static_cast<int>((static_cast<int64_t>(ox + 1) * W + oW - 1) / oW);
        const T *p = in + pl * H * W;
        float sum = 0.f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) sum += ld(p + static_cast<int64_t>(y) * W + x);
        st(out + idx, sum / static_cast<float>(y1 - y0) / static_cast<float>(x1 - x0));
    }
}

template <typename T>
int launch_area(const void *a, void *b, int64_t planes, int H, int W, int oH, int oW, hipStream_t s) {
    const int64_t total = planes * oH * oW;
    const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((total + 255) / 256, 8192));
    hipLaunchKernelGGL(area_resize_kernel<T>, dim3(blocks), dim3(256), 0, s, static_cast<const T *>(a),
                       static_cast<T *>(b), planes, H, W, oH, oW);
    return launch_status();
}

}  // namespace

int area_resize(const void *src, void *dst, int64_t planes, int H, int W, int oH, int oW, int dtype,
                hipStream_t s) {
    switch (dtype) {
        case CERB_F32: return launch_area<float>(src, dst, planes, H, W, oH, oW, s);
        case CERB_F16: return launch_area<__half>(src, dst, planes, H, W, oH, oW, s);
        case CERB_BF16: return launch_area<hip_bfloat16>(src, dst, planes, H, W, oH, oW, s);
        default: return CERB_EDTYPE;
    }
}

int flow_upsample(bool forward, const void *src, void *dst, int64_t planes, int H, int W, int factor,
                  int dtype, hipStream_t s) {
    const int oH = H * factor, oW = W * factor;
    const float f = static_cast<float>(factor);
    switch (dtype) {
        case CERB_F32: return launch<float>(forward, src, dst, planes, H, W, oH, oW, f, s);
        case CERB_F16: return launch<__half>(forward, src, dst, planes, H, W, oH, oW, f, s);
        case CERB_BF16: return launch<hip_bfloat16>(forward, src, dst, planes, H, W, oH, oW, f, s);
        default: return CERB_EDTYPE;
    }
}

}  // namespace cerb
