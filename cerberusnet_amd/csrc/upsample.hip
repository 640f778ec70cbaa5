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

// One workgroup per output row (grid-stride over planes x rows): the row's vertical tap is wave-uniform, the index
// arithmetic 32-bit (round 2 decomposed a 64-bit flat index per element: 22.6 us for the 16.8 MB of the x 4 upsample).
template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                           int64_t planes, int H, int W, int oH, int oW,
                                                           float factor) {
    const float ry = ratio(H, oH), rx = ratio(W, oW);
    const int64_t rows = planes * oH;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int oy = static_cast<int>(row % oH);
        const int64_t pl = row / oH;
        const Tap ty = tap_of(oy, ry, H);
        const T *p0 = in + pl * H * W + static_cast<int64_t>(ty.i0) * W, *p1 = in + pl * H * W + static_cast<int64_t>(ty.i1) * W;
        T *o = out + row * oW;
        for (int ox = threadIdx.x; ox < oW; ox += 256) {
            const Tap tx = tap_of(ox, rx, W);
            const float v00 = ld(p0 + tx.i0), v01 = ld(p0 + tx.i1);
            const float v10 = ld(p1 + tx.i0), v11 = ld(p1 + tx.i1);
            const float v = ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
            st(o + ox, v * factor);
        }
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

// Round 5: one workgroup per INPUT row (grid-stride over planes x rows).  The candidate output rows and their vertical
// weights are the same for the whole row (wave-uniform, computed once), the taps of a thread's candidate columns are
// computed once (they were recomputed for every candidate row: ~144 tap evaluations per element at factor 4), only the
// matching columns are loaded, index arithmetic is 32-bit.  Same products, same order of additions: identical bits.
// (In the host model's training step this kernel ran 14.8 times at 18.8 us on average -- the largest item of this
// package's kernel time once the loss side was fixed: profiles/r05_kernel_stats_step_model.csv.)
constexpr int kUpMaxCand = 16;   // candidate outputs per axis kept in registers: 2 * factor + 4 <= 16 up to factor 6
constexpr int kUpMaxRows = 10;   // matching output rows staged in LDS (2 * factor + 1 <= 9 at factor 4)
constexpr int kUpMaxW = 1024;    // widest output row the LDS copy takes (40 KB for the ten rows)

template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T *__restrict__ gout, T *__restrict__ gin,
                                                           int64_t planes, int H, int W, int oH, int oW,
                                                           float factor) {
    __shared__ __attribute__((aligned(16))) float s_g[kUpMaxRows * kUpMaxW];
    const float ry = ratio(H, oH), rx = ratio(W, oW);
    const int64_t rows = planes * H;
    for (int64_t rowi = blockIdx.x; rowi < rows; rowi += gridDim.x) {
        const int y = static_cast<int>(rowi % H);
        const int64_t pl = rowi / H;
        const T *g = gout + pl * oH * oW;
        // candidate output rows: every dst whose taps (i0, i1) contain this row; the exact taps are recomputed, so the
        // conservative bounds only cost a few iterations
        const int y_lo = first_dst(y, ry, oH), y_hi = last_dst(y, ry, oH);
        // the matching output rows (wave-uniform), in ascending order: their gradient rows are copied into LDS with
        // coalesced 16-byte loads -- a lane's candidate columns lie `factor` elements from its neighbour's, and read
        // straight from global memory every gather instruction touched factor x the cache lines (24.8 us for the x 4
        // upsample's 16.8 MB) -- and the gathers below read LDS.  Same values, same order: identical bits.
        int nrow = 0;
        __shared__ int s_row[kUpMaxRows];
        __shared__ float s_wy[kUpMaxRows];
        // (factor 2: the direct gather is as fast -- 9.4 vs 11.3 us at 4 x 2 x 128 x 256 -- and stays)
        const bool staged = oW >= 3 * W && oW % 4 == 0 && oW <= kUpMaxW && (reinterpret_cast<uintptr_t>(gout) & (4 * sizeof(T) - 1)) == 0;
        if (staged) {
            for (int oy = y_lo; oy <= y_hi && nrow < kUpMaxRows + 1; ++oy) {
                const Tap ty = tap_of(oy, ry, H);
                if (ty.i0 != y && ty.i1 != y) continue;
                if (nrow < kUpMaxRows && threadIdx.x == 0) {
                    s_row[nrow] = oy;
                    s_wy[nrow] = (ty.i0 == y ? ty.l0 : 0.f) + (ty.i1 == y ? ty.l1 : 0.f);
                }
                ++nrow;
            }
        }
        if (staged && nrow <= kUpMaxRows) {
            __syncthreads();              // s_row / s_wy written; the previous row's LDS reads are over
            const int q4 = oW / 4;
            for (int i = threadIdx.x; i < nrow * q4; i += 256) {
                const int r = i / q4, c4 = i - r * q4;
                const T *src = g + static_cast<int64_t>(s_row[r]) * oW + 4 * c4;
                float4 v;
                if constexpr (sizeof(T) == 4) {
                    v = *reinterpret_cast<const float4 *>(src);
                } else {
                    const uint2 raw = *reinterpret_cast<const uint2 *>(src);
                    T e[4];
                    __builtin_memcpy(e, &raw, 8);
                    v = make_float4(ld(&e[0]), ld(&e[1]), ld(&e[2]), ld(&e[3]));
                }
                *reinterpret_cast<float4 *>(s_g + r * kUpMaxW + 4 * c4) = v;
            }
            __syncthreads();
            for (int x = threadIdx.x; x < W; x += 256) {
                const int x_lo = first_dst(x, rx, oW), x_hi = last_dst(x, rx, oW);
                float acc = 0.f;
                if (x_hi - x_lo < kUpMaxCand) {
                    float wx[kUpMaxCand];
                    unsigned hits = 0;
#pragma unroll
                    for (int j = 0; j < kUpMaxCand; ++j) {
                        const int ox = min(x_lo + j, oW - 1);
                        const Tap tx = tap_of(ox, rx, W);
                        const bool hit = x_lo + j <= x_hi && (tx.i0 == x || tx.i1 == x);
                        wx[j] = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
                        hits |= hit ? (1u << j) : 0u;
                    }
                    for (int r = 0; r < nrow; ++r) {
                        const float *grow = s_g + r * kUpMaxW + x_lo;
                        float row = 0.f;
#pragma unroll
                        for (int j = 0; j < kUpMaxCand; ++j)
                            if ((hits >> j) & 1u) row += wx[j] * grow[j];
                        acc += s_wy[r] * row;
                    }
                } else {
                    for (int r = 0; r < nrow; ++r) {
                        float row = 0.f;
                        for (int ox = x_lo; ox <= x_hi; ++ox) {
                            const Tap tx = tap_of(ox, rx, W);
                            if (tx.i0 != x && tx.i1 != x) continue;
                            const float wx = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
                            row += wx * s_g[r * kUpMaxW + ox];
                        }
                        acc += s_wy[r] * row;
                    }
                }
                st(gin + rowi * W + x, acc * factor);
            }
            __syncthreads();              // the next input row overwrites s_row / s_g
            continue;
        }
        for (int x = threadIdx.x; x < W; x += 256) {
            const int x_lo = first_dst(x, rx, oW), x_hi = last_dst(x, rx, oW);
            float acc = 0.f;
            if (x_hi - x_lo < kUpMaxCand) {
                // column weights once; bit j of `hits`: candidate column x_lo + j has a tap on x
                float wx[kUpMaxCand];
                unsigned hits = 0;
#pragma unroll
                for (int j = 0; j < kUpMaxCand; ++j) {
                    const int ox = min(x_lo + j, oW - 1);
                    const Tap tx = tap_of(ox, rx, W);
                    const bool hit = x_lo + j <= x_hi && (tx.i0 == x || tx.i1 == x);
                    wx[j] = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
                    hits |= hit ? (1u << j) : 0u;
                }
                for (int oy = y_lo; oy <= y_hi; ++oy) {
                    const Tap ty = tap_of(oy, ry, H);
                    if (ty.i0 != y && ty.i1 != y) continue;
                    const float wy = (ty.i0 == y ? ty.l0 : 0.f) + (ty.i1 == y ? ty.l1 : 0.f);
                    const T *grow = g + static_cast<int64_t>(oy) * oW + x_lo;
                    float row = 0.f;
#pragma unroll
                    for (int j = 0; j < kUpMaxCand; ++j)
                        if ((hits >> j) & 1u) row += wx[j] * ld(grow + j);
                    acc += wy * row;
                }
            } else {
                for (int oy = y_lo; oy <= y_hi; ++oy) {
                    const Tap ty = tap_of(oy, ry, H);
                    const float wy = (ty.i0 == y ? ty.l0 : 0.f) + (ty.i1 == y ? ty.l1 : 0.f);
                    if (ty.i0 != y && ty.i1 != y) continue;
                    float row = 0.f;
                    for (int ox = x_lo; ox <= x_hi; ++ox) {
                        const Tap tx = tap_of(ox, rx, W);
                        if (tx.i0 != x && tx.i1 != x) continue;
                        const float wx = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
                        row += wx * ld(g + static_cast<int64_t>(oy) * oW + ox);
                    }
                    acc += wy * row;
                }
            }
            st(gin + rowi * W + x, acc * factor);
        }
    }
}

template <typename T>
int launch(bool fwd, const void *a, void *b, int64_t planes, int H, int W, int oH, int oW, float factor,
           hipStream_t s) {
    // one workgroup per output row (forward) / input row (backward), grid-stride beyond 64 K rows
    const int64_t rows = planes * (fwd ? oH : H);
    const unsigned blocks = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(rows, 65536)));
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

// ---- every scale of the loss pyramid from ONE pass over the source -------------------------------
// unFlowLoss resizes both target images to every flow scale (UnFlowLoss.py:279-280): four launches per
// image that each re-read the 25 MB full-resolution image (one of them the identity): 93 us per image.  Here a
// workgroup owns a band of RMAX source rows x SEG columns of one plane (32 KB), copies it into LDS with ONE round
// of coalesced 16-byte loads, and writes the outputs of EVERY scale that lie in the band from there: one trip to
// HBM per workgroup, whatever the number of scales (the first version read the windows from global memory scale
// after scale -- 8 dependent round trips per workgroup: 15.5 us; this one: see profiles/).  Integer ratios 2, 4, ... 64
// (the model's loss pyramid is 1/2, 1/4, 1/8 of the frame); one thread per output element adds its window in row-major order and divides by kh, then
// kw, exactly as area_resize_kernel (and ATen's CPU kernel) does: bit-identical to the per-scale launches.  The
// scales are walked coarsest first, so that the longest chains of adds (a 16 x 16 window: 256) start first.
constexpr int kPyrMax = 4;
constexpr int kPyrBand = 8192;   // floats of LDS per workgroup: RMAX rows x SEG columns
struct AreaPyr {
    void *dst[kPyrMax];
    int ratio[kPyrMax];          // descending
    int n;
    int vec;                     // 4: the 1/2 scale's rows take 16-byte stores (four outputs per thread there), else 1
};

template <int R>
__device__ __forceinline__ float area_window_sum(const float *__restrict__ q, int pitch) {
    if constexpr (R == 2) {
        const float2 a = *reinterpret_cast<const float2 *>(q), b = *reinterpret_cast<const float2 *>(q + pitch);
        float sum = 0.f;
        sum += a.x; sum += a.y; sum += b.x; sum += b.y;
        return sum;
    }
    float sum = 0.f;
    constexpr int kRowsInFlight = R <= 8 ? R : 4;
#pragma unroll kRowsInFlight
    for (int y = 0; y < R; ++y) {
#pragma unroll
        for (int x4 = 0; x4 < R / 4; ++x4) {
            const float4 v = *reinterpret_cast<const float4 *>(q + y * pitch + 4 * x4);
            sum += v.x; sum += v.y; sum += v.z; sum += v.w;
        }
    }
    return sum;
}

template <typename T>
__global__ __launch_bounds__(256) void area_pyramid_kernel(const T *__restrict__ in, AreaPyr pyr, int H, int W, int rmax,
                                                           int seg, int nseg) {
    __shared__ __attribute__((aligned(16))) float band[kPyrBand];
    int id = blockIdx.x;
    const int sg = id % nseg; id /= nseg;
    const int nband = H / rmax;
    const int bnd = id % nband;
    const int64_t pl = id / nband;
    const int xs = sg * seg, xw = min(seg, W - xs);
    const T *src = in + pl * H * W + static_cast<int64_t>(bnd) * rmax * W + xs;
    // the band -> LDS: every thread's loads (up to 8 x 16 bytes) are ALL issued before the first is written to LDS (a loop that
    // loads, converts and stores cell by cell is a chain of dependent trips to HBM: 4 - 8 of them per workgroup, 17 us)
    const int cpr = xw / 4, cells = rmax * cpr;
    constexpr int kCells = kPyrBand / 4 / 256;
    const float inv_cpr = 1.0f / static_cast<float>(cpr);
    float4 v[kCells];
#pragma unroll
    for (int i = 0; i < kCells; ++i) {
        const int k = min(static_cast<int>(threadIdx.x) + 256 * i, cells - 1);
        const int row = static_cast<int>((static_cast<float>(k) + 0.5f) * inv_cpr), c4 = k - row * cpr;
        const T *q = src + static_cast<int64_t>(row) * W + 4 * c4;
        if constexpr (sizeof(T) == 4) {
            v[i] = *reinterpret_cast<const float4 *>(q);
        } else {
            const uint2 raw = *reinterpret_cast<const uint2 *>(q);
            T e[4];
            __builtin_memcpy(e, &raw, 8);
            v[i] = make_float4(ld(&e[0]), ld(&e[1]), ld(&e[2]), ld(&e[3]));
        }
    }
#pragma unroll
    for (int i = 0; i < kCells; ++i) {
        const int k = static_cast<int>(threadIdx.x) + 256 * i;
        if (k < cells) {
            const int row = static_cast<int>((static_cast<float>(k) + 0.5f) * inv_cpr), c4 = k - row * cpr;
            *reinterpret_cast<float4 *>(band + row * xw + 4 * c4) = v[i];
        }
    }
    __syncthreads();
    // The 1/2 scale (2048 outputs per band): a thread owns FOUR horizontally adjacent outputs -- two 16-byte LDS reads per
    // source row, one 16-byte store (a store of one dword per lane is issue-bound).  The coarser scales: ONE output per
    // thread, so that neighbouring lanes read neighbouring 16-byte cells of the band (four adjacent outputs per thread put
    // the lanes 64 / 128 bytes apart: 4- to 8-way bank conflicts, and the window sums took 9.4 of the launch's 14.8 us).
    // Every output is summed on its own in row-major order either way.
    const bool vec2 = pyr.vec == 4 && pyr.ratio[pyr.n - 1] == 2;       // ratios are sorted descending: 2 can only be last
    // (index arithmetic: scale by scale, row = unit / columns by one float multiply -- exact for the <= 8192 units of a band;
    // a flat index over all scales cost a search and two integer divisions per output: 3.6 us of the launch)
    for (int s = 0; s < pyr.n; ++s) {
        const int r = pyr.ratio[s];
        const int rows = rmax / r, oW = W / r, oH = H / r;
        T *dbase = static_cast<T *>(pyr.dst[s]) + pl * oH * oW + static_cast<int64_t>(bnd) * rows * oW + xs / r;
        if (vec2 && s == pyr.n - 1) {
            const int cols = xw / 8;                       // groups of four outputs
            const float inv_cols = 1.0f / static_cast<float>(cols);
            for (int j = threadIdx.x; j < rows * cols; j += 256) {
                const int oy = static_cast<int>((static_cast<float>(j) + 0.5f) * inv_cols), ox = (j - oy * cols) * 4;
                const float *q = band + oy * 2 * xw + ox * 2;
                const float4 a0 = *reinterpret_cast<const float4 *>(q), a1 = *reinterpret_cast<const float4 *>(q + 4);
                const float4 b0 = *reinterpret_cast<const float4 *>(q + xw), b1 = *reinterpret_cast<const float4 *>(q + xw + 4);
                float o[4];
                { float t = 0.f; t += a0.x; t += a0.y; t += b0.x; t += b0.y; o[0] = t / 2.f / 2.f; }
                { float t = 0.f; t += a0.z; t += a0.w; t += b0.z; t += b0.w; o[1] = t / 2.f / 2.f; }
                { float t = 0.f; t += a1.x; t += a1.y; t += b1.x; t += b1.y; o[2] = t / 2.f / 2.f; }
                { float t = 0.f; t += a1.z; t += a1.w; t += b1.z; t += b1.w; o[3] = t / 2.f / 2.f; }
                T *dst = dbase + static_cast<int64_t>(oy) * oW + ox;
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    T e[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) st(&e[k], o[k]);
                    uint2 raw;
                    __builtin_memcpy(&raw, e, 8);
                    *reinterpret_cast<uint2 *>(dst) = raw;
                }
            }
            continue;
        }
        const int cols = xw / r;
        const float inv_cols = 1.0f / static_cast<float>(cols);
        for (int j = threadIdx.x; j < rows * cols; j += 256) {
            const int oy = static_cast<int>((static_cast<float>(j) + 0.5f) * inv_cols), ox = j - oy * cols;
            const float *q = band + oy * r * xw + ox * r;
            float sum;
            switch (r) {
                case 2: sum = area_window_sum<2>(q, xw); break;
                case 4: sum = area_window_sum<4>(q, xw); break;
                case 8: sum = area_window_sum<8>(q, xw); break;
                case 16: sum = area_window_sum<16>(q, xw); break;
                case 32: sum = area_window_sum<32>(q, xw); break;
                default: sum = area_window_sum<64>(q, xw); break;
            }
            st(dbase + static_cast<int64_t>(oy) * oW + ox, sum / static_cast<float>(r) / static_cast<float>(r));
        }
    }
}

}  // namespace

// CERB_EUNSUPPORTED when a scale is not an integer ratio in {2, 4, 8, 16, 32, 64} of the source (the caller then resizes
// scale by scale with area_resize)
int area_pyramid(const void *src, void *const *dsts, const int *out_h, const int *out_w, int n, int64_t planes, int H,
                 int W, int dtype, hipStream_t s) {
    if (n < 1 || n > kPyrMax || dtype == CERB_F64) return CERB_EUNSUPPORTED;
    AreaPyr pyr;
    pyr.n = n;
    int rmax = 0;
    for (int i = 0; i < n; ++i) {
        const int r = H / out_h[i];
        if (r * out_h[i] != H || r * out_w[i] != W || !(r == 2 || r == 4 || r == 8 || r == 16 || r == 32 || r == 64))
            return CERB_EUNSUPPORTED;
        // insertion by descending ratio
        int at = i;
        while (at > 0 && pyr.ratio[at - 1] < r) { pyr.ratio[at] = pyr.ratio[at - 1]; pyr.dst[at] = pyr.dst[at - 1]; --at; }
        pyr.dst[at] = dsts[i];
        pyr.ratio[at] = r;
        rmax = std::max(rmax, r);
    }
    const size_t esz = dtype == CERB_F32 ? 4 : 2;
    if (H % rmax || W % rmax || W % 4 || (reinterpret_cast<uintptr_t>(src) & (4 * esz - 1))) return CERB_EUNSUPPORTED;
    // a band of RMAX rows x SEG columns fills the workgroup's LDS: 16 x 512 for the loss pyramid (768 workgroups for
    // 12 planes of 512 x 1024); SEG is a multiple of every ratio
    const int seg = std::min(W, kPyrBand / rmax);
    const int nseg = (W + seg - 1) / seg;
    pyr.vec = 4;
    for (int i = 0; i < n; ++i) {
        const int r = pyr.ratio[i];
        // the 1/2 scale in groups of four outputs: every segment a whole number of groups, rows 16-byte aligned in memory
        if (r == 2 && ((W / r) % 4 || (seg / r) % 4 || ((W % seg) / r) % 4 || (reinterpret_cast<uintptr_t>(pyr.dst[i]) & (4 * esz - 1)))) pyr.vec = 1;
    }
    const int64_t blocks = planes * (H / rmax) * nseg;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    const dim3 grid(static_cast<unsigned>(blocks));
    switch (dtype) {
        case CERB_F32:
            hipLaunchKernelGGL(area_pyramid_kernel<float>, grid, dim3(256), 0, s, static_cast<const float *>(src), pyr, H, W,
                               rmax, seg, nseg);
            break;
        case CERB_F16:
            hipLaunchKernelGGL(area_pyramid_kernel<__half>, grid, dim3(256), 0, s, static_cast<const __half *>(src), pyr, H,
                               W, rmax, seg, nseg);
            break;
        case CERB_BF16:
            hipLaunchKernelGGL(area_pyramid_kernel<hip_bfloat16>, grid, dim3(256), 0, s,
                               static_cast<const hip_bfloat16 *>(src), pyr, H, W, rmax, seg, nseg);
            break;
        default: return CERB_EDTYPE;
    }
    return launch_status();
}

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
