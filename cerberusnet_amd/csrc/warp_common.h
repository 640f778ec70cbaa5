// warp_common.h -- device helpers shared by the warp translation units (warp.hip: fp32 / general kernels,
// warp16.hip: the 16-bit kernels with two elements per lane).  Everything lives in an anonymous namespace: each
// translation unit gets its own copy (g_warp_zero included).
#pragma once
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace cerb {
namespace {


constexpr int kPix = 64;         // pixels per workgroup = one wavefront

// A "strip" is the 64 pixels one wavefront owns: kStripH rows x kStripW columns.  Two rows
// instead of one: the lower tap row of the strip's first pixel row is the upper tap row of
// its second, so a wave touches 3 source rows for 2 rows of pixels instead of 2 for 1
// (round 1: 1 x 64 strips, fabric reads 1.9x the algorithmic bytes), and strips are walked
// in an XCD-aware order so that vertical neighbours meet in one L2.
constexpr int kStripH = 2, kStripW = kPix / kStripH;
struct Strips {
    int nx, ny;                                    // strips per row / per column of one image
    __host__ __device__ Strips(int H, int W)
        : nx((W + kStripW - 1) / kStripW), ny((H + kStripH - 1) / kStripH) {}
    __host__ __device__ int per_image() const { return nx * ny; }
    // pixel of (strip j of an image, lane); false when the lane is outside the image
    __device__ __forceinline__ bool pixel(int j, int lane, int H, int W, int &x, int &y) const {
        y = (j / nx) * kStripH + lane / kStripW;
        x = (j % nx) * kStripW + lane % kStripW;
        return x < W && y < H;
    }
};

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
    // ATen grid_sampler_unnormalize, align_corners = false: ((g + 1) * size - 1) / 2.  Both
    // builds of the reference runtime FUSE the multiply-subtract (nvcc fmad on the GPU, gcc
    // -ffp-contract on the vectorised CPU kernel: verified against torch CPU, 1-ulp
    // coordinate differences otherwise), so this one product is an explicit fma while
    // everything around it stays uncontracted.
    A p = fma(g + A(1.0), static_cast<A>(size), A(-1.0)) / A(2.0);
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

// Branch-free tap fetch: a tap outside the image reads a block of zeros instead of being
// skipped.  (A branch around a load -- or a select on its result -- makes hipcc wait for
// that load before issuing the next one; measured on the correlation backward gather:
// 162 serialised round trips.  Exact zeros, so NaN/Inf in neighbouring pixels cannot leak.)
__device__ __attribute__((aligned(16))) float g_warp_zero[4] = {0.f, 0.f, 0.f, 0.f};
template <typename T>
__device__ __forceinline__ const T *tap_ptr(const T *real, bool ok) {
    return ok ? real : reinterpret_cast<const T *>(g_warp_zero);
}
// Variant: the two horizontal taps as ONE load when both are inside -- 8 bytes at 4-byte
// alignment (fp32) or 4 bytes at 2-byte alignment (fp16 / bf16; global memory takes unaligned
// addresses) --, separate guarded loads at the border.  Half the gather instructions, but a
// branch.
struct __attribute__((packed, aligned(4))) f32x2_u { float a, b; };
struct __attribute__((packed, aligned(2))) u16x2_u { unsigned short a, b; };
template <typename T> __device__ __forceinline__ float widen16(unsigned short bits);
template <> __device__ __forceinline__ float widen16<__half>(unsigned short bits) {
    __half h;
    __builtin_memcpy(&h, &bits, 2);
    return __half2float(h);
}
template <> __device__ __forceinline__ float widen16<hip_bfloat16>(unsigned short bits) {
    return __uint_as_float(static_cast<unsigned int>(bits) << 16);
}
template <bool PAIR, typename T, typename A>
__device__ __forceinline__ void load_taps(const T *q, bool ok0, bool ok1, A &v0, A &v1) {
    if constexpr (PAIR && sizeof(T) == 4 && sizeof(A) == 4) {
        if (ok0 && ok1) {
            const f32x2_u t = *reinterpret_cast<const f32x2_u *>(q);
            v0 = t.a; v1 = t.b;
        } else {
            v0 = ok0 ? ld(q) : A(0);
            v1 = ok1 ? ld(q + 1) : A(0);
        }
    } else if constexpr (PAIR && sizeof(T) == 2 && sizeof(A) == 4) {
        if (ok0 && ok1) {
            const u16x2_u t = *reinterpret_cast<const u16x2_u *>(q);
            v0 = widen16<T>(t.a); v1 = widen16<T>(t.b);
        } else {
            v0 = ok0 ? ld(q) : A(0);
            v1 = ok1 ? ld(q + 1) : A(0);
        }
    } else {
        v0 = ld(tap_ptr(q, ok0));
        v1 = ld(tap_ptr(q + 1, ok1));
    }
}

// ---- backward context ----------------------------------------------------------
// What the forward already knows and the backward needs again: every pixel's sample
// position (after unnormalise + padding) and, per 64-pixel strip, the SIGNED range of tap
// displacements (tap - own pixel) of the strip's in-image taps.  Signed ranges instead of a
// magnitude: under a smooth flow a grad_image tile is fed by (the tile shifted by -flow), not
// by (the tile grown by |flow|) -- 1.1x instead of 2.3x the tile at +-6 px.
// Saved by the forward (cerberus_flow_warp_forward_ctx) the backward is one launch with no
// pre-pass; without it the backward first runs warp_context_kernel over the flow.
//   int4  ext[B * strips]       {dx_lo, dx_hi, dy_lo, dy_hi} of every 2 x 32 strip (image-major,
//                               then row-major over the strips of an image); empty: lo > hi
//   float pos[B][2][H][W]       sample positions (x plane, y plane)
constexpr int kExtEmptyLo = 0x3fffffff, kExtEmptyHi = -0x3fffffff;
__host__ __device__ inline int64_t ctx_header_bytes(int B, int H, int W) {
    return static_cast<int64_t>(B) * Strips(H, W).per_image() * 4 * sizeof(int);
}
__host__ __device__ inline int64_t ctx_bytes(int B, int H, int W) {
    return ctx_header_bytes(B, H, W) + static_cast<int64_t>(B) * 2 * H * W * sizeof(float);
}
__host__ __device__ __forceinline__ float *ctx_pos(void *ctx, int B, int H, int W) {
    return reinterpret_cast<float *>(static_cast<char *>(ctx) + ctx_header_bytes(B, H, W));
}
__host__ __device__ __forceinline__ const float *ctx_pos(const void *ctx, int B, int H, int W) {
    return reinterpret_cast<const float *>(static_cast<const char *>(ctx) + ctx_header_bytes(B, H, W));
}

struct TapRange {
    int xlo = kExtEmptyLo, xhi = kExtEmptyHi, ylo = kExtEmptyLo, yhi = kExtEmptyHi;
    // taps x0, x0+1 / y0, y0+1 of the pixel (x, y); only taps inside the image count
    __device__ __forceinline__ void add(int x, int y, int x0, int y0, int W, int H) {
        const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
        const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
        if ((okx0 || okx1) && (oky0 || oky1)) {
            xlo = min(xlo, (okx0 ? x0 : x0 + 1) - x); xhi = max(xhi, (okx1 ? x0 + 1 : x0) - x);
            ylo = min(ylo, (oky0 ? y0 : y0 + 1) - y); yhi = max(yhi, (oky1 ? y0 + 1 : y0) - y);
        }
    }
    // wave-wide union, lane 0 publishes
    __device__ __forceinline__ void publish(void *ctx, int slot, int lane) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            xlo = min(xlo, __shfl_xor(xlo, m, 64)); xhi = max(xhi, __shfl_xor(xhi, m, 64));
            ylo = min(ylo, __shfl_xor(ylo, m, 64)); yhi = max(yhi, __shfl_xor(yhi, m, 64));
        }
        if (lane == 0) static_cast<int4 *>(ctx)[slot] = make_int4(xlo, xhi, ylo, yhi);
    }
};

// float -> int that cannot overflow later index arithmetic (positions can be anything,
// including NaN / Inf from a diverged flow: v_cvt_i32_f32 saturates, NaN -> 0)
__device__ __forceinline__ int tap_index(float f) {
    return min(max(static_cast<int>(f), -(1 << 24)), 1 << 24);
}

// One channel's contribution to grad_flow (the derivative of the bilinear sample by its position, times
// gradOutput): d out / d x = (vne - vnw) * (y1 - iy) + (vse - vsw) * (iy - y0), d out / d y likewise.  Round 4: as
// differences and explicit fused multiply-adds -- 5 VALU per component instead of the 9 of the term-by-term form
// (the warp backward executes ~100 VALU instructions per pixel-channel: profiles/r04_pmc_counters.csv) -- and ONE
// definition for every grad_flow role, so that the roles keep producing identical bits.
template <typename A>
__device__ __forceinline__ void flow_grad_terms(A vnw, A vne, A vsw, A vse, A ax, A fx, A ay, A fy, A g, A &gix, A &giy) {
    gix = fma(fma(vne - vnw, ay, (vse - vsw) * fy), g, gix);
    giy = fma(fma(vsw - vnw, ax, (vse - vne) * fx), g, giy);
}

// ---- forward, LDS-staged window -------------------------------------------------------
// The direct gather above moves every 128-byte line a wave's taps touch from L2 to the CU
// (measured at the 32x128x256 level: 0.93 M line reads = 119 MB for a 16.8 MB image, PMC
// TCC_HIT + TCC_MISS in profiles/r02_pmc_counters.csv) and spends most of its instructions on
// per-tap address arithmetic and selects.  Here a workgroup owns an 8 x 32 pixel tile (4
// strips, one per wave) for a range of channels.  It finds the bounding box of the tile's
// taps (under a smooth flow: the tile shifted by the flow, ~11 x 36), copies that window --
// plus a one-pixel apron of zeros where it leaves the image -- into LDS with coalesced
// 16-byte buffer loads whose channel advance is a scalar offset, and takes the four taps of a
// channel from LDS with two ds_read2.  Same arithmetic in the same order as warp_fwd_kernel:
// results are bit-identical.
// A window that does not fit kStageCap floats x the channel count is walked in smaller
// channel groups; one that does not fit for a single channel (a diverged flow) falls back to
// the direct gather for that workgroup.
[[maybe_unused]] constexpr int kStageCap = 8192;   // floats of LDS window per workgroup
[[maybe_unused]] constexpr int kStageMaxArea = 4096;   // largest window of ONE channel (1024 16-byte cells)
constexpr int kStageRows = 4;     // strips (waves) per workgroup, stacked vertically
[[maybe_unused]] constexpr int kDeadOffset = static_cast<int>(0x80000000u);   // buffer offset that reads 0 / drops a store

// Wave-wide min / max through DPP row shifts and row broadcasts; the result is wave-uniform.  Round 6: the six steps are
// v_min_i32_dpp / v_max_i32_dpp themselves (inline asm: the compiler makes v_mov + v_mov_dpp + v_min of every
// __builtin_amdgcn_update_dpp step, 18 VALU instructions per reduction instead of 6; a forward wave runs eight of them, a
// backward tile workgroup eight + two per channel group).  `s_nop 1`: a DPP operand written by the previous VALU instruction
// needs two wait states, and the hazard recogniser does not look into inline asm.  A lane whose DPP source does not exist
// (bound_ctrl off) keeps its value: the same folds as before, bit for bit.
template <bool MAX> __device__ __forceinline__ int wave_minmax(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MAX)
        asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                     "s_nop 1" : "+v"(v));
    else
        asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                     "s_nop 1" : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
#else
    return v;
#endif
}

// 4 consecutive pixels of storage type T through a buffer resource, widened to fp32
template <typename T>
__device__ __forceinline__ float4 buffer_load_px4(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    if constexpr (sizeof(T) == 4) {
        const u4v r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z),
                           __uint_as_float(r.w));
    } else {
        const u2v r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
        return make_float4(widen16<T>(r.x & 0xFFFFu), widen16<T>(r.x >> 16),
                           widen16<T>(r.y & 0xFFFFu), widen16<T>(r.y >> 16));
    }
}
template <typename T>
__device__ __forceinline__ float buffer_load_px1(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    if constexpr (sizeof(T) == 4)
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    else
        return widen16<T>(__builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, soff, 0));
}
template <typename T>
__device__ __forceinline__ void buffer_store_px(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, float v) {
    if constexpr (sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, voff, soff, 0);
    } else {
        T t;
        st(&t, v);
        unsigned short bits;
        __builtin_memcpy(&bits, &t, 2);
        __builtin_amdgcn_raw_buffer_store_b16(bits, rsrc, voff, soff, 0);
    }
}

// The LDS window of a workgroup: the bounding box of its lanes' taps (a one-pixel apron of
// zeros included where a tap leaves the image), and this thread's share of copying it.
struct StageWindow {
    int wx0, wy0, pitch, rows, area;   // uniform; columns start at a multiple of 4, pitch % 4 == 0
    bool empty;                        // no tap of the workgroup is inside the image
    int ncell;                         // 16-byte cells of one channel's window / 256, rounded up
    int grp, ngrp;                     // ncell == 1: this thread copies channels grp, grp + ngrp, ...
    int voff[4], slot[4];              // per owned cell: byte offset in channel 0 (or dead), float slot (or -1)

    // (x0, y0): the lane's north-west tap; dead: no tap of the lane is inside the image.
    // One barrier; boxes: one int4 per wave.
    __device__ __forceinline__ void reduce(bool dead, int x0, int y0, int4 *boxes, int wave, int lane) {
        reduce_lanes(dead ? kExtEmptyLo : x0, dead ? kExtEmptyHi : x0 + 1, dead ? kExtEmptyLo : y0,
                     dead ? kExtEmptyHi : y0 + 1, boxes, wave, lane);
    }
    // the same from per-lane bounds (a lane that owns several pixels folds them first; empty: lo > hi)
    __device__ __forceinline__ void reduce_lanes(int xl, int xh, int yl, int yh, int4 *boxes, int wave, int lane) {
        xl = wave_minmax<false>(xl); xh = wave_minmax<true>(xh);
        yl = wave_minmax<false>(yl); yh = wave_minmax<true>(yh);
        if (lane == 0) boxes[wave] = make_int4(xl, xh, yl, yh);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int4 e = boxes[k];
            xl = min(xl, e.x); xh = max(xh, e.y); yl = min(yl, e.z); yh = max(yh, e.w);
        }
        xl = __builtin_amdgcn_readfirstlane(xl); xh = __builtin_amdgcn_readfirstlane(xh);
        yl = __builtin_amdgcn_readfirstlane(yl); yh = __builtin_amdgcn_readfirstlane(yh);
        empty = xl > xh;
        wx0 = empty ? 0 : xl & ~3;
        wy0 = empty ? 0 : yl;
        pitch = empty ? 4 : (xh - wx0 + 4) & ~3;
        rows = empty ? 1 : yh - yl + 1;
        const int64_t a = static_cast<int64_t>(pitch) * rows;
        area = a > kStageMaxArea ? kStageMaxArea + 1 : static_cast<int>(a);   // "does not fit"
    }
    __device__ __forceinline__ bool fits() const { return area <= kStageMaxArea; }

    // Cell -> (row, column) once per thread; the channel is added to the byte offset per load.
    // cell / p4 for cell < 1024, p4 <= 1024 as (cell * ceil(2^20 / p4)) >> 20: exact.
    __device__ __forceinline__ void map_cells(int tid, int H, int W, int esz) {
        const int p4 = pitch >> 2, cells = rows * p4;
        const unsigned m_row = static_cast<unsigned>(ceilf(1048576.f / static_cast<float>(p4)));
        auto place = [&](int j, int cell) {
            const int row = static_cast<int>((cell * m_row) >> 20), q = cell - row * p4;
            const int gx = wx0 + 4 * q, gy = wy0 + row;
            const bool in = !empty && cell < cells && gx >= 0 && gx < W && gy >= 0 && gy < H;
            voff[j] = in ? (gy * W + gx) * esz : kDeadOffset;
            slot[j] = cell < cells ? 4 * cell : -1;
        };
        if (cells <= 256) {
            // fewer cells than threads: 256 >> shift thread groups take every ngrp-th channel
            const int shift = cells <= 1 ? 0 : 32 - __builtin_clz(cells - 1);
            ncell = 1;
            grp = tid >> shift;
            ngrp = 256 >> shift;
            place(0, tid & ((1 << shift) - 1));
#pragma unroll
            for (int j = 1; j < 4; ++j) { voff[j] = kDeadOffset; slot[j] = -1; }
        } else {
            ncell = (cells + 255) >> 8;
            grp = 0;
            ngrp = 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) place(j, tid + 256 * j);
        }
    }

    // Copy channels [c, c + n) into win[n][rows][pitch]: eight loads in flight per thread, all
    // issued before the first LDS store.  The caller brackets this with barriers.
    template <typename T>
    __device__ __forceinline__ void stage(float *win, __amdgpu_buffer_rsrc_t rsrc, int c, int n, int plane) const {
        constexpr int esz = sizeof(T);
        if (ncell == 1) {
            for (int h0 = 0; h0 < n; h0 += 8 * ngrp) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int h = h0 + grp + ngrp * u;
                    v[u] = buffer_load_px4<T>(rsrc, h < n ? voff[0] + (c + h) * plane * esz : kDeadOffset, 0);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int h = h0 + grp + ngrp * u;
                    if (slot[0] >= 0 && h < n) *reinterpret_cast<float4 *>(win + h * area + slot[0]) = v[u];
                }
            }
        } else {
            for (int h = 0; h < n; ++h) {
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = buffer_load_px4<T>(rsrc, voff[j], (c + h) * plane * esz);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (slot[j] >= 0) *reinterpret_cast<float4 *>(win + h * area + slot[j]) = v[j];
            }
        }
    }
};

}  // namespace
}  // namespace cerb
