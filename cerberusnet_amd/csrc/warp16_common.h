// warp16_common.h -- what the 16-bit (fp16 / bf16 storage) warp kernels share: the raw 16-bit LDS window filled by
// LDS-DMA, the per-lane copy plan, packed fp32 arithmetic and packed conversions.  Included by warp16.hip (forward) and
// warp.hip (the grad_flow role of the backward's launch).  Anonymous namespace: one copy per translation unit.
#pragma once
#include "warp_common.h"

namespace cerb {
namespace {

typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int kTile16W = 64, kTile16H = 8;


// min / max over each 16-lane DPP row; lane 15 of every row holds its row's result (fused DPP steps: see wave_minmax)
template <bool MAX> __device__ __forceinline__ int row_minmax(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MAX)
        asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1" : "+v"(v));
    else
        asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1" : "+v"(v));
#endif
    return v;
}

template <typename T> __device__ __forceinline__ unsigned short bits16(float v) {
    T t;
    st(&t, v);
    unsigned short b;
    __builtin_memcpy(&b, &t, 2);
    return b;
}
template <typename T> __device__ __forceinline__ float lo16(unsigned q) { return widen16<T>(static_cast<unsigned short>(q & 0xFFFFu)); }
template <typename T> __device__ __forceinline__ float hi16(unsigned q) { return widen16<T>(static_cast<unsigned short>(q >> 16)); }

// Two fp32 values side by side: products and sums of a channel PAIR are one v_pk_mul_f32 / v_pk_add_f32 each (the same
// IEEE operations as the scalar forms, uncontracted: identical bits, half the instructions)
typedef float f2v __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ f2v widen2(unsigned q) { return f2v{lo16<T>(q), hi16<T>(q)}; }
// two fp32 -> one dword of two 16-bit values: v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, round to nearest even -- bit for bit
// __float2half / st<hip_bfloat16> (tools/ubench/cvt_check.hip, all 2^32 inputs)
template <typename T> __device__ __forceinline__ unsigned narrow2(float a, float b) {
    if constexpr (std::is_same<T, __half>::value) {
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, h2v));
    } else {
        typedef __bf16 b2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, b2v));
    }
}

// ---- the LDS window, filled by LDS-DMA -------------------------------------------------------------------------------
// One channel's window is rows x pitch 16-bit pixels, RAW (a dword = two horizontally adjacent pixels), copied from the
// image by `buffer_load_dwordx4 ... lds`: no VGPR destination, no ds_write, nothing for the VALU to do, and asynchronous
// -- the next channel group's copy is in flight while this one's taps are blended.  The hardware writes the 64 lanes of
// such an instruction to 64 consecutive 16-byte LDS slots, so lane l of DMA instruction q owns CELL 64 q + l of the
// window (8 pixels, row-major), for every channel: its source offset is computed once, the channel is the scalar offset.
// A cell outside the image gets an out-of-range offset and arrives as zeros (the apron of the `zeros` padding mode and
// of border taps).  Columns start at a multiple of 8 pixels (16-byte aligned sources: W % 8 == 0).
[[maybe_unused]] constexpr int kDmaBuf = 14336;        // bytes per window buffer; two buffers
[[maybe_unused]] constexpr int kDmaMaxCells = 256;     // cells of ONE channel's window (4 DMA instructions); larger: direct gathers
[[maybe_unused]] constexpr int kDmaMaxCh = 8;          // channels per pass

// ceil(65536 / d): cell / d == (cell * m) >> 16 exactly for cell < 256, d <= 256
struct RowMul {
    unsigned v[257];
    constexpr RowMul() : v{} {
        for (int d = 1; d <= 256; ++d) v[d] = (65536u + d - 1) / d;
    }
};
__device__ const RowMul g_rowmul{};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most n (wave-uniform, 0 .. 24) vector-memory operations still outstanding
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
    switch (n) {
#define W(k) case k: wait_vmcnt<k>(); break;
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16)
        W(17) W(18) W(19) W(20) W(21) W(22) W(23)
#undef W
        default: wait_vmcnt<24>(); break;
    }
}
typedef __attribute__((address_space(3))) void *lds_void_ptr;

struct DmaWindow {
    int wx0, wy0, pitch, rows, cells;   // uniform; pitch in pixels, a multiple of the cell's pixels
    bool empty;                         // no tap of the workgroup is inside the image
    // per-lane bounds of the north-west taps (lo > hi: none) -> the workgroup's window.  One barrier.
    // cpx: pixels per 16-byte cell (8 of 16-bit storage, 4 of fp32)
    __device__ __forceinline__ void reduce(int xl, int xh, int yl, int yh, int4 *boxes, int wave, int lane, int cpx = 8) {
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
        wx0 = empty ? 0 : xl & ~(cpx - 1);
        wy0 = empty ? 0 : yl;
        const int64_t pw = empty ? cpx : (static_cast<int64_t>(xh) - wx0 + cpx) & ~static_cast<int64_t>(cpx - 1);
        const int64_t rw = empty ? 1 : static_cast<int64_t>(yh) - yl + 1;
        const int64_t c = (pw / cpx) * rw;
        const bool ok = c <= kDmaMaxCells;
        pitch = ok ? static_cast<int>(pw) : cpx;
        rows = ok ? static_cast<int>(rw) : 1;
        cells = ok ? static_cast<int>(c) : kDmaMaxCells + 1;   // "does not fit"
    }
    __device__ __forceinline__ bool fits() const { return cells <= kDmaMaxCells; }
};


// The lane's share of copying a window: DMA instruction q of a channel copies cells 64 q .. 64 q + 63, lane l the cell
// 64 q + l; the source offsets are computed once, the channel is the scalar offset of the instruction.
struct DmaPlan {
    int voff[4];        // source byte offset of the lane's cell in channel 0 (out of range: zeros arrive)
    bool mine[4];       // the cell exists
    int ninst;          // copy instructions per channel (<= 4)
    int chan_bytes;     // LDS bytes of one channel's window
    int esz;            // bytes per pixel (2 / 4): a cell holds 16 / esz pixels
    __device__ __forceinline__ void init(const DmaWindow &w, int lane, int H, int W, int esz_ = 2) {
        esz = esz_;
        const int cpx = 16 / esz;
        const int p8 = w.pitch / cpx;
        ninst = (w.cells + 63) >> 6;
        chan_bytes = w.cells * 16;
        const unsigned rowmul = g_rowmul.v[p8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cell = 64 * q + lane;
            const int row = static_cast<int>((static_cast<unsigned>(cell) * rowmul) >> 16), col = cell - row * p8;
            const int gx = w.wx0 + cpx * col, gy = w.wy0 + row;
            mine[q] = cell < w.cells;
            voff[q] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? (gy * W + gx) * esz : kDeadOffset;
        }
    }
    // channels c0 .. c0 + n - 1 of the image behind `rsrc` -> buf[n][cells][16 bytes]; channel i by wave i % 4
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rsrc, char *buf, int c0, int n, int wave, int plane) const {
#if defined(__HIP_DEVICE_COMPILE__)
        for (int i = wave; i < n; i += 4) {
            const int soff = __builtin_amdgcn_readfirstlane((c0 + i) * plane * esz);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < ninst && mine[q])
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(buf + i * chan_bytes + q * 1024), 16,
                                                             voff[q], soff, 0, 0);
            }
        }
#endif
    }
    // copy instructions wave `wave` issues for n channels
    __device__ __forceinline__ int count(int n, int wave) const { return n <= 0 ? 0 : ((n - wave + 3) >> 2) * ninst; }
};

// ---- backward, grad_flow for 16-bit storage ---------------------------------------------------------------------------
// The FLOW role of warp_bwd_tile_kernel (warp.hip: flow_role_staged) with the forward's machinery: an 8 x 64 pixel tile
// per workgroup, two pixels per lane, the raw 16-bit source window copied by LDS-DMA eight channels ahead, gradOutput
// as one dword per lane (two pixels), the derivative terms as packed fp32 operations.  Channel c adds into partial
// c & 3 in ascending order, the partials are summed 0..3: the order of every grad_flow role -- identical bits (test).
// `lds`: the accumulators' LDS of a tile workgroup (two window buffers); false (nothing written): a channel group of
// four does not fit, the caller gathers directly.
template <typename T, typename F>
__device__ __forceinline__ bool flow_role16(
    char *__restrict__ lds, int lds_bytes, int4 *__restrict__ boxes, const T *__restrict__ image,
    const T *__restrict__ gout, const void *__restrict__ ctx, F *__restrict__ gflow, int flow_block,
    int nflow_blocks, int B, int C, int H, int W, int pad_mode) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int plane = H * W;
    const int tid = threadIdx.x;
    const int lane = tid & (kPix - 1), wave = __builtin_amdgcn_readfirstlane(tid / kPix);
    const int ntx = (W + kTile16W - 1) / kTile16W, nty = (H + kTile16H - 1) / kTile16H;
    int id = xcd_chunk(flow_block, nflow_blocks);
    const int tx = __builtin_amdgcn_readfirstlane(id % ntx); id /= ntx;
    const int ty = __builtin_amdgcn_readfirstlane(id % nty);
    const int b = __builtin_amdgcn_readfirstlane(id / nty);
    const int y = ty * kTile16H + wave * 2 + (lane >> 5);
    const int xa = tx * kTile16W + 2 * (lane & 31);       // pixels xa, xa + 1 (W % 8 == 0: both inside or both outside)
    const bool live = y < H && xa < W;
    const int p = y * W + xa, pc = live ? p : 0;
    const float *pos = ctx_pos(ctx, B, H, W) + static_cast<int64_t>(b) * 2 * plane;
    const float2 ixp = *reinterpret_cast<const float2 *>(pos + pc), iyp = *reinterpret_cast<const float2 *>(pos + plane + pc);
    const float ix[2] = {ixp.x, ixp.y}, iy[2] = {iyp.x, iyp.y};
    int x0[2], y0[2];
    bool dead[2];
    float ax[2], fx[2], ay[2], fy[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float x0f = floorf(ix[k]), y0f = floorf(iy[k]);
        ax[k] = (x0f + 1.f) - ix[k]; fx[k] = ix[k] - x0f;
        ay[k] = (y0f + 1.f) - iy[k]; fy[k] = iy[k] - y0f;
        x0[k] = live ? tap_index(x0f) : -2; y0[k] = live ? tap_index(y0f) : -2;
        dead[k] = !(x0[k] >= -1 && x0[k] <= W - 1 && y0[k] >= -1 && y0[k] <= H - 1);
    }
    DmaWindow w;
    {
        int xl = kExtEmptyLo, xh = kExtEmptyHi, yl = kExtEmptyLo, yh = kExtEmptyHi;
        if (!dead[0]) { xl = x0[0]; xh = x0[0] + 1; yl = y0[0]; yh = y0[0] + 1; }
        if (!dead[1]) { xl = min(xl, x0[1]); xh = max(xh, x0[1] + 1); yl = min(yl, y0[1]); yh = max(yh, y0[1] + 1); }
        w.reduce(xl, xh, yl, yh, boxes, wave, lane);
    }
    if (!w.fits()) return false;                          // uniform: a diverged flow
    DmaPlan plan;
    plan.init(w, lane, H, W);
    const int half = (lds_bytes / 2) & ~1023;
    const int nch = min(kDmaMaxCh, half / plan.chan_bytes) & ~3;   // whole groups of four partials per pass
    if (nch < 4) return false;
    const __amdgpu_buffer_rsrc_t rsrc_img = uniform_rsrc(image + static_cast<int64_t>(b) * C * plane, C * plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_go = uniform_rsrc(gout + static_cast<int64_t>(b) * C * plane, C * plane * 2);
    const int go_voff = live ? p * 2 : kDeadOffset;
    const int npass = (C + nch - 1) / nch;
    auto issue = [&](int k) { plan.issue(rsrc_img, lds + (k & 1) * half, k * nch, min(nch, C - k * nch), wave, plane); };
    auto dma_count = [&](int k) { return k >= npass ? 0 : plan.count(min(nch, C - k * nch), wave); };

    const int P2 = w.pitch >> 1;
    const int xr0 = dead[0] ? 0 : x0[0] - w.wx0, xr1 = dead[1] ? 0 : x0[1] - w.wx0;
    const int j0 = (dead[0] ? 0 : (y0[0] - w.wy0) * P2) + (xr0 >> 1), j1 = (dead[1] ? 0 : (y0[1] - w.wy0) * P2) + (xr1 >> 1);
    const unsigned sh0 = (xr0 & 1) * 16u, sh1 = (xr1 & 1) * 16u;
    const f2v ax2 = {ax[0], ax[1]}, fx2 = {fx[0], fx[1]}, ay2 = {ay[0], ay[1]}, fy2 = {fy[0], fy[1]};
    const bool any_dead = __ballot(dead[0] || dead[1]) != 0ull;
    f2v gix[4], giy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) gix[k] = giy[k] = f2v{0.f, 0.f};

    issue(0);
    if (npass > 1) issue(1);
    for (int k = 0; k < npass; ++k) {
        const int n = min(nch, C - k * nch);
        // this wave's gradOutput loads of the previous pass have all been consumed, so everything older has retired:
        // only the copies of pass k + 1 may still be outstanding
        wait_vmcnt_upto(dma_count(k + 1));
        __builtin_amdgcn_s_barrier();
        const unsigned *buf = reinterpret_cast<const unsigned *>(lds + (k & 1) * half);
        const int cdw = plan.chan_bytes >> 2;
        auto channels = [&](auto with_dead) {
            for (int h0 = 0; h0 < n; h0 += 4) {    // k * nch and h0 are multiples of 4: channel k * nch + h0 + u is partial u
                unsigned g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    g[u] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_go, h0 + u < n ? go_voff : kDeadOffset,
                                                                (k * nch + min(h0 + u, n - 1)) * plane * 2, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (h0 + u >= n) break;   // uniform
                    const unsigned *q0 = buf + (h0 + u) * cdw + j0, *q1 = buf + (h0 + u) * cdw + j1;
                    unsigned an = __builtin_amdgcn_alignbit(q0[1], q0[0], sh0), as = __builtin_amdgcn_alignbit(q0[P2 + 1], q0[P2], sh0);
                    unsigned bn = __builtin_amdgcn_alignbit(q1[1], q1[0], sh1), bs = __builtin_amdgcn_alignbit(q1[P2 + 1], q1[P2], sh1);
                    if constexpr (decltype(with_dead)::value) {
                        if (dead[0]) an = as = 0u;
                        if (dead[1]) bn = bs = 0u;
                    }
                    const f2v vnw = {lo16<T>(an), lo16<T>(bn)}, vne = {hi16<T>(an), hi16<T>(bn)};
                    const f2v vsw = {lo16<T>(as), lo16<T>(bs)}, vse = {hi16<T>(as), hi16<T>(bs)};
                    const f2v gg = {lo16<T>(g[u]), hi16<T>(g[u])};
                    // flow_grad_terms (warp_common.h) for the lane's two pixels at once
                    gix[u] = __builtin_elementwise_fma(__builtin_elementwise_fma(vne - vnw, ay2, (vse - vsw) * fy2), gg, gix[u]);
                    giy[u] = __builtin_elementwise_fma(__builtin_elementwise_fma(vsw - vnw, ax2, (vse - vne) * fx2), gg, giy[u]);
                }
            }
        };
        if (any_dead) channels(std::true_type{}); else channels(std::false_type{});
        if (k + 2 < npass) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // every wave has read buffer k & 1
            issue(k + 2);
        }
    }
    f2v sx = {0.f, 0.f}, sy = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) { sx += gix[k]; sy += giy[k]; }
    if (live) {
        float mx[2], my[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            mx[k] = static_cast<float>(W) / 2.0f; my[k] = static_cast<float>(H) / 2.0f;
            if (pad_mode == CERB_PAD_BORDER) {
                if (ix[k] <= 0.f || ix[k] >= static_cast<float>(W - 1)) mx[k] = 0.f;
                if (iy[k] <= 0.f || iy[k] >= static_cast<float>(H - 1)) my[k] = 0.f;
            }
        }
        // autograd order: grad_grid = mult * sum ; through norm_grid: / (size-1) then * 2.0
        const float rx0 = mx[0] * sx.x / static_cast<float>(W - 1) * 2.0f, rx1 = mx[1] * sx.y / static_cast<float>(W - 1) * 2.0f;
        const float ry0 = my[0] * sy.x / static_cast<float>(H - 1) * 2.0f, ry1 = my[1] * sy.y / static_cast<float>(H - 1) * 2.0f;
        F *gf = gflow + static_cast<int64_t>(b) * 2 * plane + p;
        if constexpr (sizeof(F) == 4) {
            *reinterpret_cast<float2 *>(gf) = make_float2(rx0, rx1);
            *reinterpret_cast<float2 *>(gf + plane) = make_float2(ry0, ry1);
        } else {
            *reinterpret_cast<unsigned *>(gf) = narrow2<F>(rx0, rx1);
            *reinterpret_cast<unsigned *>(gf + plane) = narrow2<F>(ry0, ry1);
        }
    }
    return true;
#else
    return false;
#endif
}

}  // namespace
}  // namespace cerb
