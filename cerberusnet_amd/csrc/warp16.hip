// warp16.hip -- the flow-warp forward for 16-bit storage (fp16 / bf16 under AMP: BASELINE config 5 and the bf16
// pass), two elements per lane and instruction.
//
// Same operation as warp.hip (/root/reference/nnet_training/loss_functions/UnFlowLoss.py:83-94 as called with 16-bit
// features from nnet_models/pwcnet_sfd.py:178), same fp32 arithmetic in the same order -- the outputs and the backward
// context are bit-identical to warp_fwd_kernel's / warp_fwd_staged_kernel's (test) -- but built for the fact that the
// general kernels cost the same microseconds in fp16 as in fp32 (profiles/r05_16bit_kernels.txt 3): they are bound by
// instructions per element (two LDS tap reads, one 2-byte store, one widening per element), not by bytes.  Here
//   * a lane owns TWO horizontally adjacent pixels: every output store is a dword (two pixels of one channel), a wave's
//     store instruction writes two whole 128-byte lines;
//   * the LDS window holds the RAW 16-bit pixels (half the LDS of the general kernel's fp32 window) and is filled by
//     LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging instructions on the VALU, no ds_write, and the next group of
//     channels is in flight while this one is blended (two buffers, counted vmcnt);
//   * the two pixels' products and sums are v_pk_mul_f32 / v_pk_add_f32 (the same IEEE operations as the scalar forms,
//     uncontracted), the two results leave as one v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32;
//   * a workgroup owns an 8 x 64 pixel tile (a wave: 2 rows x 64 columns = the two 2 x 32 strips of the context side
//     by side) for a range of channels.
#include "warp16_common.h"

namespace cerb {
namespace {

#ifdef CERB_STAMP
// diagnostic build only (-DCERB_STAMP): cycle counter of thread 0 at the phase boundaries of the first 64 workgroups
// (blockIdx.x * 8: spread over the XCDs), fetched with cerberus_debug_stamps16(); never compiled into the product
__device__ unsigned long long g_stamps16[64][16];
#define CERB_STAMP16(k)                                                                                     \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 64) g_stamps16[blockIdx.x][k] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define CERB_STAMP16(k) do {} while (0)
#endif

// One pixel's sampling state (the forward's arithmetic of warp_fwd_kernel, verbatim)
struct Px {
    float wnw = 0.f, wne = 0.f, wsw = 0.f, wse = 0.f;
    float px = 0.f, py = 0.f;   // sample position (context)
    int x0 = -2, y0 = -2;       // north-west tap; a pixel outside the image has no in-image tap
    template <typename F>
    __device__ __forceinline__ void init(int x, int y, float fx, float fy, int H, int W, int pad_mode) {
        const Coord<float> cx = source_coord<float>(x, fx, W, pad_mode);
        const Coord<float> cy = source_coord<float>(y, fy, H, pad_mode);
        const float x0f = floorf(cx.pos), y0f = floorf(cy.pos);
        const float x1f = x0f + 1.f, y1f = y0f + 1.f;
        wnw = (x1f - cx.pos) * (y1f - cy.pos);
        wne = (cx.pos - x0f) * (y1f - cy.pos);
        wsw = (x1f - cx.pos) * (cy.pos - y0f);
        wse = (cx.pos - x0f) * (cy.pos - y0f);
        x0 = tap_index(x0f); y0 = tap_index(y0f);
        px = cx.pos; py = cy.pos;
    }
    __device__ __forceinline__ bool dead(int H, int W) const { return !(x0 >= -1 && x0 <= W - 1 && y0 >= -1 && y0 <= H - 1); }
    // the sum a pixel without any in-image tap gets for every channel (0 * weight: NaN under a non-finite flow)
    __device__ __forceinline__ float zsum() const {
        float a = 0.f * wnw; a += 0.f * wne; a += 0.f * wsw; a += 0.f * wse;
        return a;
    }
    __device__ __forceinline__ float blend(float v0, float v1, float v2, float v3) const {
        float acc = v0 * wnw;   // absent taps contribute exact zeros
        acc += v1 * wne;
        acc += v2 * wsw;
        acc += v3 * wse;
        return acc;
    }
    // the same for the two channels of a window dword at once
    __device__ __forceinline__ f2v blend2(f2v v0, f2v v1, f2v v2, f2v v3) const {
        f2v acc = v0 * f2v{wnw, wnw};
        acc += v1 * f2v{wne, wne};
        acc += v2 * f2v{wsw, wsw};
        acc += v3 * f2v{wse, wse};
        return acc;
    }
};

template <typename T, typename F>
__global__ __launch_bounds__(256) void warp_fwd_staged16_kernel(
    const T *__restrict__ image, const F *__restrict__ flow, T *__restrict__ out,
    void *__restrict__ ctx, int B, int C, int H, int W, int pad_mode, int crange, int nrange, int ablate) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int esz = sizeof(T), cpx = 16 / esz;   // 16-bit storage, or fp32 (round 6: the same kernel, 4-pixel cells, no widening)
#ifndef CERB_ABLATE
    ablate = 0;     // timing ablations (WRONG results) exist in -DCERB_ABLATE builds only: 1 no stores, 2 no staging loads, 4 no taps
#endif
    __shared__ __attribute__((aligned(16))) unsigned win[2 * kDmaBuf / 4];
    __shared__ int4 boxes[4];

    const int plane = H * W;   // the launcher guarantees C * plane * 4 < 2^31 and W % 4 == 0
    const int tid = threadIdx.x;
    const int lane = tid & (kPix - 1), wave = __builtin_amdgcn_readfirstlane(tid / kPix);
    const int ntx = (W + kTile16W - 1) / kTile16W, nty = (H + kTile16H - 1) / kTile16H;
    int id = xcd_chunk(blockIdx.x, gridDim.x);
    const int r = __builtin_amdgcn_readfirstlane(id % nrange); id /= nrange;
    const int tx = __builtin_amdgcn_readfirstlane(id % ntx); id /= ntx;
    const int ty = __builtin_amdgcn_readfirstlane(id % nty);
    const int b = __builtin_amdgcn_readfirstlane(id / nty);
    if (ablate & 32) return;
    CERB_STAMP16(0);
    const int y = ty * kTile16H + wave * 2 + (lane >> 5);
    const int xa = tx * kTile16W + 2 * (lane & 31);       // the lane's pixels: xa, xa + 1 (W % 4 == 0: both inside or both outside)
    const bool live = y < H && xa < W;
    const int p = y * W + xa;
    Px px[2];
    if (live) {
        const F *fl = flow + static_cast<int64_t>(b) * 2 * plane + p;
        float fx[2], fy[2];
        if constexpr (sizeof(F) == 4) {
            const float2 vx = *reinterpret_cast<const float2 *>(fl), vy = *reinterpret_cast<const float2 *>(fl + plane);
            fx[0] = vx.x; fx[1] = vx.y; fy[0] = vy.x; fy[1] = vy.y;
        } else {
            const unsigned vx = *reinterpret_cast<const unsigned *>(fl), vy = *reinterpret_cast<const unsigned *>(fl + plane);
            fx[0] = lo16<F>(vx); fx[1] = hi16<F>(vx); fy[0] = lo16<F>(vy); fy[1] = hi16<F>(vy);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) px[k].template init<F>(xa + k, y, fx[k], fy[k], H, W, pad_mode);
    }
    CERB_STAMP16(1);
    if (ablate & 64) { if (px[0].wnw + px[1].wse == 1.2345f) win[0] = 1; return; }
    // context: positions + the tap range of the lane's 2 x 32 strip (DPP rows 0, 2: the left strip; 1, 3: the right one)
    const Strips strips(H, W);
    const int jy = ty * (kTile16H / kStripH) + wave;
    if (ctx && r == 0) {
        TapRange range;
        if (live) {
            float *pos = ctx_pos(ctx, B, H, W) + static_cast<int64_t>(b) * 2 * plane + p;
            *reinterpret_cast<float2 *>(pos) = make_float2(px[0].px, px[1].px);
            *reinterpret_cast<float2 *>(pos + plane) = make_float2(px[0].py, px[1].py);
#pragma unroll
            for (int k = 0; k < 2; ++k) range.add(xa + k, y, px[k].x0, px[k].y0, W, H);
        }
        if (jy < strips.ny) {
            const int xl = row_minmax<false>(range.xlo), xh = row_minmax<true>(range.xhi);
            const int yl = row_minmax<false>(range.ylo), yh = row_minmax<true>(range.yhi);
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int l0 = 15 + 16 * side, l1 = 47 + 16 * side;
                const int4 e = make_int4(min(__builtin_amdgcn_readlane(xl, l0), __builtin_amdgcn_readlane(xl, l1)),
                                         max(__builtin_amdgcn_readlane(xh, l0), __builtin_amdgcn_readlane(xh, l1)),
                                         min(__builtin_amdgcn_readlane(yl, l0), __builtin_amdgcn_readlane(yl, l1)),
                                         max(__builtin_amdgcn_readlane(yh, l0), __builtin_amdgcn_readlane(yh, l1)));
                const int jx = tx * 2 + side;
                if (lane == 0 && jx < strips.nx)
                    static_cast<int4 *>(ctx)[b * strips.per_image() + jy * strips.nx + jx] = e;
            }
        }
    }
    CERB_STAMP16(2);
    if (ablate & 128) { if (px[0].wnw + px[1].wse == 1.2345f) win[0] = 1; return; }
    const bool dead0 = px[0].dead(H, W), dead1 = px[1].dead(H, W);
    const float zs0 = px[0].zsum(), zs1 = px[1].zsum();
    DmaWindow w;
    {
        int xl = kExtEmptyLo, xh = kExtEmptyHi, yl = kExtEmptyLo, yh = kExtEmptyHi;
        if (!dead0) { xl = px[0].x0; xh = px[0].x0 + 1; yl = px[0].y0; yh = px[0].y0 + 1; }
        if (!dead1) { xl = min(xl, px[1].x0); xh = max(xh, px[1].x0 + 1); yl = min(yl, px[1].y0); yh = max(yh, px[1].y0 + 1); }
        w.reduce(xl, xh, yl, yh, boxes, wave, lane, cpx);
    }
    CERB_STAMP16(3);
    if (ablate & 256) { if (w.cells == 12345) win[0] = 1; return; }

    const int c_begin = r * crange, c_end = min(C, c_begin + crange);
    const T *img = image + static_cast<int64_t>(b) * C * plane;
    const __amdgpu_buffer_rsrc_t rsrc_img = uniform_rsrc(img, C * plane * esz);
    const __amdgpu_buffer_rsrc_t rsrc_out = uniform_rsrc(out + static_cast<int64_t>(b) * C * plane, C * plane * esz);
    const int out_voff = live ? p * esz : kDeadOffset;
    auto store2 = [&](int c, float a, float bv) {
        if (ablate & 1) { if (a == 1.2345f) win[0] = 1; return; }
        if constexpr (esz == 2) {
            __builtin_amdgcn_raw_buffer_store_b32(narrow2<T>(a, bv), rsrc_out, out_voff, c * plane * 2, 0);
        } else {
            typedef unsigned u2s __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(u2s{__float_as_uint(a), __float_as_uint(bv)}, rsrc_out, out_voff, c * plane * 4, 0);
        }
    };
    if (w.empty) {   // no tap of the tile is inside the image
        for (int c = c_begin; c < c_end; ++c) store2(c, zs0, zs1);
        return;
    }
    if (!w.fits()) {
        // diverged flow: direct gather, four channels in flight per pixel
        if (!live) return;
        for (int c = c_begin; c < c_end; c += 4) {
            float res[2][4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int x0 = px[k].x0, y0 = px[k].y0;
                const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
                const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
                const int64_t o00 = static_cast<int64_t>(y0) * W + x0;
                float v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const T *q = img + static_cast<int64_t>(min(c + u, c_end - 1)) * plane + o00;
                    load_taps<true, T, float>(q, oky0 && okx0, oky0 && okx1, v[u][0], v[u][1]);
                    load_taps<true, T, float>(q + W, oky1 && okx0, oky1 && okx1, v[u][2], v[u][3]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) res[k][u] = px[k].blend(v[u][0], v[u][1], v[u][2], v[u][3]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c + u < c_end) store2(c + u, res[0][u], res[1][u]);
        }
        return;
    }

    // ---- the lane's cells: source offsets of DMA instruction q (cell 64 q + lane), once for all channels ----
    DmaPlan plan;
    plan.init(w, lane, H, W, esz);
    const int chan_bytes = plan.chan_bytes;
    const int nch = min(kDmaMaxCh, kDmaBuf / chan_bytes);                 // channels per pass (>= 3)
    const int ntot = c_end - c_begin;
    const int npass = (ntot + nch - 1) / nch;
    auto issue = [&](int k) {
        if (ablate & 2) return;
        plan.issue(rsrc_img, reinterpret_cast<char *>(win) + (k & 1) * kDmaBuf, c_begin + k * nch, min(nch, ntot - k * nch), wave, plane);
    };
    // this wave's copy instructions of pass k
    auto dma_count = [&](int k) { return (k >= npass || (ablate & 2)) ? 0 : plan.count(min(nch, ntot - k * nch), wave); };

    // taps: dword index of the pixel pair that holds the north-west tap, and its parity
    // (fp32: a dword IS a pixel -- P2 is the row pitch in dwords either way)
    const int P2 = esz == 2 ? w.pitch >> 1 : w.pitch;
    const int xr0 = dead0 ? 0 : px[0].x0 - w.wx0, xr1 = dead1 ? 0 : px[1].x0 - w.wx0;
    const int j0 = (dead0 ? 0 : (px[0].y0 - w.wy0) * P2) + (esz == 2 ? xr0 >> 1 : xr0);
    const int j1 = (dead1 ? 0 : (px[1].y0 - w.wy0) * P2) + (esz == 2 ? xr1 >> 1 : xr1);
    [[maybe_unused]] const unsigned sh0 = (xr0 & 1) * 16u, sh1 = (xr1 & 1) * 16u;
    const f2v wnw = {px[0].wnw, px[1].wnw}, wne = {px[0].wne, px[1].wne};
    const f2v wsw = {px[0].wsw, px[1].wsw}, wse = {px[0].wse, px[1].wse};
    // a lane without any in-image tap (zeros padding far outside, a non-finite flow) stores its zsum instead of the
    // blend: rare, so the two selects per channel sit behind a wave-uniform branch
    const bool any_dead = __ballot(live && (dead0 || dead1)) != 0ull;
    CERB_STAMP16(4);
    if (ablate & 512) { if (j0 + plan.voff[0] + plan.voff[3] + nch == j1) win[1] = 1; return; }

    issue(0);
    if (npass > 1) issue(1);
    CERB_STAMP16(5);
    int stores_prev = 0;     // output stores this wave issued in the previous pass (younger than this pass's copies)
    for (int k = 0; k < npass; ++k) {
        const int n = min(nch, ntot - k * nch);
        // pass k's copies have landed once only the younger operations remain: the previous pass's output stores
        // and the copies of pass k + 1
        wait_vmcnt_upto(stores_prev + dma_count(k + 1));
        __builtin_amdgcn_s_barrier();                     // ... and every other wave's too
        if (k == 0) CERB_STAMP16(6);
        const unsigned *buf = win + (k & 1) * (kDmaBuf / 4);
        auto channels = [&](auto with_dead) {
            const unsigned *q0 = buf + j0, *q0s = buf + j0 + P2, *q1 = buf + j1, *q1s = buf + j1 + P2;
            const int cdw = chan_bytes >> 2;
            for (int i = 0; i < n; ++i, q0 += cdw, q0s += cdw, q1 += cdw, q1s += cdw) {
                const int c = c_begin + k * nch + i;
                if (ablate & 4) { store2(c, zs0, zs1); continue; }
                f2v acc;
                if constexpr (esz == 2) {
                    // (north-west, north-east) and (south-west, south-east) of both pixels as packed 16-bit pairs
                    const unsigned an = __builtin_amdgcn_alignbit(q0[1], q0[0], sh0), as = __builtin_amdgcn_alignbit(q0s[1], q0s[0], sh0);
                    const unsigned bn = __builtin_amdgcn_alignbit(q1[1], q1[0], sh1), bs = __builtin_amdgcn_alignbit(q1s[1], q1s[0], sh1);
                    acc = f2v{lo16<T>(an), lo16<T>(bn)} * wnw;   // pixel 0 | pixel 1; absent taps contribute exact zeros
                    acc += f2v{hi16<T>(an), hi16<T>(bn)} * wne;
                    acc += f2v{lo16<T>(as), lo16<T>(bs)} * wsw;
                    acc += f2v{hi16<T>(as), hi16<T>(bs)} * wse;
                } else {
                    acc = f2v{__uint_as_float(q0[0]), __uint_as_float(q1[0])} * wnw;
                    acc += f2v{__uint_as_float(q0[1]), __uint_as_float(q1[1])} * wne;
                    acc += f2v{__uint_as_float(q0s[0]), __uint_as_float(q1s[0])} * wsw;
                    acc += f2v{__uint_as_float(q0s[1]), __uint_as_float(q1s[1])} * wse;
                }
                if constexpr (decltype(with_dead)::value) {
                    if (dead0) acc.x = zs0;
                    if (dead1) acc.y = zs1;
                }
                store2(c, acc.x, acc.y);
            }
        };
        if (any_dead) channels(std::true_type{}); else channels(std::false_type{});
        stores_prev = (ablate & 1) ? 0 : n;
        if (k == 0) CERB_STAMP16(7);
        if (k + 2 < npass) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // every wave has read buffer k & 1: it can be overwritten
            issue(k + 2);
        }
    }
    CERB_STAMP16(8);
#endif
}

}  // namespace

#ifdef CERB_STAMP
extern "C" int cerberus_debug_stamps16(void *dst, int bytes) {
    return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps16), std::min<size_t>(bytes, sizeof(g_stamps16))));
}
#endif

// The 16-bit forward through the DMA window; CERB_EUNSUPPORTED when it does not apply (the caller then takes the
// general kernels).  Preconditions checked by the caller: bilinear, image 16-byte aligned, C * H * W * 4 < 2^31.
int warp16_forward(const void *image, const void *flow, void *out, void *ctx, int B, int C, int H, int W, int pad_mode,
                   int dtype, int flow_dtype, int crange_opt, hipStream_t s) {
    if (dtype != CERB_F16 && dtype != CERB_BF16 && dtype != CERB_F32) return CERB_EUNSUPPORTED;
    if (W % (dtype == CERB_F32 ? 4 : 8)) return CERB_EUNSUPPORTED;     // 16-byte cells
    if ((reinterpret_cast<uintptr_t>(out) & 7) || (reinterpret_cast<uintptr_t>(flow) & 7)) return CERB_EUNSUPPORTED;
    const int ntx = (W + kTile16W - 1) / kTile16W, nty = (H + kTile16H - 1) / kTile16H;
    const int64_t tiles = static_cast<int64_t>(B) * ntx * nty;
    // channels per workgroup: as few as keep the launch at <= 1024 workgroups, between 8 and 32 (the rule of the general
    // staged kernel: the box reduction and the window set-up amortise over the channels)
    int crange = 8;
    while (crange < 32 && tiles * ((C + crange - 1) / crange) > 1024) crange *= 2;
    if (crange_opt >= 4) crange = crange_opt;
    crange = std::max(1, std::min(C, crange));
    const int nrange = (C + crange - 1) / crange;
    const int64_t blocks = tiles * nrange;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    // a small map gives too few 8 x 64 tiles to fill the chip: the general kernel's 8 x 32 tiles do better there
    // (128 x 32 x 64 at 4 pairs: 256 workgroups here, 6.5 us, against 512 and 6.2 us; 64 x 64 x 128: 512, 8.0 against 8.6)
    if (blocks < 512 && crange_opt < 4) return CERB_EUNSUPPORTED;
    // (a phase shift of the workgroups that share a CU -- the k-th of a CU starting k x 1 .. 8 k cycles late, so that one's set-up
    // runs beside another's channel phase -- was tried and loses at every step and level: 23.7 -> 23.9 .. 28.3 us at 32 x 256 x 512)
#define CERB_LAUNCH16(T, F)                                                                                        \
    hipLaunchKernelGGL((warp_fwd_staged16_kernel<T, F>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,     \
                       static_cast<const T *>(image), static_cast<const F *>(flow), static_cast<T *>(out), ctx, B, \
                       C, H, W, pad_mode, crange, nrange, debug_mask())
    if (dtype == CERB_F32) {
        CERB_LAUNCH16(float, float);
    } else if (dtype == CERB_F16) {
        if (flow_dtype == CERB_F32) CERB_LAUNCH16(__half, float); else CERB_LAUNCH16(__half, __half);
    } else {
        if (flow_dtype == CERB_F32) CERB_LAUNCH16(hip_bfloat16, float); else CERB_LAUNCH16(hip_bfloat16, hip_bfloat16);
    }
#undef CERB_LAUNCH16
    return launch_status();
}

}  // namespace cerb
