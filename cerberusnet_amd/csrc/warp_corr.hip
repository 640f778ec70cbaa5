// warp_corr.hip -- f2 of SURVEY.md 8(f): the flow warp FUSED into the correlation forward.
//
//   out[(dy+4)*9 + (dx+4)][y][x] = leaky( 1/C sum_c f1[c][y][x] * warped[c][y+dy][x+dx] ),   warped = flow_warp(f2, flow)
//   (reference: nnet_training/nnet_models/pwcnet_sfd.py:178 -> :181-182, i.e. loss_functions/UnFlowLoss.py:83-94 followed by
//    correlation_package/correlation_cuda_kernel.cu:29-95 with pad = d = 4, k = 1, s1 = s2 = 1)
//
// The warped feature map is never written: a workgroup samples the (TH + 8) x (TW + 8) window of its 8 x 32 output tile
// straight into LDS -- the bilinear taps of every window pixel with the warp kernels' own coordinate arithmetic
// (warp_common.h: source_coord, the reference's fp32 rounding order), rounded through the storage type as the stand-alone
// warp's output would be -- four channels at a time (the next four's taps in flight meanwhile), and correlates it against f1 from there: a thread owns one output
// pixel and its 81 accumulators.  Window pixels outside the image are the correlation's zero padding.
//
// This is the form rounds 2, 4 and 5 PRICED and rejected without building it (the window's halo makes a tile
// sample (TH + 8)(TW + 8) / (TH TW) = 2.5x the pixels of the stand-alone warp; the saved round trip of `warped` is <= 6 us at
// the top level).  Round 6 builds it so that the row exists as code with parity tests and a measured line
// (bench.py extra.f2_fused, profiles/r06_f2_fused.txt): it loses, as priced, and is opt-in only
// (cerberus::warp_correlation_leaky, PWCNetHead(fuse_warp=True)).  What it does buy is MEMORY: the training path built on it
// (WarpCorrelation in correlation_package/correlation.py) saves neither `warped` nor the warp's context; its backward
// recomputes the warp with the tuned kernels and then runs the tuned correlation / warp backward.
#include "warp_common.h"

namespace cerb {
namespace {

constexpr int kFD = 4, kFND = 2 * kFD + 1;
constexpr int kFTH = 8, kFTW = 32;                        // output tile: one pixel per thread
constexpr int kFWH = kFTH + 2 * kFD, kFWW = kFTW + 2 * kFD;   // window: 16 x 40
constexpr int kFWP = kFWW + 1;                            // LDS row pitch (odd: the nine rows a wave reads fall on different banks)
constexpr int kFCC = 4;                                   // channels per chunk (its 48 tap loads per thread travel during the previous chunk's FMAs)
constexpr int kFWN = (kFWH * kFWW + 255) / 256;           // window pixels per thread (3)

// SPLIT: a small map has too few 8 x 32 tiles for the chip (128 x 32 x 64 at 4 pairs: 32), so `nslice` workgroups share a tile,
// each summing `cslice` channels into an fp32 scratch volume with float atomics; warp_corr_finish_kernel scales, applies the
// LeakyReLU and stores T.  (The order of those adds is not fixed: results agree to fp32 rounding, not bit for bit, run to run.)
template <typename T, typename F, bool SPLIT>
__global__ __launch_bounds__(256) void warp_corr_fwd_kernel(
    const T *__restrict__ f1, const T *__restrict__ f2, const F *__restrict__ flow, T *__restrict__ out, float *__restrict__ scratch,
    int B, int C, int H, int W, int pad_mode, float slope, int64_t out_bstride, int cslice, int nslice) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float win[kFCC][kFWH][kFWP];
    constexpr int esz = sizeof(T);
    const int plane = H * W;
    const int tid = threadIdx.x;
    const int ntx = (W + kFTW - 1) / kFTW, nty = (H + kFTH - 1) / kFTH;
    int id = xcd_chunk(blockIdx.x, gridDim.x);
    const int slice = SPLIT ? id % nslice : 0;
    if constexpr (SPLIT) id /= nslice;
    const int c_begin = slice * cslice, c_end = SPLIT ? min(C, c_begin + cslice) : C;
    const int tx = id % ntx; id /= ntx;
    const int ty = id % nty;
    const int b = id / nty;
    const int x0t = tx * kFTW, y0t = ty * kFTH;
    const __amdgpu_buffer_rsrc_t r2 = uniform_rsrc(f2 + static_cast<int64_t>(b) * C * plane, C * plane * esz);
    const __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(f1 + static_cast<int64_t>(b) * C * plane, C * plane * esz);

    // ---- the thread's window pixels: sample position, tap offsets, weights (once for all channels) ----
    float wnw[kFWN], wne[kFWN], wsw[kFWN], wse[kFWN];
    int onw[kFWN], one[kFWN], osw[kFWN], ose[kFWN], slot[kFWN];
#pragma unroll
    for (int k = 0; k < kFWN; ++k) {
        const int e = tid + 256 * k;
        const int wy = e / kFWW, wx = e - wy * kFWW;
        const int gx = x0t - kFD + wx, gy = y0t - kFD + wy;
        const bool in = e < kFWH * kFWW && gx >= 0 && gx < W && gy >= 0 && gy < H;   // else: the correlation's zero padding
        slot[k] = e < kFWH * kFWW ? wy * kFWP + wx : -1;
        wnw[k] = wne[k] = wsw[k] = wse[k] = 0.f;
        onw[k] = one[k] = osw[k] = ose[k] = kDeadOffset;
        if (in) {
            const F *fl = flow + static_cast<int64_t>(b) * 2 * plane + gy * W + gx;
            const Coord<float> cx = source_coord<float>(gx, static_cast<float>(ld(fl)), W, pad_mode);
            const Coord<float> cy = source_coord<float>(gy, static_cast<float>(ld(fl + plane)), H, pad_mode);
            const float x0f = floorf(cx.pos), y0f = floorf(cy.pos);
            const float x1f = x0f + 1.f, y1f = y0f + 1.f;
            wnw[k] = (x1f - cx.pos) * (y1f - cy.pos);
            wne[k] = (cx.pos - x0f) * (y1f - cy.pos);
            wsw[k] = (x1f - cx.pos) * (cy.pos - y0f);
            wse[k] = (cx.pos - x0f) * (cy.pos - y0f);
            const int x0 = tap_index(x0f), y0 = tap_index(y0f);
            const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
            const bool oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
            const int o00 = static_cast<int>((static_cast<unsigned>(y0) * static_cast<unsigned>(W) + static_cast<unsigned>(x0)) * esz);
            onw[k] = (oky0 && okx0) ? o00 : kDeadOffset;
            one[k] = (oky0 && okx1) ? o00 + esz : kDeadOffset;
            osw[k] = (oky1 && okx0) ? o00 + W * esz : kDeadOffset;
            ose[k] = (oky1 && okx1) ? o00 + (W + 1) * esz : kDeadOffset;
        }
    }

    // ---- the thread's output pixel ----
    const int py = tid / kFTW, px = tid - py * kFTW;
    const int oy = y0t + py, ox = x0t + px;
    const bool live = oy < H && ox < W;
    const int v1 = live ? (oy * W + ox) * esz : kDeadOffset;
    float acc[kFND * kFND];
#pragma unroll
    for (int d = 0; d < kFND * kFND; ++d) acc[d] = 0.f;

    // taps of one chunk for the thread's window pixels: all in flight at once (absent taps and channels past C: zeros)
    float v[kFWN][kFCC][4];
    auto request = [&](int c0) {
#pragma unroll
        for (int k = 0; k < kFWN; ++k)
#pragma unroll
            for (int i = 0; i < kFCC; ++i) {
                const bool on = c0 + i < c_end;
                const int soff = __builtin_amdgcn_readfirstlane(min(c0 + i, c_end - 1) * plane * esz);
                v[k][i][0] = buffer_load_px1<T>(r2, on ? onw[k] : kDeadOffset, soff);
                v[k][i][1] = buffer_load_px1<T>(r2, on ? one[k] : kDeadOffset, soff);
                v[k][i][2] = buffer_load_px1<T>(r2, on ? osw[k] : kDeadOffset, soff);
                v[k][i][3] = buffer_load_px1<T>(r2, on ? ose[k] : kDeadOffset, soff);
            }
    };
    request(c_begin);
    for (int c0 = c_begin; c0 < c_end; c0 += kFCC) {
        const int n = min(kFCC, c_end - c0);
        if (c0 != c_begin) __syncthreads();    // the previous chunk's window has been read
        // blend the chunk's taps into the LDS window
#pragma unroll
        for (int k = 0; k < kFWN; ++k) {
            if (slot[k] >= 0) {
#pragma unroll
                for (int i = 0; i < kFCC; ++i) {
                    float s = v[k][i][0] * wnw[k];     // the warp kernels' order (warp.hip: warp_fwd_kernel)
                    s += v[k][i][1] * wne[k];
                    s += v[k][i][2] * wsw[k];
                    s += v[k][i][3] * wse[k];
                    if constexpr (esz == 2) {            // the stand-alone warp stores T: the correlation then reads the rounded value
                        T t;
                        st(&t, s);
                        s = ld(&t);
                    }
                    (&win[i][0][0])[slot[k]] = s;
                }
            }
        }
        __syncthreads();
        if (c0 + kFCC < c_end) request(c0 + kFCC);   // the next chunk's taps travel during this chunk's FMAs
        for (int i = 0; i < n; ++i) {
            const float a = buffer_load_px1<T>(r1, v1, __builtin_amdgcn_readfirstlane((c0 + i) * plane * esz));
            const float *wr = &win[i][py][px];
#pragma unroll
            for (int dy = 0; dy < kFND; ++dy)
#pragma unroll
                for (int dx = 0; dx < kFND; ++dx) acc[dy * kFND + dx] = fmaf(a, wr[dy * kFWP + dx], acc[dy * kFND + dx]);
        }
    }
    if (!live) return;
    if constexpr (SPLIT) {
        float *sc = scratch + static_cast<int64_t>(b) * (kFND * kFND) * plane + oy * W + ox;
#pragma unroll
        for (int d = 0; d < kFND * kFND; ++d) unsafeAtomicAdd(sc + static_cast<int64_t>(d) * plane, acc[d]);
        return;
    }
    const float inv = 1.0f / static_cast<float>(C);
    const int64_t obs = out_bstride ? out_bstride : static_cast<int64_t>(kFND * kFND) * plane;
    T *o = out + b * obs + oy * W + ox;
#pragma unroll
    for (int d = 0; d < kFND * kFND; ++d) {
        float q = acc[d] * inv;
        q = q > 0.f ? q : q * slope;
        st(o + static_cast<int64_t>(d) * plane, q);
    }
#endif
}

template <typename T>
__global__ __launch_bounds__(256) void warp_corr_finish_kernel(const float *__restrict__ scratch, T *__restrict__ out, int64_t per_item,
                                                               int64_t total, float inv, float slope, int64_t out_bstride) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * 256) {
        const int64_t b = i / per_item, r = i - b * per_item;
        float q = scratch[i] * inv;
        q = q > 0.f ? q : q * slope;
        st(out + b * (out_bstride ? out_bstride : per_item) + r, q);
    }
}

// channel slices per tile of a small map (1: the one-launch form), and the scratch they need
int f2_slices(int B, int C, int H, int W) {
    const int64_t tiles = static_cast<int64_t>(B) * ((W + kFTW - 1) / kFTW) * ((H + kFTH - 1) / kFTH);
    if (tiles >= 256 || C <= 2 * kFCC) return 1;
    const int want = static_cast<int>((256 + tiles - 1) / tiles);
    return std::max(1, std::min(want, (C + kFCC - 1) / kFCC));
}

}  // namespace

int64_t warp_corr_workspace_bytes(int B, int C, int H, int W) {
    return f2_slices(B, C, H, W) > 1 ? static_cast<int64_t>(B) * kFND * kFND * H * W * 4 : 0;
}

// fp32 / fp16 / bf16 storage, pad = d = 4 (the only configuration a model uses); the flow has the image's type or is fp32
int warp_corr_forward(const void *f1, const void *f2, const void *flow, void *out, void *workspace, int64_t workspace_bytes, int B,
                      int C, int H, int W, int pad_mode, float slope, int64_t out_bstride, int dtype, int flow_dtype, hipStream_t s) {
    if (B == 0) return CERB_OK;
    if (dtype != CERB_F32 && dtype != CERB_F16 && dtype != CERB_BF16) return CERB_EUNSUPPORTED;
    if (!(flow_dtype == dtype || flow_dtype == CERB_F32)) return CERB_EDTYPE;
    if (static_cast<int64_t>(C) * H * W * 4 >= 0x7fffffff) return CERB_ETOOLARGE;
    const int64_t tiles = static_cast<int64_t>(B) * ((W + kFTW - 1) / kFTW) * ((H + kFTH - 1) / kFTH);
    int nslice = f2_slices(B, C, H, W);
    // (without the scratch volume a small map takes the one-launch form: correct, with few workgroups)
    if (nslice > 1 && (!workspace || workspace_bytes < warp_corr_workspace_bytes(B, C, H, W) || (reinterpret_cast<uintptr_t>(workspace) & 3))) nslice = 1;
    const int cslice = nslice > 1 ? ((C + nslice - 1) / nslice + kFCC - 1) / kFCC * kFCC : C;
    if (nslice > 1) nslice = (C + cslice - 1) / cslice;
    const int64_t blocks = tiles * nslice;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    float *scratch = static_cast<float *>(workspace);
    const int64_t per_item = static_cast<int64_t>(kFND * kFND) * H * W, total = per_item * B;
    if (nslice > 1) {
        const hipError_t e = hipMemsetAsync(scratch, 0, static_cast<size_t>(total) * 4, s);
        if (e != hipSuccess) return static_cast<int>(e);
    }
#define CERB_LAUNCH_F2(T, F)                                                                                            \
    do {                                                                                                                \
        if (nslice > 1) {                                                                                               \
            hipLaunchKernelGGL((warp_corr_fwd_kernel<T, F, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,  \
                               static_cast<const T *>(f1), static_cast<const T *>(f2), static_cast<const F *>(flow),     \
                               static_cast<T *>(out), scratch, B, C, H, W, pad_mode, slope, out_bstride, cslice, nslice); \
            hipLaunchKernelGGL((warp_corr_finish_kernel<T>), dim3(static_cast<unsigned>(std::min<int64_t>((total + 255) / 256, 4096))), \
                               dim3(256), 0, s, scratch, static_cast<T *>(out), per_item, total, 1.0f / static_cast<float>(C), slope, out_bstride); \
        } else {                                                                                                        \
            hipLaunchKernelGGL((warp_corr_fwd_kernel<T, F, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, \
                               static_cast<const T *>(f1), static_cast<const T *>(f2), static_cast<const F *>(flow),     \
                               static_cast<T *>(out), scratch, B, C, H, W, pad_mode, slope, out_bstride, C, 1);         \
        }                                                                                                               \
    } while (0)
    if (dtype == CERB_F32) CERB_LAUNCH_F2(float, float);
    else if (dtype == CERB_F16) { if (flow_dtype == CERB_F32) CERB_LAUNCH_F2(__half, float); else CERB_LAUNCH_F2(__half, __half); }
    else { if (flow_dtype == CERB_F32) CERB_LAUNCH_F2(hip_bfloat16, float); else CERB_LAUNCH_F2(hip_bfloat16, hip_bfloat16); }
#undef CERB_LAUNCH_F2
    note_kernel(0, "warp_corr_fwd_8x32");
    return launch_status();
}

}  // namespace cerb
