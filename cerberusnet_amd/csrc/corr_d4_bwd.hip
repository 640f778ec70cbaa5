// corr_d4_bwd.hip -- the TILE formulations of the d = 4 correlation backward (fp32 on the vector unit, 16-bit storage without
// the matrix cores): register-staged and LDS-DMA all-81-per-lane kernels, the three-displacement-group kernel, the
// displacement-row streaming kernel.  The benched pyramid does not reach them any more -- its levels are
// served by corr_strip.hip (256- and 128-wide maps) and corr_coarse.hip (W <= 64) -- they are the general fallback of
// corr_d4_backward for every other width, batch size and alignment (KITTI-sized 1280 x 384 pyramids, W % 4 != 0, maps with too
// few rows for the strip kernel), and the dispatch that chooses among all of them lives here.
//
//   gI1[n][c][y][x] = 1/C * sum_d gO[n][d][y][x]       * x2[n][c][y+dy][x+dx]       (correlation_cuda_kernel.cu:97-172)
//   gI2[n][c][y][x] = 1/C * sum_d gO[n][d][y-dy][x-dx] * x1[n][c][y-dy][x-dx]       (correlation_cuda_kernel.cu:174-242)
#include "corr_d4_common.h"

namespace cerb {
namespace {

// ============================================================================
// backward
// ============================================================================
// TSXP pixel pairs per tile row (tile width 2*TSXP); a wavefront covers
// 64/TSXP rows, the workgroup's 4 wavefronts stack vertically.
// RS is the halo row stride: TW+8 for TSXP=32 (each 32-lane half of a
// ds_read_b64 is one row = 64 consecutive dwords), 96 for TSXP=16 (rows r,r+1
// of a half must differ by 32 banks).
template <int TSXP_, int CC_, int RS_, int NW_ = 4>
struct BwdCfg {
    static constexpr int TSXP = TSXP_, CC = CC_, RS = RS_;
    static constexpr int TW = 2 * TSXP;
    static constexpr int RPW = 64 / TSXP;      // rows per wavefront
    static constexpr int NW = NW_;             // wavefronts per workgroup (stacked vertically)
    static constexpr int TH = NW * RPW;
    static constexpr int HR = TH + 2 * kD;
    static constexpr int HW4 = (TW + 2 * kD) / 4;
    static constexpr int PS = HR * RS;
    static constexpr int THREADS = 64 * NW;
    static constexpr int N = CC * HR * HW4;
    static constexpr int NSLOT = (N + THREADS - 1) / THREADS;
    static constexpr int BUF = CC * PS;
    static constexpr size_t LDS_BYTES = 2 * sizeof(float) * BUF;
};

__device__ __forceinline__ float2 ld2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
struct __attribute__((packed, aligned(4))) float2_u { float x, y; };  // dword-aligned pair
__device__ __forceinline__ float2 ld2u(const float *p) {
    const float2_u t = *reinterpret_cast<const float2_u *>(p);
    return make_float2(t.x, t.y);
}

template <typename K, typename T, bool VEC>
__global__ __launch_bounds__(K::THREADS, 2) void corr_bwd_d4_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, const T *__restrict__ gout,
    T *__restrict__ gin1, T *__restrict__ gin2, int C, int H, int W, int tiles_x,
    int tiles_y, int cslice, int nslice, int dbg) {
#ifndef CERB_ABLATE
    dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CC = K::CC;
    if (dbg & 64) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = wave * K::RPW + lane / K::TSXP;
    const int sxp = lane % K::TSXP;

    // (tile, channel slice, side) with the side fastest: the two workgroups that read
    // the same gradOutput tile are adjacent in the swizzled order -> same XCD, same time
    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int side = bid & 1; bid >>= 1;         // 0: gradInput1, 1: gradInput2
    const int slice = bid % nslice; bid /= nslice;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx * K::TW, y0 = ty * K::TH;
    const int c_begin = slice * cslice;
    const int c_end = min(C, c_begin + cslice);
    const int plane = H * W;

    const T *src = (side == 0 ? x2 : x1) + static_cast<int64_t>(b) * C * plane;
    T *dstb = (side == 0 ? gin1 : gin2) + static_cast<int64_t>(b) * C * plane;
    const T *gob = gout + static_cast<int64_t>(b) * (kND * kND) * plane;

    const int y = y0 + r, x = x0 + 2 * sxp;
    const bool live = y < H && x < W;

    // ---- staging descriptors ----
    int goff[K::NSLOT];  // element offset from `src` for channel 0 of a chunk, <0: slot reads zeros
    int loff[K::NSLOT], gx0[K::NSLOT], chi[K::NSLOT];
#pragma unroll
    for (int j = 0; j < K::NSLOT; ++j) {
        const int id = tid + j * K::THREADS;
        goff[j] = -1; loff[j] = -1; gx0[j] = 0; chi[j] = 0;
        if (id < K::N) {
            const int pl = id / (K::HR * K::HW4);
            const int rem = id % (K::HR * K::HW4);
            const int row = rem / K::HW4, c4 = rem % K::HW4;
            const int gy = y0 - kD + row, gx = x0 - kD + 4 * c4;
            loff[j] = pl * K::PS + row * K::RS + 4 * c4;
            gx0[j] = gx; chi[j] = pl;
            const bool in = gy >= 0 && gy < H && (VEC ? (gx >= 0 && gx < W) : (gx > -4 && gx < W));
            // +4 keeps the offset non-negative for gx in (-4, 0) on the scalar path
            if (in) goff[j] = pl * plane + gy * W + gx + 4;
        }
    }
    float4 stage[K::NSLOT];
    auto prefetch = [&](int c_first) {
#pragma unroll
        for (int j = 0; j < K::NSLOT; ++j) {
            const bool on = goff[j] >= 0 && c_first + chi[j] < c_end;
            if (VEC) {
                const T *p = on ? src + static_cast<int64_t>(c_first) * plane + (goff[j] - 4)
                                : reinterpret_cast<const T *>(g_zero16);
                stage[j] = Gmem<T>::load4(p);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (on) {
                    const T *p = src + static_cast<int64_t>(c_first) * plane + (goff[j] - 4);
                    const int gx = gx0[j];
                    if (gx >= 0 && gx < W) v.x = Gmem<T>::load1(p);
                    if (gx + 1 >= 0 && gx + 1 < W) v.y = Gmem<T>::load1(p + 1);
                    if (gx + 2 >= 0 && gx + 2 < W) v.z = Gmem<T>::load1(p + 2);
                    if (gx + 3 >= 0 && gx + 3 < W) v.w = Gmem<T>::load1(p + 3);
                }
                stage[j] = v;
            }
        }
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int j = 0; j < K::NSLOT; ++j)
            if (loff[j] >= 0) st4(buf + loff[j], stage[j]);
    };

    prefetch(c_begin);  // in flight while the gradOutput registers are gathered

    // ---- the 81 gradOutput values of this lane's two pixels, in registers ----
    // side 0: g[d][p] = gO[d][y][x+p]
    // side 1: g[d][p] = gO[80-d][y+dy][x+p+dx]   (d = (dy+4)*9 + dx+4), 0 outside
    // Stored so that every packed FMA pairs two horizontal displacements whose window
    // operands form an ALIGNED float2 of the LDS row segment w[0..9]:
    //   pixel 0: pairs dx=(0,1)(2,3)(4,5)(6,7) * w[(0,1)..(6,7)], single dx=8 * w[8]
    //   pixel 1: single dx=0 * w[1], pairs dx=(1,2)(3,4)(5,6)(7,8) * w[(2,3)..(8,9)]
    float2v g0p[kND][4], g1p[kND][4];
    float g0s[kND], g1s[kND];
    {
        // Branch-free gather (the first version, one guarded load per value, spent ~14 us of
        // EVERY backward launch here: 1300 basic blocks of bounds checks).  The address of
        // value d is  gob + [uniform offset of d] + [per-lane pixel offset]; validity of the
        // shifted taps is three 9-bit masks computed once per lane; dead lanes read their
        // own pixel of plane 0 (always in bounds) and the value is zeroed afterwards.
        const int lane_off = live ? y * W + x : 0;
        unsigned ymask = 0, xmask0 = 0, xmask1 = 0;
#pragma unroll
        for (int k = 0; k < kND; ++k) {
            const int yy = y + k - kD, xx = x + k - kD;
            if (yy >= 0 && yy < H) ymask |= 1u << k;
            if (xx >= 0 && xx < W) xmask0 |= 1u << k;
            if (xx + 1 >= 0 && xx + 1 < W) xmask1 |= 1u << k;
        }
        if (!live) ymask = 0;
        const bool pair_ok = live && (VEC || x + 1 < W);
#pragma unroll
        for (int d = 0; d < kND * kND; ++d) {
            const int dyi = d / kND, dxi = d % kND;
            float v0, v1;
            // no arithmetic on the loaded values: a select after the load would make every
            // load wait for its data before the next one issues (162 serialised round trips)
            if (side == 0) {  // wave-uniform
                const T *pd = gob + static_cast<int64_t>(d) * plane;  // scalar base
                if (VEC) {
                    const float2 t = Gmem<T>::load2(pd + lane_off);  // dead lanes: value unused
                    v0 = t.x; v1 = t.y;
                } else {
                    v0 = Gmem<T>::load1(pd + lane_off);
                    v1 = Gmem<T>::load1(pd + (pair_ok ? lane_off + 1 : lane_off));
                }
            } else {
                // gO[80-d][y+dy][x+dx (+1)]; taps outside the image read the zero block
                const int uni = (kND * kND - 1 - d) * plane + (dyi - kD) * W + (dxi - kD);
                const bool oky = (ymask >> dyi) & 1u;
                const bool ok0 = oky && ((xmask0 >> dxi) & 1u);
                const bool ok1 = oky && ((xmask1 >> dxi) & 1u);
                const T *zero = reinterpret_cast<const T *>(g_zero16);
                // two dword loads, no branch: a pair load plus a patch branch for border lanes
                // was tried and re-serialised the whole gather (59 vs 46 us at level 3)
                v0 = Gmem<T>::load1(ok0 ? gob + (uni + lane_off) : zero);
                v1 = Gmem<T>::load1(ok1 ? gob + (uni + lane_off + 1) : zero);
            }
            if (dxi == 8) g0s[dyi] = v0; else if (dxi & 1) g0p[dyi][dxi / 2].y = v0; else g0p[dyi][dxi / 2].x = v0;
            if (dxi == 0) g1s[dyi] = v1; else if (dxi & 1) g1p[dyi][(dxi - 1) / 2].x = v1; else g1p[dyi][(dxi - 1) / 2].y = v1;
        }
    }

    if (dbg & 32) {  // ablation: gather only
        if (live) Gmem<T>::store1(dstb + static_cast<int64_t>(c_begin) * plane + y * W + x,
                                  g0s[0] + g1s[8] + g0p[4][2].x + stage[0].x);
        return;
    }
    const float inv_nelems = 1.0f / static_cast<float>(C);
    int it = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += CC, ++it) {
        float *buf = smem + (it & 1) * K::BUF;
        commit(buf);
        __syncthreads();
        if (!(dbg & 2)) prefetch(c0 + CC);  // unconditional: zeros past the end of the slice
        const float *wbase = buf + r * K::RS + 2 * sxp;
        // results are kept in registers and stored after the channel loop: a store
        // inside a rolled inner loop made hipcc drain vmcnt(0) before the loop, which
        // serialised the prefetch above against the FMAs (seen in the first version)
        float res[CC][2] = {};
#pragma unroll 1
        for (int i = 0; i < CC; ++i) {
            if (dbg & 4) break;
            const float *wp = wbase + i * K::PS;
            // three independent packed accumulator chains per pixel (rows mod 3)
            float2v a0[3] = {float2v{0.f, 0.f}, float2v{0.f, 0.f}, float2v{0.f, 0.f}};
            float2v a1[3] = {float2v{0.f, 0.f}, float2v{0.f, 0.f}, float2v{0.f, 0.f}};
            float s0 = 0.f, s1 = 0.f;
            // row r+1's reads ahead of row r's FMAs, one lgkmcnt(5) per row (see bwd_dma_step);
            // not where it would spill (the fp32 8x64 tile also holds its staging registers)
            constexpr bool kPipe = K::TSXP == 16 || sizeof(T) == 2;
            float2v w[kND][5];
#pragma unroll
            for (int q = 0; q < 5; ++q) w[0][q] = ld2v_nomerge(wp + 2 * q);
#pragma unroll
            for (int dyi = 0; dyi < kND; ++dyi) {
                if (kPipe) {
                    if (dyi + 1 < kND) {
#pragma unroll
                        for (int q = 0; q < 5; ++q)
                            w[dyi + 1][q] = ld2v_nomerge(wp + (dyi + 1) * K::RS + 2 * q);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (dyi + 1 < kND) __builtin_amdgcn_s_waitcnt(0xC57F);
                    else __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (dyi > 0) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) w[dyi][q] = ld2v_nomerge(wp + dyi * K::RS + 2 * q);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a0[dyi % 3] = pkfma(g0p[dyi][j], w[dyi][j], a0[dyi % 3]);
                    a1[dyi % 3] = pkfma(g1p[dyi][j], w[dyi][j + 1], a1[dyi % 3]);
                }
                s0 = fmaf(g0s[dyi], w[dyi][4].x, s0);
                s1 = fmaf(g1s[dyi], w[dyi][0].y, s1);
                if (kPipe) __builtin_amdgcn_sched_barrier(0);
            }
            const float2v t0 = a0[0] + a0[1] + a0[2], t1 = a1[0] + a1[1] + a1[2];
            const float r0 = (t0.x + t0.y + s0) * inv_nelems, r1 = (t1.x + t1.y + s1) * inv_nelems;
            // static register indices only (a runtime-indexed array would go to scratch)
#pragma unroll
            for (int q = 0; q < CC; ++q)
                if (q == i) { res[q][0] = r0; res[q][1] = r1; }
        }
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            if (live && c0 + i < c_end && !((dbg & 1) && (c0 + i) != c_begin)) {
                T *dst = dstb + static_cast<int64_t>(c0 + i) * plane + y * W + x;
                if (VEC) {
                    Gmem<T>::stream2(dst, res[i][0], res[i][1]);
                } else {
                    Gmem<T>::store1(dst, res[i][0]);
                    if (x + 1 < W) Gmem<T>::store1(dst + 1, res[i][1]);
                }
            }
        }
    }
}

// ============================================================================
// backward, LDS-DMA variant (fp32, vector path)
// ============================================================================
// Same ownership and arithmetic as corr_bwd_d4_kernel (lane = 2 pixels x 81 gradOutput
// registers).  The channel window no longer passes through VGPRs: every wavefront issues
// buffer_load ... lds for chunk k+NB-1 into a ring of NB LDS buffers, then computes chunk
// k.  Two things make that legal AND fast from plain HIP:
//  * the chunk step is an inlined function whose read / write buffers are __restrict__:
//    the alias scopes let hipcc's waitcnt pass see that the ds_reads cannot touch the
//    buffer a DMA is filling (without them it drains vmcnt(0) before every ds_read);
//  * every vector-memory operation of the loop is unconditional (buffer resources: dead
//    lanes and channels past the slice use an out-of-range offset, which reads zeros /
//    drops the store), so "chunk k has landed" is a fixed s_waitcnt vmcnt(N).  N counts
//    only the younger DMA loads, which is correct whether or not stores retire in order
//    with loads.
template <int CC_, int NB_, int NW_ = 4, int LPI_ = 48, int TSXP_ = 32>
struct BwdDmaCfg {
    static constexpr int CC = CC_, NB = NB_;
    static constexpr int TSXP = TSXP_, TW = 2 * TSXP, RPW = 64 / TSXP, NW = NW_, TH = NW * RPW;
    static constexpr int HR = TH + 2 * kD, HW4 = (TW + 2 * kD) / 4, RS = HW4 * 4, PS = HR * RS;
    static constexpr int THREADS = 64 * NW;
    static constexpr int LPI = LPI_;                      // active lanes per DMA instruction
    static constexpr int SLOTS = CC * HR * HW4;           // 16-byte slots per chunk
    static constexpr int DW = SLOTS / (NW * LPI);         // DMA instructions per wave per chunk
    static constexpr int BUF = CC * PS;
    static constexpr size_t LDS_BYTES = sizeof(float) * NB * BUF;
    static constexpr int WAITN = (NB - 2) * DW;
    static_assert(DW * NW * LPI == SLOTS && (HR * HW4) % LPI == 0, "DMA partition");
    static_assert(WAITN <= 63, "vmcnt is 6 bits");
};

#if defined(__HIP_DEVICE_COMPILE__)
template <typename K>
__device__ __forceinline__ void bwd_dma_issue(float *__restrict__ wr, __amdgpu_buffer_rsrc_t rsrc,
                                              const int (&voff)[K::DW], int wave, int lane,
                                              int c_first, int c_end, int plane) {
    constexpr int kDead = static_cast<int>(0x80000000u);
    const int soff = __builtin_amdgcn_readfirstlane(c_first * plane * 4);
    bool chok[K::DW];
#pragma unroll
    for (int q = 0; q < K::DW; ++q)
        chok[q] = c_first + (wave + K::NW * q) / (K::HR * K::HW4 / K::LPI) < c_end;
    if (lane < K::LPI) {
#pragma unroll
        for (int q = 0; q < K::DW; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (lds_void_ptr)(wr + (wave + K::NW * q) * (K::LPI * 4)), 16,
                chok[q] ? voff[q] : kDead, soff, 0, 0);
    }
}

// one chunk: start the DMA of a later chunk into `wr`, consume the chunk in `rd`
template <typename K>
__device__ __forceinline__ void bwd_dma_step(
    const float *__restrict__ rd, float *__restrict__ wr, __amdgpu_buffer_rsrc_t rsrc_src,
    __amdgpu_buffer_rsrc_t rsrc_dst, const int (&voff)[K::DW], int wave, int lane, int c0,
    int c_next, int c_end, int plane, const float2v (&g0p)[kND][4], const float2v (&g1p)[kND][4],
    const float (&g0s)[kND], const float (&g1s)[kND], int woff, int dst_voff, float inv_nelems,
    int dbg) {
    constexpr int kDead = static_cast<int>(0x80000000u);
    constexpr int CC = K::CC;
    if (!(dbg & 2)) bwd_dma_issue<K>(wr, rsrc_src, voff, wave, lane, c_next, c_end, plane);
    const float *wbase = rd + woff;
    float res[CC][2] = {};
#pragma unroll 1
    for (int i = 0; i < CC; ++i) {
        if (dbg & 4) break;
        const float *wp = wbase + i * K::PS;
        float2v a0[3] = {float2v{0.f, 0.f}, float2v{0.f, 0.f}, float2v{0.f, 0.f}};
        float2v a1[3] = {float2v{0.f, 0.f}, float2v{0.f, 0.f}, float2v{0.f, 0.f}};
        float s0 = 0.f, s1 = 0.f;
        // Row r+1's window reads are issued BEFORE row r's FMAs and pinned there: left to
        // itself hipcc reads a row, waits for its first value at once and steps lgkmcnt(4..1)
        // through the FMAs -- four waits per row and the LDS latency exposed nine times per
        // channel in an issue-bound loop.  Pipelined, one lgkmcnt(5) per row remains.
        float2v w[kND][5];
#pragma unroll
        for (int q = 0; q < 5; ++q) w[0][q] = ld2v_nomerge(wp + 2 * q);
#pragma unroll
        for (int dyi = 0; dyi < kND; ++dyi) {
            if (dyi + 1 < kND) {
#pragma unroll
                for (int q = 0; q < 5; ++q) w[dyi + 1][q] = ld2v_nomerge(wp + (dyi + 1) * K::RS + 2 * q);
            }
            __builtin_amdgcn_sched_barrier(0);
            // this row has landed once at most the next row's 5 reads are pending (in-order
            // return): s_waitcnt lgkmcnt(5) / (0) with vmcnt and expcnt left at their maxima
            if (dyi + 1 < kND) __builtin_amdgcn_s_waitcnt(0xC57F);
            else __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0[dyi % 3] = pkfma(g0p[dyi][j], w[dyi][j], a0[dyi % 3]);
                a1[dyi % 3] = pkfma(g1p[dyi][j], w[dyi][j + 1], a1[dyi % 3]);
            }
            s0 = fmaf(g0s[dyi], w[dyi][4].x, s0);
            s1 = fmaf(g1s[dyi], w[dyi][0].y, s1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float2v t0 = a0[0] + a0[1] + a0[2], t1 = a1[0] + a1[1] + a1[2];
        const float r0 = (t0.x + t0.y + s0) * inv_nelems, r1 = (t1.x + t1.y + s1) * inv_nelems;
#pragma unroll
        for (int q = 0; q < CC; ++q)
            if (q == i) { res[q][0] = r0; res[q][1] = r1; }
    }
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    if (dbg & 1) return;
#pragma unroll
    for (int i = 0; i < CC; ++i)
        __builtin_amdgcn_raw_buffer_store_b64(
            __builtin_bit_cast(uint2v, float2v{res[i][0], res[i][1]}), rsrc_dst,
            c0 + i < c_end ? dst_voff : kDead,
            __builtin_amdgcn_readfirstlane((c0 + i) * plane * 4), 2 /* nt */);
}
#endif

template <typename K>
__global__ __launch_bounds__(K::THREADS, K::NW == 4 ? 2 : 1) void corr_bwd_d4_dma_kernel(
    const float *__restrict__ x1, const float *__restrict__ x2, const float *__restrict__ gout,
    float *__restrict__ gin1, float *__restrict__ gin2, int C, int H, int W, int tiles_x,
    int tiles_y, int cslice, int nslice, int dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef CERB_ABLATE
    dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CC = K::CC, NB = K::NB;
    constexpr int kDead = static_cast<int>(0x80000000u);
    if (dbg & 64) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = wave * K::RPW + lane / K::TSXP;
    const int sxp = lane % K::TSXP;

    // runtime divisions run on the VALU: pin the (uniform) results to SGPRs, a buffer
    // resource held in VGPRs costs a waterfall loop around every DMA
    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int side = __builtin_amdgcn_readfirstlane(bid & 1); bid >>= 1;  // 0: gradInput1
    const int slice = __builtin_amdgcn_readfirstlane(bid % nslice); bid /= nslice;
    const int tx = __builtin_amdgcn_readfirstlane(bid % tiles_x); bid /= tiles_x;
    const int ty = __builtin_amdgcn_readfirstlane(bid % tiles_y);
    const int b = __builtin_amdgcn_readfirstlane(bid / tiles_y);
    const int x0 = tx * K::TW, y0 = ty * K::TH;
    const int c_begin = slice * cslice;
    const int c_end = min(C, c_begin + cslice);
    const int plane = H * W;
    const int nchunks = (c_end - c_begin + CC - 1) / CC;

    const float *src = (side == 0 ? x2 : x1) + static_cast<int64_t>(b) * C * plane;
    float *dstb = (side == 0 ? gin1 : gin2) + static_cast<int64_t>(b) * C * plane;
    const float *gob = gout + static_cast<int64_t>(b) * (kND * kND) * plane;
    const __amdgpu_buffer_rsrc_t rsrc_src = uniform_rsrc(src, C * plane * 4);
    const __amdgpu_buffer_rsrc_t rsrc_dst = uniform_rsrc(dstb, C * plane * 4);

    const int y = y0 + r, x = x0 + 2 * sxp;
    const bool live = y < H && x < W;
    const int dst_voff = live ? (y * W + x) * 4 : kDead;

    // per-lane byte offsets of this wave's DMA slots (channel 0 of a chunk)
    int voff[K::DW];
#pragma unroll
    for (int q = 0; q < K::DW; ++q) {
        const int id = (wave + K::NW * q) * K::LPI + lane;
        const int pl = id / (K::HR * K::HW4), rem = id % (K::HR * K::HW4);
        const int row = rem / K::HW4, c4 = rem % K::HW4;
        const int gy = y0 - kD + row, gx = x0 - kD + 4 * c4;
        voff[q] = (lane < K::LPI && gy >= 0 && gy < H && gx >= 0 && gx < W)
                      ? (pl * plane + gy * W + gx) * 4 : kDead;
    }
#pragma unroll
    for (int k = 0; k < NB - 1; ++k)   // in flight while the gradOutput registers are gathered
        bwd_dma_issue<K>(smem + k * K::BUF, rsrc_src, voff, wave, lane, c_begin + k * CC, c_end, plane);

    float2v g0p[kND][4], g1p[kND][4];
    float g0s[kND], g1s[kND];
    {
        // branch-free gather as in corr_bwd_d4_kernel, through a buffer resource: the
        // displacement's plane/shift is the SCALAR offset, the lane's pixel the vector offset,
        // and a tap outside the image is an out-of-range vector offset (reads 0) -- one
        // v_cndmask per value instead of a 64-bit pointer select (the gather is ~1300 VALU
        // instructions per wave at 2 waves/SIMD otherwise)
        const __amdgpu_buffer_rsrc_t rsrc_go = uniform_rsrc(gob, kND * kND * plane * 4);
        const int lane_byte = live ? (y * W + x) * 4 : kDead;
        // ONE load site per value for both sides (two sites writing the same registers made
        // hipcc load into temporaries and shuffle them under shallow counted waits): side 0
        // is the same gather with every shift 0 and every tap valid.
        int vx0[kND], vx1[kND];   // per horizontal displacement: byte offset or "outside"
        bool oky[kND];
#pragma unroll
        for (int k = 0; k < kND; ++k) {
            const int yy = side ? y + k - kD : y, xx = side ? x + k - kD : x;
            oky[k] = live && yy >= 0 && yy < H;
            vx0[k] = (xx >= 0 && xx < W) ? lane_byte : kDead;
            vx1[k] = (xx + 1 >= 0 && xx + 1 < W) ? lane_byte + 4 : kDead;
        }
#pragma unroll
        for (int d = 0; d < kND * kND; ++d) {
            const int dyi = d / kND, dxi = d % kND;
            // side 0: gO[d][y][x (+1)];  side 1: gO[80-d][y+dy][x+dx (+1)], whose scalar offset
            // is never negative (80-d >= 9*(8-dyi))
            // (arithmetic on the 0/1 side instead of a select: hipcc turns scalar selects into branches)
            const int soff = ((d + side * (kND * kND - 1 - 2 * d)) * plane +
                              side * ((dyi - kD) * W + (dxi - kD))) * 4;
            const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                 rsrc_go, oky[dyi] ? vx0[dxi] : kDead, soff, 0));
            const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                 rsrc_go, oky[dyi] ? vx1[dxi] : kDead, soff, 0));
            if (dxi == 8) g0s[dyi] = v0; else if (dxi & 1) g0p[dyi][dxi / 2].y = v0; else g0p[dyi][dxi / 2].x = v0;
            if (dxi == 0) g1s[dyi] = v1; else if (dxi & 1) g1p[dyi][(dxi - 1) / 2].x = v1; else g1p[dyi][(dxi - 1) / 2].y = v1;
        }
    }

    if (dbg & 32) {  // ablation: prologue + gather only
        if (live) gin1[y * W + x] = g0s[0] + g1s[8] + g0p[4][2].x;
        return;
    }
    const float inv_nelems = 1.0f / static_cast<float>(C);
    const int woff = r * K::RS + 2 * sxp;
    for (int k = 0; k < nchunks; ++k) {
        // own DMAs of chunk k have landed: at most the NB-2 younger chunks' loads may be pending
        wait_vmcnt<K::WAITN>();
        __builtin_amdgcn_s_barrier();   // chunk k complete in LDS; everyone is done with chunk k-1
        const int c0 = c_begin + k * CC;
        bwd_dma_step<K>(smem + (k % NB) * K::BUF, smem + ((k + NB - 1) % NB) * K::BUF, rsrc_src,
                        rsrc_dst, voff, wave, lane, c0, c0 + (NB - 1) * CC, c_end, plane, g0p, g1p,
                        g0s, g1s, woff, dst_voff, inv_nelems, dbg);
        // LDS reads of chunk k have returned (the FMAs consumed them) before the next barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#endif
}

// ============================================================================
// backward, displacement-group variant
// ============================================================================
// Same arithmetic as corr_bwd_d4_kernel, different ownership: a workgroup is 3 vertical
// displacement groups (dy in {-4..-2}, {-1..1}, {2..4}) x NSW spatial wavefronts.  A lane
// keeps only the 27 gradOutput values of its group (54 registers instead of 162), so the
// kernel fits ~100 VGPRs instead of ~250: 2-3x the wavefronts in flight, a 3x shorter
// gather per lane, and the three partial sums of a pixel meet in LDS once per channel chunk.
template <int TSXP_, int CC_, int RS_, int NSW_>
struct BwdG3Cfg {
    static constexpr int TSXP = TSXP_, CC = CC_, RS = RS_, NSW = NSW_;
    static constexpr int TW = 2 * TSXP;
    static constexpr int RPW = 64 / TSXP;
    static constexpr int TH = NSW * RPW;
    static constexpr int HR = TH + 2 * kD;
    static constexpr int HW4 = (TW + 2 * kD) / 4;
    static constexpr int PS = HR * RS;
    static constexpr int THREADS = 3 * NSW * 64;
    static constexpr int N = CC * HR * HW4;
    static constexpr int NSLOT = (N + THREADS - 1) / THREADS;
    static constexpr int BUF = CC * PS;                    // floats per window buffer
    static constexpr int PART = CC * 3 * NSW * 64 * 2;     // floats per partial-sum buffer
    static constexpr size_t LDS_BYTES = sizeof(float) * (2 * BUF + 2 * PART);
};

template <typename K, typename T, bool VEC>
__global__ __launch_bounds__(K::THREADS, 3) void corr_bwd_d4_g3_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, const T *__restrict__ gout,
    T *__restrict__ gin1, T *__restrict__ gin2, int C, int H, int W, int tiles_x,
    int tiles_y, int cslice, int nslice) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CC = K::CC, NSW = K::NSW;
    float *part_base = smem + 2 * K::BUF;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int grp = wave / NSW;          // displacement group: dyi in [3*grp, 3*grp+2]
    const int sw = wave % NSW;           // spatial wavefront
    const int r = sw * K::RPW + lane / K::TSXP;
    const int sxp = lane % K::TSXP;
    const int slot_lane = sw * 64 + lane;  // index of this lane's pixel pair inside the tile

    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int side = bid & 1; bid >>= 1;
    const int slice = bid % nslice; bid /= nslice;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx * K::TW, y0 = ty * K::TH;
    const int c_begin = slice * cslice;
    const int c_end = min(C, c_begin + cslice);
    const int plane = H * W;

    const T *src = (side == 0 ? x2 : x1) + static_cast<int64_t>(b) * C * plane;
    T *dstb = (side == 0 ? gin1 : gin2) + static_cast<int64_t>(b) * C * plane;
    const T *gob = gout + static_cast<int64_t>(b) * (kND * kND) * plane;

    const int y = y0 + r, x = x0 + 2 * sxp;
    const bool live = y < H && x < W;

    // ---- staging descriptors (whole workgroup stages the window) ----
    int goff[K::NSLOT], loff[K::NSLOT], gx0[K::NSLOT], chi[K::NSLOT];
#pragma unroll
    for (int j = 0; j < K::NSLOT; ++j) {
        const int id = tid + j * K::THREADS;
        goff[j] = -1; loff[j] = -1; gx0[j] = 0; chi[j] = 0;
        if (id < K::N) {
            const int pl = id / (K::HR * K::HW4);
            const int rem = id % (K::HR * K::HW4);
            const int row = rem / K::HW4, c4 = rem % K::HW4;
            const int gy = y0 - kD + row, gx = x0 - kD + 4 * c4;
            loff[j] = pl * K::PS + row * K::RS + 4 * c4;
            gx0[j] = gx; chi[j] = pl;
            const bool in = gy >= 0 && gy < H && (VEC ? (gx >= 0 && gx < W) : (gx > -4 && gx < W));
            if (in) goff[j] = pl * plane + gy * W + gx + 4;
        }
    }
    float4 stage[K::NSLOT];
    auto prefetch = [&](int c_first) {
#pragma unroll
        for (int j = 0; j < K::NSLOT; ++j) {
            const bool on = goff[j] >= 0 && c_first + chi[j] < c_end;
            if (VEC) {
                const T *p = on ? src + static_cast<int64_t>(c_first) * plane + (goff[j] - 4)
                                : reinterpret_cast<const T *>(g_zero16);
                stage[j] = Gmem<T>::load4(p);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (on) {
                    const T *p = src + static_cast<int64_t>(c_first) * plane + (goff[j] - 4);
                    const int gx = gx0[j];
                    if (gx >= 0 && gx < W) v.x = Gmem<T>::load1(p);
                    if (gx + 1 >= 0 && gx + 1 < W) v.y = Gmem<T>::load1(p + 1);
                    if (gx + 2 >= 0 && gx + 2 < W) v.z = Gmem<T>::load1(p + 2);
                    if (gx + 3 >= 0 && gx + 3 < W) v.w = Gmem<T>::load1(p + 3);
                }
                stage[j] = v;
            }
        }
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int j = 0; j < K::NSLOT; ++j)
            if (loff[j] >= 0) st4(buf + loff[j], stage[j]);
    };

    prefetch(c_begin);

    // ---- this group's 27 gradOutput values per pixel (branch-free, see corr_bwd_d4_kernel) ----
    float2v g0p[3][4], g1p[3][4];
    float g0s[3], g1s[3];
    {
        const int lane_off = live ? y * W + x : 0;
        unsigned ymask = 0, xmask0 = 0, xmask1 = 0;
#pragma unroll
        for (int k = 0; k < kND; ++k) {
            const int yy = y + k - kD, xx = x + k - kD;
            if (yy >= 0 && yy < H) ymask |= 1u << k;
            if (xx >= 0 && xx < W) xmask0 |= 1u << k;
            if (xx + 1 >= 0 && xx + 1 < W) xmask1 |= 1u << k;
        }
        if (!live) ymask = 0;
        const bool pair_ok = live && (VEC || x + 1 < W);
        const T *zero = reinterpret_cast<const T *>(g_zero16);
#pragma unroll
        for (int dl = 0; dl < 3; ++dl) {
            const int dyi = 3 * grp + dl;  // wave-uniform
#pragma unroll
            for (int dxi = 0; dxi < kND; ++dxi) {
                const int d = dyi * kND + dxi;
                float v0, v1;
                if (side == 0) {
                    const T *pd = gob + static_cast<int64_t>(d) * plane;
                    if (VEC) {
                        const float2 t = Gmem<T>::load2(pd + lane_off);
                        v0 = t.x; v1 = t.y;
                    } else {
                        v0 = Gmem<T>::load1(pd + lane_off);
                        v1 = Gmem<T>::load1(pd + (pair_ok ? lane_off + 1 : lane_off));
                    }
                } else {
                    const int uni = (kND * kND - 1 - d) * plane + (dyi - kD) * W + (dxi - kD);
                    const bool oky = (ymask >> dyi) & 1u;
                    const bool ok0 = oky && ((xmask0 >> dxi) & 1u);
                    const bool ok1 = oky && ((xmask1 >> dxi) & 1u);
                    v0 = Gmem<T>::load1(ok0 ? gob + (uni + lane_off) : zero);
                    v1 = Gmem<T>::load1(ok1 ? gob + (uni + lane_off + 1) : zero);
                }
                if (dxi == 8) g0s[dl] = v0; else if (dxi & 1) g0p[dl][dxi / 2].y = v0; else g0p[dl][dxi / 2].x = v0;
                if (dxi == 0) g1s[dl] = v1; else if (dxi & 1) g1p[dl][(dxi - 1) / 2].x = v1; else g1p[dl][(dxi - 1) / 2].y = v1;
            }
        }
    }

    const float inv_nelems = 1.0f / static_cast<float>(C);
    int it = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += CC, ++it) {
        float *buf = smem + (it & 1) * K::BUF;
        float *part = part_base + (it & 1) * K::PART;
        commit(buf);
        __syncthreads();
        prefetch(c0 + CC);
        const float *wbase = buf + (r + 3 * grp) * K::RS + 2 * sxp;
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            const float *wp = wbase + i * K::PS;
            float2v a0 = float2v{0.f, 0.f}, a1 = float2v{0.f, 0.f};
            float s0 = 0.f, s1 = 0.f;
            float2v w[3][5];
#pragma unroll
            for (int q = 0; q < 5; ++q) w[0][q] = ld2v_nomerge(wp + 2 * q);
#pragma unroll
            for (int dl = 0; dl < 3; ++dl) {
                if (dl + 1 < 3) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) w[dl + 1][q] = ld2v_nomerge(wp + (dl + 1) * K::RS + 2 * q);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (dl + 1 < 3) __builtin_amdgcn_s_waitcnt(0xC57F);
                else __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a0 = pkfma(g0p[dl][j], w[dl][j], a0);
                    a1 = pkfma(g1p[dl][j], w[dl][j + 1], a1);
                }
                s0 = fmaf(g0s[dl], w[dl][4].x, s0);
                s1 = fmaf(g1s[dl], w[dl][0].y, s1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // partial sums of this displacement group -> LDS
            *reinterpret_cast<float2v *>(part + ((i * 3 + grp) * (NSW * 64) + slot_lane) * 2) =
                float2v{a0.x + a0.y + s0, a1.x + a1.y + s1};
        }
        __syncthreads();
        // group g finishes channels g, g+3, ... of the chunk: add the three partials, store
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            if (i % 3 != grp) continue;   // wave-uniform
            const float *pp = part + (i * 3 * (NSW * 64) + slot_lane) * 2;
            const float2v p0 = *reinterpret_cast<const float2v *>(pp);
            const float2v p1 = *reinterpret_cast<const float2v *>(pp + NSW * 64 * 2);
            const float2v p2 = *reinterpret_cast<const float2v *>(pp + 2 * NSW * 64 * 2);
            const float r0 = (p0.x + p1.x + p2.x) * inv_nelems, r1 = (p0.y + p1.y + p2.y) * inv_nelems;
            if (live && c0 + i < c_end) {
                T *dst = dstb + static_cast<int64_t>(c0 + i) * plane + y * W + x;
                if (VEC) {
                    Gmem<T>::stream2(dst, r0, r1);
                } else {
                    Gmem<T>::store1(dst, r0);
                    if (x + 1 < W) Gmem<T>::store1(dst + 1, r1);
                }
            }
        }
    }
}

// ============================================================================
// backward, displacement-row streaming variant (fp32, vector path)
// ============================================================================
// The kernels above keep a pixel's 81 gradOutput values in registers (162 VGPRs for two
// pixels: 2 waves/SIMD) and stream CHANNELS through LDS; every channel slice gathers those 81
// values again, and the gather (12 of 38 us at the 32x128x256 level, the same ~85 MB at every
// level) cannot overlap the FMA phase of its own workgroup.  This kernel turns the loops
// inside out -- the dual of the forward: the ACCUMULATORS (4 channels x a 4-pixel strip per
// lane) stay in registers and the nine vertical displacements are streamed, for all the
// channels of the workgroup at once:
//
//   gI1[c][y][x] = 1/C sum_dy sum_dx gO[dy,dx][y][x]       * x2[c][y+dy][x+dx]
//   gI2[c][y][x] = 1/C sum_ey sum_ex gO[-ey,-ex][y+ey][x+ex] * x1[c][y+ey][x+ex]
//
//   step s (dy or ey = s-4), lane = (tile row r, strip sx), wave = 4 channels:
//     g[j][0..3]  the 9 gradOutput planes of this step at the lane's strip (side 2: the
//                 plane (-ey,-ex) at the strip shifted by (ey,ex))           <- LDS, 9-15 b128
//     w[0..11]    window row y+dy of one channel, columns x-4 .. x+7          <- LDS, 3 b128
//     acc[ch][px] += g[j][px] * w[px+j]                                        36 FMAs / channel
//
// so gradOutput is read ONCE per (tile, side) whatever the channel count, nothing is
// gathered into registers up front, a lane needs ~90 VGPRs (4-5 waves/SIMD), and both LDS
// images arrive by LDS-DMA a step (gradOutput) or two (window rows) ahead of their use:
//   * window: a ring of TH+2 image rows x all channels of the workgroup ([row][channel][72
//     floats]; rows are 32*72 floats apart = 0 mod 64 banks, which is exactly what the
//     ds_read_b128 lane groups want); each step retires one row and lands one.
//   * gradOutput: two buffers of [9 planes][TH rows][72 floats] (aligned 16-byte slots with
//     a 4-pixel halo, out-of-image slots read zeros through the buffer resource: exact, also
//     for NaN / Inf neighbours); side 2's shift by ex is a compile-time register selection
//     inside the 12 floats a lane reads.
// Every wave issues its share of the DMA instructions of a step before it computes, waits
// with a counted vmcnt for everything but the row it just requested, and the workgroup meets
// at one barrier per step.
template <int CB_, int NG_>
struct BwdRowsCfg {
    static constexpr int CB = CB_, NG = NG_, CR = CB * NG;     // channels per wave / waves / channels per workgroup
    static constexpr int TH = 4, TSX = 16, TW = TSX * kP;       // 4 x 64 tile: one wave of strips
    static constexpr int NRING = TH + 2;
    static constexpr int RSF = 72;                              // floats per staged row (64 + 2*4 halo)
    static constexpr int ROWF = CR * RSF;                       // floats per ring row (all channels)
    static constexpr int GBUF = kND * TH * RSF;                 // floats per gradOutput buffer
    static constexpr int NGB = 2;
    static constexpr int THREADS = 64 * NG;
    static constexpr int ROW_SLOTS = CR * (RSF / 4), G_SLOTS = kND * TH * (RSF / 4);
    static constexpr int ROW_INSTR = (ROW_SLOTS + 63) / 64, G_INSTR = (G_SLOTS + 63) / 64;
    static constexpr int ROW_PW = (ROW_INSTR + NG - 1) / NG, G_PW = (G_INSTR + NG - 1) / NG;  // per wave
    static constexpr size_t LDS_BYTES = sizeof(float) * (NRING * ROWF + NGB * GBUF);
    // waves per SIMD the register allocator must leave room for: as many workgroups as LDS admits
    static constexpr int WG_PER_CU = static_cast<int>((160 * 1024) / LDS_BYTES);
    static constexpr int WPS = (WG_PER_CU * NG + 3) / 4 > 4 ? 4 : (WG_PER_CU * NG + 3) / 4;
    static_assert(ROWF % 64 == 0, "ring rows must be a multiple of 64 banks apart");
    static_assert(ROW_PW <= 4, "the counted wait handles up to four row DMA instructions per wave");
};

// one ds_read_b128, exactly as written: volatile keeps hipcc from splitting a 16-byte LDS read
// whose elements are only partly used into ds_read_b32 / ds_read2 pieces (seen in the first
// build of the kernel below: 96 ds_read_b32 + 56 ds_read2 instead of 48 ds_read_b128)
typedef float f4v_lds __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) f4v_lds *lds_f4_volatile_ptr;
__device__ __forceinline__ float4 ld4_lds(const float *p) {
    const f4v_lds v = *(lds_f4_volatile_ptr)(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

#if defined(__HIP_DEVICE_COMPILE__)
// DMA of one window row (image row gy of every channel of the workgroup) into `wr`
template <typename K>
__device__ __forceinline__ void rows_issue_row(float *__restrict__ wr, __amdgpu_buffer_rsrc_t rsrc,
                                               const int (&voff)[K::ROW_PW], int wave, int gy,
                                               int H, int W, int c_begin, int plane, bool alive) {
    constexpr int kDead = static_cast<int>(0x80000000u);
    constexpr int kNoSlot = static_cast<int>(0x80000001u);   // lane past the last slot: masked off
    const bool ok = alive && gy >= 0 && gy < H;                          // wave-uniform
    const int soff = __builtin_amdgcn_readfirstlane(ok ? (c_begin * plane + gy * W) * 4 : 0);
#pragma unroll
    for (int q = 0; q < K::ROW_PW; ++q) {
        const int inst = wave + K::NG * q;
        if (inst < K::ROW_INSTR && (K::ROW_SLOTS % 64 == 0 || voff[q] != kNoSlot))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(wr + inst * 256), 16,
                                                     ok ? voff[q] : kDead, soff, 0, 0);
    }
}

// DMA of the nine gradOutput planes of step s into `wr` ([plane][row][72 floats])
template <typename K, int SIDE>
__device__ __forceinline__ void rows_issue_g(float *__restrict__ wr, __amdgpu_buffer_rsrc_t rsrc,
                                             int wave, int lane, int s, int x0, int y0, int H,
                                             int W, int plane, bool alive) {
    constexpr int kDead = static_cast<int>(0x80000000u);
#pragma unroll
    for (int q = 0; q < K::G_PW; ++q) {
        const int inst = wave + K::NG * q;
        if (inst >= K::G_INSTR) continue;
        const int i = inst * 64 + lane;                 // slot index in [plane][row][18 slots]
        const int j = i / (K::TH * 18), rem = i % (K::TH * 18);
        const int r = rem / 18, sl = rem % 18;
        // side 0: plane (s, j) at (y0+r, x0-4+4sl), the halo slots are never read;
        // side 1: plane (-ey,-ex) = (8-s, 8-j) at the row shifted by ey = s-4; the shift by
        //         ex = j-4 is applied when the lane picks its 4 floats out of the 12 it reads
        const int pl = SIDE ? (kND - 1 - s) * kND + (kND - 1 - j) : s * kND + j;
        const int gy = y0 + r + (SIDE ? s - kD : 0);
        const int gx = x0 - kD + 4 * sl;
        const bool ok = alive && gy >= 0 && gy < H && gx >= 0 && gx < W &&
                        (SIDE || (sl >= 1 && sl <= 16));
        // lanes past the last slot are masked off: an out-of-range lane still WRITES its zeros,
        // and the last instruction would run 896 bytes into the other buffer
        if (i < K::G_SLOTS)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(wr + inst * 256), 16,
                                                     ok ? (pl * plane + gy * W + gx) * 4 : kDead, 0, 0, 0);
    }
}

// the arithmetic of one step: 9 gradOutput planes x CB channels of the lane's window row
template <typename K, int SIDE>
__device__ __forceinline__ void rows_compute(const float *__restrict__ ring, const float *__restrict__ gbuf,
                                             int rowoff0, int goff, float (&acc)[K::CB][kP], int dbg) {
    const int rowoff[1] = {rowoff0};
    // ---- gradOutput of this step: g[j][px] ----
    // (sched_barriers pin the order "reads of one block, then its FMAs": left alone hipcc
    // hoists every LDS read of the step to the top and spills 112 VGPRs)
    float g[kND][kP];
    const float *gp = gbuf + goff;
#pragma unroll
    for (int j = 0; j < kND; ++j) {
        const float *pj = gp + j * (K::TH * K::RSF);
        if (SIDE == 0) {
            const float4 q = ld4_lds(pj + 4);
            g[j][0] = q.x; g[j][1] = q.y; g[j][2] = q.z; g[j][3] = q.w;
        } else {
            // floats j .. j+3 of the 12-float span: one or two aligned quads
            const int q0 = j / 4;
            const float4 a = ld4_lds(pj + 4 * q0);
            float sp[8] = {a.x, a.y, a.z, a.w, 0.f, 0.f, 0.f, 0.f};
            if (j % 4) {
                const float4 b = ld4_lds(pj + 4 * q0 + 4);
                sp[4] = b.x; sp[5] = b.y; sp[6] = b.z; sp[7] = b.w;
            }
#pragma unroll
            for (int p = 0; p < kP; ++p) g[j][p] = sp[j % 4 + p];
        }
    }
    // ---- window rows, CB channels, 36 FMAs each; channel i+1's reads ride under channel i's FMAs ----
    const float *wp = ring + rowoff[0];
    float4 w0 = ld4_lds(wp), w1 = ld4_lds(wp + 4), w2 = ld4_lds(wp + 8);
#pragma unroll
    for (int i = 0; i < K::CB; ++i) {
        float4 n0 = w0, n1 = w1, n2 = w2;
        if (i + 1 < K::CB) {
            n0 = ld4_lds(wp + (i + 1) * K::RSF);
            n1 = ld4_lds(wp + (i + 1) * K::RSF + 4);
            n2 = ld4_lds(wp + (i + 1) * K::RSF + 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        // channel i's three reads have returned once at most channel i+1's three are pending
        if (i + 1 < K::CB) __builtin_amdgcn_s_waitcnt(0xC37F);   // lgkmcnt(3)
        else __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        const float w[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
        for (int jj = 0; jj < kND; ++jj)
#pragma unroll
            for (int p = 0; p < kP; ++p)
                if (!(dbg & 4) || jj == 0) acc[i][p] = fmaf(g[jj][p], w[p + jj], acc[i][p]);
        __builtin_amdgcn_sched_barrier(0);
        w0 = n0; w1 = n1; w2 = n2;
    }
}

// one step: request the next gradOutput planes and the row after next, then consume step s
template <typename K, int SIDE>
__device__ __forceinline__ void rows_step(
    const float *__restrict__ ring, const float *__restrict__ gbuf, float *__restrict__ row_wr,
    float *__restrict__ g_wr, __amdgpu_buffer_rsrc_t rsrc_src, __amdgpu_buffer_rsrc_t rsrc_go,
    const int (&voff)[K::ROW_PW], int wave, int lane, int s, int x0, int y0, int H, int W,
    int c_begin, int plane, const int (&rowoff)[K::TH > 0 ? 1 : 1], int goff,
    float (&acc)[K::CB][kP], int dbg) {
    rows_issue_g<K, SIDE>(g_wr, rsrc_go, wave, lane, s + 1, x0, y0, H, W, plane, s + 1 < kND && !(dbg & 1));
    // the counted wait at the end of the step relies on THIS order (gradOutput, then the row):
    // the two DMA groups write disjoint restrict regions, so nothing else stops hipcc from
    // swapping them (it did, in the copy of the loop body it made for odd steps)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    rows_issue_row<K>(row_wr, rsrc_src, voff, wave, y0 - kD + s + K::TH + 1, H, W, c_begin, plane,
                      s + 2 < kND && !(dbg & 2));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    rows_compute<K, SIDE>(ring, gbuf, rowoff[0], goff, acc, dbg);
}
#endif

template <typename K>
__global__ __launch_bounds__(K::THREADS, K::WPS) void corr_bwd_d4_rows_kernel(
    const float *__restrict__ x1, const float *__restrict__ x2, const float *__restrict__ gout,
    float *__restrict__ gin1, float *__restrict__ gin2, int C, int H, int W, int tiles_x,
    int tiles_y, int nrange, int dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef CERB_ABLATE
    dbg = 0;   // timing ablations (1: no gradOutput DMA, 2: no row DMA, 4: no FMAs) exist in -DCERB_ABLATE builds only
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kDead = static_cast<int>(0x80000000u);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane / K::TSX, sx = lane % K::TSX;

    // (tile, channel range, side) with the side fastest: the two workgroups that read the
    // same gradOutput tile are neighbours in the XCD-contiguous order
    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int side = __builtin_amdgcn_readfirstlane(bid & 1); bid >>= 1;
    const int range = __builtin_amdgcn_readfirstlane(bid % nrange); bid /= nrange;
    const int tx = __builtin_amdgcn_readfirstlane(bid % tiles_x); bid /= tiles_x;
    const int ty = __builtin_amdgcn_readfirstlane(bid % tiles_y);
    const int b = __builtin_amdgcn_readfirstlane(bid / tiles_y);
    const int x0 = tx * K::TW, y0 = ty * K::TH;
    const int c_begin = range * K::CR, c_end = min(C, c_begin + K::CR);
    const int plane = H * W;

    const float *src = (side == 0 ? x2 : x1) + static_cast<int64_t>(b) * C * plane;
    float *dstb = (side == 0 ? gin1 : gin2) + static_cast<int64_t>(b) * C * plane;
    const __amdgpu_buffer_rsrc_t rsrc_src = uniform_rsrc(src, C * plane * 4);
    const __amdgpu_buffer_rsrc_t rsrc_go =
        uniform_rsrc(gout + static_cast<int64_t>(b) * (kND * kND) * plane, kND * kND * plane * 4);

    float *ring = smem;
    float *gbufs = smem + K::NRING * K::ROWF;

    // this wave's window-row DMA slots: slot i of [channel][18 slots] -> per-lane byte offset
    // relative to (first channel of the range, start of the image row)
    int voff[K::ROW_PW];
#pragma unroll
    for (int q = 0; q < K::ROW_PW; ++q) {
        constexpr int kNoSlot = static_cast<int>(0x80000001u);
        const int i = (wave + K::NG * q) * 64 + lane;
        const int ch = i / 18, sl = i % 18;
        const int gx = x0 - kD + 4 * sl;
        voff[q] = i >= K::ROW_SLOTS ? kNoSlot
                  : (c_begin + ch < c_end && gx >= 0 && gx < W) ? (ch * plane + gx) * 4 : kDead;
    }

    // ---- prologue: the TH+1 rows of steps 0 and 1, gradOutput of step 0 ----
#pragma unroll
    for (int k = 0; k <= K::TH; ++k)
        rows_issue_row<K>(ring + k * K::ROWF, rsrc_src, voff, wave, y0 - kD + k, H, W, c_begin, plane, true);
    if (side == 0) rows_issue_g<K, 0>(gbufs, rsrc_go, wave, lane, 0, x0, y0, H, W, plane, true);
    else rows_issue_g<K, 1>(gbufs, rsrc_go, wave, lane, 0, x0, y0, H, W, plane, true);

    float acc[K::CB][kP];
#pragma unroll
    for (int i = 0; i < K::CB; ++i)
#pragma unroll
        for (int p = 0; p < kP; ++p) acc[i][p] = 0.f;

    const int goff = r * K::RSF + 4 * sx;                       // lane's span inside a gradOutput plane
    const int choff = wave * K::CB * K::RSF + 4 * sx;           // lane's span inside a ring row
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    for (int s = 0; s < kND; ++s) {
        // ring slot of the lane's window row (image row y0-4+s+r) and of the row requested now
        int slot = s + r;
        if (slot >= K::NRING) slot -= K::NRING;
        if (slot >= K::NRING) slot -= K::NRING;
        const int rowoff[1] = {slot * K::ROWF + choff};
        const int wslot = (s + K::TH + 1) % K::NRING;
        const float *g_rd = gbufs + (s & 1) * K::GBUF;
        float *g_wr = gbufs + ((s + 1) & 1) * K::GBUF;
        if (side == 0)
            rows_step<K, 0>(ring, g_rd, ring + wslot * K::ROWF, g_wr, rsrc_src, rsrc_go, voff, wave, lane,
                            s, x0, y0, H, W, c_begin, plane, rowoff, goff, acc, dbg);
        else
            rows_step<K, 1>(ring, g_rd, ring + wslot * K::ROWF, g_wr, rsrc_src, rsrc_go, voff, wave, lane,
                            s, x0, y0, H, W, c_begin, plane, rowoff, goff, acc, dbg);
        // the next step's gradOutput has landed once only this step's row request(s), issued
        // after it, may still be in flight; every LDS read of this step has returned
        {
            int nrow = 0;   // row DMA instructions this wave issued in this step (wave-uniform)
#pragma unroll
            for (int q = 0; q < K::ROW_PW; ++q) nrow += (wave + K::NG * q < K::ROW_INSTR) ? 1 : 0;
            if (nrow == 4) wait_vmcnt<4>();
            else if (nrow == 3) wait_vmcnt<3>();
            else if (nrow == 2) wait_vmcnt<2>();
            else if (nrow == 1) wait_vmcnt<1>();
            else wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: 1/C, coalesced 16-byte stores ----
    const float inv_nelems = 1.0f / static_cast<float>(C);
    const int y = y0 + r, x = x0 + 4 * sx;
    if (y < H && x < W) {
#pragma unroll
        for (int i = 0; i < K::CB; ++i) {
            const int c = c_begin + wave * K::CB + i;
            if (c < c_end) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(
                    f4v{acc[i][0] * inv_nelems, acc[i][1] * inv_nelems, acc[i][2] * inv_nelems,
                        acc[i][3] * inv_nelems},
                    reinterpret_cast<f4v *>(dstb + static_cast<int64_t>(c) * plane + y * W + x));
            }
        }
    }
#endif
}

// ---- host side -------------------------------------------------------------
template <typename K, typename T>
int launch_bwd(const char *name, const void *in1, const void *in2, const void *goutp, void *g1p,
               void *g2p, const CorrGeom &g, bool vec, hipStream_t s) {
    const T *x1 = static_cast<const T *>(in1), *x2 = static_cast<const T *>(in2);
    const T *gout = static_cast<const T *>(goutp);
    T *gin1 = static_cast<T *>(g1p), *gin2 = static_cast<T *>(g2p);
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int64_t tiles = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    if (tiles > 0x7fffffff) return CERB_ETOOLARGE;
    // channel slice per workgroup: every slice repeats the 81-value gradOutput gather, so
    // slice only as far as needed to put ~2 workgroups on each of the 256 CUs (sweep on
    // MI355X: 32 / 16 / 8 / 8 channels for the four W32 levels at batch 4)
    int cslice = g.C;
    while (cslice > 8 && tiles * 2 * ((g.C + cslice - 1) / cslice) < 512) cslice /= 2;
    if (const int forced = option(OPT_CORR_BWD_CSLICE)) cslice = forced;
    cslice = ((cslice + K::CC - 1) / K::CC) * K::CC;
    const int nslice = (g.C + cslice - 1) / cslice;
    if (tiles * nslice * 2 > 0x7fffffff) return CERB_ETOOLARGE;
    const dim3 grid(static_cast<unsigned>(tiles * nslice * 2));
    int rc;
    static std::atomic<uint64_t> lds_v{0}, lds_s{0};
    const int dbg = debug_mask();
    if (vec) {
        note_kernel(1, name);
        if ((rc = ensure_lds(corr_bwd_d4_kernel<K, T, true>, K::LDS_BYTES, &lds_v))) return rc;
        hipLaunchKernelGGL((corr_bwd_d4_kernel<K, T, true>), grid, dim3(K::THREADS), K::LDS_BYTES,
                           s, x1, x2, gout, gin1, gin2, g.C, g.H, g.W, tiles_x, tiles_y, cslice,
                           nslice, dbg);
    } else if constexpr (sizeof(T) == 4) {
        note_kernel(1, name);
        if ((rc = ensure_lds(corr_bwd_d4_kernel<K, T, false>, K::LDS_BYTES, &lds_s))) return rc;
        hipLaunchKernelGGL((corr_bwd_d4_kernel<K, T, false>), grid, dim3(K::THREADS), K::LDS_BYTES,
                           s, x1, x2, gout, gin1, gin2, g.C, g.H, g.W, tiles_x, tiles_y, cslice,
                           nslice, dbg);
    } else {
        return CERB_EUNSUPPORTED;
    }
    return launch_status();
}

template <typename K>
int launch_bwd_dma(const char *name, const void *in1, const void *in2, const void *goutp, void *g1p,
                   void *g2p, const CorrGeom &g, hipStream_t s) {
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int64_t tiles = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    int cslice = option(OPT_CORR_BWD_CSLICE);
    if (cslice <= 0) {
        // enough workgroups for every CU: two 4-wave workgroups or one 8-wave workgroup each
        const int64_t want = K::NW == 4 ? 512 : 256;
        cslice = g.C;
        while (cslice > 8 && 2 * tiles * ((g.C + cslice - 1) / cslice) < want) cslice = (cslice + 1) / 2;
    }
    cslice = std::max(K::CC, (cslice + K::CC - 1) / K::CC * K::CC);
    const int nslice = (g.C + cslice - 1) / cslice;
    const int64_t blocks = tiles * nslice * 2;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    static std::atomic<uint64_t> lds_done{0};
    int rc;
    if ((rc = ensure_lds(corr_bwd_d4_dma_kernel<K>, K::LDS_BYTES, &lds_done))) return rc;
    note_kernel(1, name);
    hipLaunchKernelGGL((corr_bwd_d4_dma_kernel<K>), dim3(static_cast<unsigned>(blocks)),
                       dim3(K::THREADS), K::LDS_BYTES, s, static_cast<const float *>(in1),
                       static_cast<const float *>(in2), static_cast<const float *>(goutp),
                       static_cast<float *>(g1p), static_cast<float *>(g2p), g.C, g.H, g.W, tiles_x,
                       tiles_y, cslice, nslice, debug_mask());
    return launch_status();
}

template <typename K, typename T>
int launch_bwd_g3(const char *name, const void *in1, const void *in2, const void *goutp, void *g1p,
                  void *g2p, const CorrGeom &g, bool vec, hipStream_t s) {
    const T *x1 = static_cast<const T *>(in1), *x2 = static_cast<const T *>(in2);
    const T *gout = static_cast<const T *>(goutp);
    T *gin1 = static_cast<T *>(g1p), *gin2 = static_cast<T *>(g2p);
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int64_t tiles = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    int cslice = g.C;
    while (cslice > 16 && tiles * 2 * ((g.C + cslice - 1) / cslice) < 256) cslice /= 2;
    if (const int forced = option(OPT_CORR_BWD_CSLICE)) cslice = forced;
    cslice = ((cslice + K::CC - 1) / K::CC) * K::CC;
    const int nslice = (g.C + cslice - 1) / cslice;
    if (tiles * nslice * 2 > 0x7fffffff) return CERB_ETOOLARGE;
    const dim3 grid(static_cast<unsigned>(tiles * nslice * 2));
    int rc;
    static std::atomic<uint64_t> lds_v{0}, lds_s{0};
    if (vec) {
        note_kernel(1, name);
        if ((rc = ensure_lds(corr_bwd_d4_g3_kernel<K, T, true>, K::LDS_BYTES, &lds_v))) return rc;
        hipLaunchKernelGGL((corr_bwd_d4_g3_kernel<K, T, true>), grid, dim3(K::THREADS),
                           K::LDS_BYTES, s, x1, x2, gout, gin1, gin2, g.C, g.H, g.W, tiles_x,
                           tiles_y, cslice, nslice);
    } else if constexpr (sizeof(T) == 4) {
        note_kernel(1, name);
        if ((rc = ensure_lds(corr_bwd_d4_g3_kernel<K, T, false>, K::LDS_BYTES, &lds_s))) return rc;
        hipLaunchKernelGGL((corr_bwd_d4_g3_kernel<K, T, false>), grid, dim3(K::THREADS),
                           K::LDS_BYTES, s, x1, x2, gout, gin1, gin2, g.C, g.H, g.W, tiles_x,
                           tiles_y, cslice, nslice);
    } else {
        return CERB_EUNSUPPORTED;
    }
    return launch_status();
}

using BwdG3Wide = BwdG3Cfg<32, 4, 72, 2>;    // 4x64 tile, 6 wavefronts
using BwdG3Wide4 = BwdG3Cfg<32, 2, 72, 4>;   // 8x64 tile, 12 wavefronts
using BwdDma2x5 = BwdDmaCfg<2, 5>;   // 8x64 tile, 2-channel chunks, ring of 5
using BwdDmaNarrow = BwdDmaCfg<2, 5, 4, 60, 16>;   // 16x32 tile (rows 40 floats apart: partly 2-way LDS conflicts)
using BwdWide = BwdCfg<32, 2, 72>;    // 8x64 tile
// Tried and rejected on MI355X (level 3 / level 2, 4 pairs): 16x64 tiles with 8 wavefronts
// (55.8 / 39.0 us vs 45.6 / 29.2: fewer workgroups in flight outweighs the smaller halo) and
// 4x64 tiles with 2 wavefronts (spills; 83 / 50 us).
using BwdNarrow = BwdCfg<16, 2, 96>;  // 16x32 tile

using BwdRows = BwdRowsCfg<4, 8>;     // 4x64 tile, 8 waves x 4 channels, 2 workgroups per CU
using BwdRows44 = BwdRowsCfg<4, 4>;   // 4 waves x 4 channels = 16 channels per workgroup, 3 per CU
using BwdRows84 = BwdRowsCfg<8, 4>;   // 4 waves x 8 channels
using BwdRows82 = BwdRowsCfg<8, 2>;   // 2 waves x 8 channels = 16 channels per workgroup

template <typename K>
int launch_bwd_rows(const char *name, const void *in1, const void *in2, const void *goutp, void *g1p,
                    void *g2p, const CorrGeom &g, hipStream_t s) {
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int64_t tiles = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    const int nrange = (g.C + K::CR - 1) / K::CR;
    const int64_t blocks = tiles * nrange * 2;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    static std::atomic<uint64_t> lds_done{0};
    int rc;
    if ((rc = ensure_lds(corr_bwd_d4_rows_kernel<K>, K::LDS_BYTES, &lds_done))) return rc;
    note_kernel(1, name);
    hipLaunchKernelGGL((corr_bwd_d4_rows_kernel<K>), dim3(static_cast<unsigned>(blocks)),
                       dim3(K::THREADS), K::LDS_BYTES, s, static_cast<const float *>(in1),
                       static_cast<const float *>(in2), static_cast<const float *>(goutp),
                       static_cast<float *>(g1p), static_cast<float *>(g2p), g.C, g.H, g.W, tiles_x,
                       tiles_y, nrange, debug_mask());
    return launch_status();
}

#ifdef CERB_EXPERIMENTS
// the column-walking backward (measured and rejected): compiled into -DCERB_EXPERIMENTS test builds only
#include "corr_d4_experiments.inc"
#endif


template <typename T>
int bwd_dispatch(const void *x1, const void *x2, const void *go, void *g1, void *g2,
                 const CorrGeom &g, bool vec, bool half, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        // 16-bit storage: the matrix-core kernel (corr_mfma.hip; 11: its row-per-wave form of rounds 2-4); variants 1-3 keep the VALU kernels
        const int v = option(OPT_CORR_BWD_VARIANT);
        if (vec && dma_ok(g) && option(OPT_CORR_NO_MFMA) == 0 && (v == 0 || v == 11))
            return corr_mfma_backward(x1, x2, go, g1, g2, g,
                                      std::is_same<T, __half>::value ? CERB_F16 : CERB_BF16, s);
    }
    if constexpr (sizeof(T) == 4) {
        // coarse levels (W <= 64): three waves per (output row, gradient, channel set), no loader, one barrier
        // (corr_coarse.hip).  4 pairs: 6.9 vs 11.9 us at 256 x 16 x 32, 11.4 vs 16.4-18.7 us at 128 x 32 x 64; 8 pairs of
        // the latter (4096 workgroups, since the items are chunked over the XCDs by image): 18.9 vs 20.0 us.  14 forces it, 15 is the
        // dispatch without it.
        const int v = option(OPT_CORR_BWD_VARIANT);
        const int64_t coarse_wgs = static_cast<int64_t>(g.B) * 2 * g.H * (g.C / (g.W > 32 ? 16 : 32));
        if ((vec || half) && dma_ok(g) && (v == 14 || ((v == 0 || v == 13) && g.W <= 64 && coarse_wgs <= 4096))) {   // 13 = auto, minus the strip kernel on 64-wide maps
            const int rc = corr_coarse_backward(x1, x2, go, g1, g2, g, s);
            if (rc != CERB_EUNSUPPORTED) return rc;
        }
    }
    if constexpr (sizeof(T) == 4) {
        // variant 12 forces the strip kernel on every shape it supports (any W % 4 == 0 up to 256), the narrow maps included
        if (g.W <= 32 && vec && dma_ok(g) && option(OPT_CORR_BWD_VARIANT) == 12) {
            const int rc = corr_strip_backward(x1, x2, go, g1, g2, g, s);
            if (rc != CERB_EUNSUPPORTED) return rc;
        }
    }
    if (g.W <= 32) {
        if constexpr (sizeof(T) == 4) {
            if (vec && dma_ok(g) && option(OPT_CORR_BWD_VARIANT) != 1)
                return launch_bwd_dma<BwdDmaNarrow>("corr_bwd_d4_dma_16x32", x1, x2, go, g1, g2, g, s);
        }
        return launch_bwd<BwdNarrow, T>("corr_bwd_d4_16x32", x1, x2, go, g1, g2, g, vec, s);
    }
    if constexpr (sizeof(T) == 4) {
        // whole image rows per wavefront, horizontal neighbours by DPP (corr_strip.hip).  Ten
        // barrier-separated steps per workgroup make it latency-bound on small problems: it is
        // the default where a level has enough eight-wave workgroups for the chip -- 256-wide:
        // 31.6 vs 38.1 us at 4 pairs of 32 x 128 x 256, 20.4 vs 25.0 at two, 18.1 vs 16.2 at one;
        // 128- and 64-wide (2 / 4 rows per wavefront): a tie or a small loss launch by launch
        // (21.0 vs 21.3, 18.5 vs 16.4 us) but 2.9 % more pairs/s in the whole step (0.3765 vs
        // 0.3875 ms, three alternating runs: it leaves LDS and L2 to the other stream's kernel).
        // Variant 12 forces it on every shape it supports, 13 keeps it off the 64-wide maps.
        // Round 6: widths between the powers of two run on the lanes of the next one (StripCfg RAG: lanes past the row's
        // end idle, any H, any C).  4 pairs, us, the tile kernels -> this: 32 x 112 x 224 44.8 -> 32.0, 64 x 44 x 152
        // 30.7 -> 28.1, 128 x 22 x 76 28.6 -> 20.1, 64 x 56 x 112 20.9 -> 20.7; at 64 lanes and below (128 x 28 x 56:
        // 17.1 -> 19.4) the displacement-group kernel stays ahead, so only the exact 64-wide maps take it there.
        const int v = option(OPT_CORR_BWD_VARIANT);
        const int rows_per_wg = g.W > 128 ? 2 : g.W > 64 ? 4 : 8;
        const int64_t strip_wgs = static_cast<int64_t>(g.B) * ((g.H + rows_per_wg - 1) / rows_per_wg) * ((g.C + 31) / 32) * 2;
        const bool strip_auto = (g.W > 64 && g.W <= 256 && strip_wgs >= 192) ||
                                (g.W == 64 && g.H % 8 == 0 && g.C % 16 == 0 && strip_wgs >= 128 && v != 13);
        if (vec && dma_ok(g) && (v == 12 || ((v == 0 || v == 13) && strip_auto))) {
            const int rc = corr_strip_backward(x1, x2, go, g1, g2, g, s);
            if (rc != CERB_EUNSUPPORTED) return rc;
        }
    }
    switch (option(OPT_CORR_BWD_VARIANT)) {
        case 1: return launch_bwd<BwdWide, T>("corr_bwd_d4_8x64", x1, x2, go, g1, g2, g, vec, s);
#ifdef CERB_EXPERIMENTS
        case 2: return launch_bwd_g3<BwdG3Wide, T>("corr_bwd_d4_g3_4x64", x1, x2, go, g1, g2, g, vec, s);
#endif
        case 3: return launch_bwd_g3<BwdG3Wide4, T>("corr_bwd_d4_g3_8x64", x1, x2, go, g1, g2, g, vec, s);
        case 4:
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g))
                    return launch_bwd_dma<BwdDma2x5>("corr_bwd_d4_dma_8x64", x1, x2, go, g1, g2, g, s);
            }
            break;
        case 5:
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g))
                    return launch_bwd_dma<BwdDmaNarrow>("corr_bwd_d4_dma_16x32", x1, x2, go, g1, g2, g, s);
            }
            break;
        case 8:
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g))
                    return launch_bwd_rows<BwdRows84>("corr_bwd_d4_rows_4x64_cb8", x1, x2, go, g1, g2, g, s);
            }
            break;
#ifdef CERB_EXPERIMENTS   // measured and rejected: test builds only
        case 6:
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g))
                    return launch_bwd_rows<BwdRows>("corr_bwd_d4_rows_4x64", x1, x2, go, g1, g2, g, s);
            }
            break;
        case 10:
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g))
                    return launch_bwd_col<BwdCol>("corr_bwd_d4_col_4x64", x1, x2, go, g1, g2, g, s);
            }
            break;
        case 7: case 9:
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g)) {
                    if (option(OPT_CORR_BWD_VARIANT) == 7)
                        return launch_bwd_rows<BwdRows44>("corr_bwd_d4_rows_4x64_c16", x1, x2, go, g1, g2, g, s);
                    return launch_bwd_rows<BwdRows82>("corr_bwd_d4_rows_4x64_cb8_c16", x1, x2, go, g1, g2, g, s);
                }
            }
            break;
#endif
        default: break;
    }
    // few tiles (coarse level): the displacement-group kernel puts 3x the wavefronts on the
    // problem (measured 18 vs 21 us on the 128x32x64 level); larger maps prefer all-81
    const int64_t tiles = static_cast<int64_t>(g.B) * ((g.W + 63) / 64) * ((g.H + 7) / 8);
    if (tiles <= 32 && g.C >= 16)
        return launch_bwd_g3<BwdG3Wide4, T>("corr_bwd_d4_g3_8x64", x1, x2, go, g1, g2, g, vec, s);
    if constexpr (sizeof(T) == 4) {
        if (vec && dma_ok(g)) {
            // medium maps: the displacement-row streaming kernel reads gradOutput once per
            // (tile, side) instead of once per channel slice (64x64x128 x4: 21.5 vs 25.9 us);
            // on the largest maps the all-81-in-registers kernel is still ahead (39 vs 41-45 us)
            if (tiles <= 128)
                return launch_bwd_rows<BwdRows84>("corr_bwd_d4_rows_4x64_cb8", x1, x2, go, g1, g2, g, s);
            // same arithmetic as corr_bwd_d4_kernel, the channel window streamed by LDS-DMA
            return launch_bwd_dma<BwdDma2x5>("corr_bwd_d4_dma_8x64", x1, x2, go, g1, g2, g, s);
        }
    }
    return launch_bwd<BwdWide, T>("corr_bwd_d4_8x64", x1, x2, go, g1, g2, g, vec, s);
}


}  // namespace

int corr_d4_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
                     const CorrGeom &g, int dtype, hipStream_t s) {
    if (!fast_config(g, dtype)) return CERB_EUNSUPPORTED;
    const bool vec = g.W % 4 == 0 && aligned_group(in1, dtype) && aligned_group(in2, dtype) &&
                     aligned_group(gout, dtype) && aligned_group(gin1, dtype) &&
                     aligned_group(gin2, dtype);
    // W % 4 == 2: the coarse-level kernel takes the row's last two pixels as half a strip (corr_coarse.hip, round 6)
    const bool half = g.W % 4 == 2 && aligned_group(in1, dtype) && aligned_group(in2, dtype) && aligned_group(gout, dtype) &&
                      aligned_group(gin1, dtype) && aligned_group(gin2, dtype);
    switch (dtype) {
        case CERB_F32: return bwd_dispatch<float>(in1, in2, gout, gin1, gin2, g, vec, half, s);
        case CERB_F16: return bwd_dispatch<__half>(in1, in2, gout, gin1, gin2, g, vec, half, s);
        case CERB_BF16: return bwd_dispatch<hip_bfloat16>(in1, in2, gout, gin1, gin2, g, vec, half, s);
        default: return CERB_EUNSUPPORTED;
    }
}


}  // namespace cerb
