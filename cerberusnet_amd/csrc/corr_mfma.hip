// corr_mfma.hip -- the correlation for 16-bit storage (fp16 / bf16) on the matrix cores: the backward (band-matrix
// formulation, below), and -- further down -- the forward in three forms: register-staged with transposing LDS writes
// (rounds 4-5; today: 65 .. 128 channels and widths off a multiple of 8), tiles by LDS-DMA + ds_read_b64_tr_b16 operands
// standing still (variant 26), and the same walking down a column of tiles (round 6, the default for 16 < C <= 64).
//
// Same function as corr_bwd_d4_kernel (reference: correlation_backward_input1 / _input2,
// /root/reference/nnet_training/correlation_package/correlation_cuda_kernel.cu:97-242, at
// pad = d = 4, k = 1, s1 = s2 = 1), both sides in one launch:
//   gradInput1[c][y][x] = 1/C * sum_{ey,ex} gradOutput[(ey,ex)][y][x]         * x2[c][y+ey][x+ex]
//   gradInput2[c][y][x] = 1/C * sum_{ey,ex} gradOutput[(-ey,-ex)][y+ey][x+ex] * x1[c][y+ey][x+ex]
// i.e. out[c][p] = sum_e g[e][p] * src[c][p + e] with g = gradOutput (side 0) or its flipped,
// shifted read (side 1).
//
// Why the matrix cores, and only for 16-bit storage: with fp32 FMAs this kernel is bound by
// LDS operand bandwidth and VALU issue, not by HBM -- at the 2048x1024 fp16 pyramid the VALU
// kernel runs at 17 % of the HBM roofline on half the bytes of fp32 (profiles/r02_*config5*).
// The products of two fp16 / bf16 values are exact in fp32 and v_mfma_f32_16x16x32_{f16,bf16}
// accumulates in fp32, so the matrix cores compute what the VALU kernel computes (fp32
// products and sums of the stored 16-bit values) up to summation order.  fp32 storage stays on
// the VALU kernels: the fp32 MFMA rate equals the packed-FMA rate on gfx950.
//
// Formulation: for an output row y, a 16-pixel segment starting at x0 and one vertical
// displacement ey, out[p][c] += sum_q A[p][q] * S[q][c] with
//   S[q][c] = src[c][y + ey][x0 - 4 + q]        (q = 0..31: a row of the source window)
//   A[p][q] = g[(ey, q - p - 4)][y][x0 + p]     for 0 <= q - p <= 8, else 0  (a band matrix)
// = ONE v_mfma_f32_16x16x32 with M = 16 pixels, N = 16 channels, K = 32 window columns (24
// used).  S is read from LDS exactly as it lies in memory (NCHW rows: a lane's 8 consecutive
// k are 8 consecutive pixels of one channel = one ds_read_b128), and the band matrix is built
// once per (row, ey) and reused for every channel block: a lane keeps ITS pixel's A-row in LDS
// (24 halves, zeros written once, the nine band slots p..p+8 rewritten per ey), so an A operand
// is one ds_read_b128 too.
//
// Work decomposition: a workgroup = 4 waves = 4 consecutive rows x 64 pixels x 32 channels of
// one side; a wave owns one row (corr_bwd_d4_mfma_kernel, rounds 2-4; variant 11) or -- the default since
// round 5, corr_bwd_d4_mfma_seg_kernel below -- one 16-pixel segment of all four rows:
// D = 4 segments (rows) x 2 channel blocks x 4 VGPRs.  LDS: the
// 12 x 72 source window of the 32 channels (62 KB, channel stride = 16 B mod 128 B: the 16
// channels of a B read start in different banks) + the A rows (12 KB): 2 workgroups / CU.
//
// Non-finite inputs: a band matrix has explicit zeros, and 0 * Inf = NaN inside the MFMA: an
// Inf / NaN in the source would reach every pixel of its 16-pixel segment whose row window
// holds it.  That can only add non-finite results, so a row with a non-finite accumulator is
// recomputed tap by tap (see the kernel): NaN / Inf reach exactly the vector kernels' elements.
#include <atomic>
#include <utility>

#include "common.h"

namespace cerb {
namespace {

[[maybe_unused]] constexpr int kD = 4, kND = 2 * kD + 1;
[[maybe_unused]] constexpr int kDead = static_cast<int>(0x80000000u);

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <typename T> struct Mma;
template <> struct Mma<__half> {
    static __device__ __forceinline__ f4v run(u4v a, u4v b, f4v c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        const __half2 h = __floats2half2_rn(a, b);
        return __builtin_bit_cast(unsigned, h);
    }
    static __device__ __forceinline__ float widen(unsigned short bits) {
        return static_cast<float>(__builtin_bit_cast(_Float16, bits));
    }
};
template <> struct Mma<hip_bfloat16> {
    static __device__ __forceinline__ f4v run(u4v a, u4v b, f4v c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, bf2v));
    }
    static __device__ __forceinline__ float widen(unsigned short bits) {
        return __uint_as_float(static_cast<unsigned>(bits) << 16);
    }
};

struct BwdMfmaCfg {
    static constexpr int TH = 4, TW = 64, NSEG = TW / 16;   // rows (= waves), pixels, 16-pixel segments
    static constexpr int CB = 2, CS = 16 * CB;              // channel blocks / channels per workgroup
    static constexpr int WR = TH + 2 * kD;                  // window rows
    static constexpr int WC = TW + 16;                      // window columns held (x0-4 .. x0+75; 72 loaded, 8 zero)
    static constexpr int UPR = WC / 4;                      // 8-byte units per window row
    static constexpr int CSTR = WR * WC + 8;                // channel stride in halves: 16 B mod 128 B
    static constexpr int WIN = CS * CSTR;                   // halves
    static constexpr int AROW = 24;                         // halves of a pixel's band row
    static constexpr int THREADS = 64 * TH;
    static constexpr size_t LDS_BYTES = 2 * (WIN + TH * TW * AROW);
    static_assert((CSTR * 2) % 128 == 16, "channel stride must stagger the banks");
    static_assert(NSEG * 16 == TW && UPR * 4 == WC, "whole segments and units");
};

template <typename T>
__global__ __launch_bounds__(BwdMfmaCfg::THREADS, 2) void corr_bwd_d4_mfma_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, const T *__restrict__ gout,
    T *__restrict__ gin1, T *__restrict__ gin2, int C, int H, int W, int tiles_x, int tiles_y,
    int nslice, int nwalk, int dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef CERB_ABLATE
    dbg = 0;   // timing ablations exist in -DCERB_ABLATE builds only (tools/ablate_mfma.py)
#endif
    using K = BwdMfmaCfg;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short *win = smem;                 // [CS][WR ring slots][WC] halves, channel stride CSTR
    unsigned short *arow = smem + K::WIN;       // [TH][TW][AROW]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    // A workgroup WALKS DOWN nwalk vertically adjacent tiles of one (item, column, channel slice,
    // side): the 12 window rows live in a ring (slot = (row + 4) mod 12), so every further tile
    // loads 4 new rows instead of 12 (window traffic 3x -> ~1x the source), and those loads, as
    // well as the next tile's 81 gradOutput values, are in flight while the current tile computes.
    const int walks_y = (tiles_y + nwalk - 1) / nwalk;
    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int side = __builtin_amdgcn_readfirstlane(bid & 1); bid >>= 1;   // 0: gradInput1
    const int slice = __builtin_amdgcn_readfirstlane(bid % nslice); bid /= nslice;
    const int tx = __builtin_amdgcn_readfirstlane(bid % tiles_x); bid /= tiles_x;
    const int wy = __builtin_amdgcn_readfirstlane(bid % walks_y);
    const int b = __builtin_amdgcn_readfirstlane(bid / walks_y);
    const int ty_begin = wy * nwalk, ty_end = min(tiles_y, ty_begin + nwalk);
    const int x0 = tx * K::TW;
    const int c_begin = slice * K::CS;
    const int plane = H * W;

    const T *src = (side == 0 ? x2 : x1) + static_cast<int64_t>(b) * C * plane;
    T *dst = (side == 0 ? gin1 : gin2) + static_cast<int64_t>(b) * C * plane;
    const __amdgpu_buffer_rsrc_t rsrc_src = uniform_rsrc(src, C * plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_dst = uniform_rsrc(dst, C * plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_go =
        uniform_rsrc(gout + static_cast<int64_t>(b) * (kND * kND) * plane, kND * kND * plane * 2);

    // ---- gradOutput addressing (lane = pixel of this wave's row) ----
    // side 0: g[e][y][x] = gO[e][y][x];  side 1: g[e][y][x] = gO[80 - e][y + ey][x + ex].
    // A load costs no vector arithmetic: nine per-lane column offsets (one per ex), a scalar
    // row / plane offset per ey, and the plane step per ex added on the scalar unit.
    const int gx = x0 + lane;
    int g_voff[kND];
#pragma unroll
    for (int i = 0; i < kND; ++i) {
        const int xx = side ? gx + i - kD : gx;
        g_voff[i] = (xx >= 0 && xx < W) ? xx * 2 : kDead;
    }
    const int g_step = side ? -plane * 2 : plane * 2;   // plane step per ex
    unsigned short gv[kND][kND];
    auto load_g = [&](int y) {
#pragma unroll
        for (int j = 0; j < kND; ++j) {
            const int yy = side ? y + j - kD : y;
            const bool rok = yy >= 0 && yy < H && y < H;     // wave-uniform
            const int d0 = side ? kND * kND - 1 - j * kND : j * kND;
            const int row_off = __builtin_amdgcn_readfirstlane((d0 * plane + yy * W) * 2);
#pragma unroll
            for (int i = 0; i < kND; ++i) {
                gv[j][i] = 0;
                if (rok && !(dbg & 1))
                    gv[j][i] = __builtin_amdgcn_raw_buffer_load_b16(rsrc_go, g_voff[i], row_off + i * g_step, 0);
            }
        }
    };
    load_g(ty_begin * K::TH + wave);   // in flight during the window copy

    // ---- first tile: the whole 12-row window -> ring; zero the A rows ----
    // A thread owns ONE 8-byte unit position (row, 4 columns) for all channels: the channel is the
    // scalar offset of the load and an immediate offset of the LDS store.
    {
        constexpr int POS = K::WR * K::UPR;
        static_assert(POS <= K::THREADS, "one unit position per thread");
        const int y0 = ty_begin * K::TH;
        const int row = tid / K::UPR, un = tid % K::UPR;
        const int sy = y0 - kD + row, sx = x0 - kD + 4 * un;
        const bool ok = tid < POS && un < (K::TW + 2 * kD) / 4 && sy >= 0 && sy < H && sx >= 0 && sx < W;
        const int w_voff = ok ? (sy * W + sx) * 2 : kDead;
        const int slot = (y0 + row) % K::WR;                // ring slot of window row (sy + 4)
        u2v v[K::CS];
#pragma unroll
        for (int ch = 0; ch < K::CS; ++ch) {
            v[ch] = u2v{0, 0};
            if (c_begin + ch < C && !(dbg & 2))             // uniform
                v[ch] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_src, w_voff, (c_begin + ch) * plane * 2, 0);
        }
        if (tid < POS && !(dbg & 4)) {
            unsigned short *w = win + slot * K::WC + un * 4;
#pragma unroll
            for (int ch = 0; ch < K::CS; ++ch) *reinterpret_cast<u2v *>(w + ch * K::CSTR) = v[ch];
        }
        unsigned short *mine = arow + (wave * K::TW + lane) * K::AROW;
#pragma unroll
        for (int i = 0; i < K::AROW / 8; ++i) *reinterpret_cast<u4v *>(mine + 8 * i) = u4v{0, 0, 0, 0};
    }
    __syncthreads();

    // next tile's 4 new rows: thread -> (unit position of the 4 x 20 units, every 3rd channel)
    constexpr int NPOS = K::TH * K::UPR, NCG = K::THREADS / NPOS, NPF = (K::CS + NCG - 1) / NCG;
    const int pf_pos = tid % NPOS, pf_cg = tid / NPOS;      // pf_cg == NCG: idle thread
    const int pf_row = pf_pos / K::UPR, pf_un = pf_pos % K::UPR;
    const int pf_sx = x0 - kD + 4 * pf_un;
    const bool pf_col_ok = pf_cg < NCG && pf_un < (K::TW + 2 * kD) / 4 && pf_sx >= 0 && pf_sx < W;

    const int p = lane & 15, kg = lane >> 4;
    unsigned short *my_arow = arow + (wave * K::TW + lane) * K::AROW + p;   // band slots p .. p + 8
    const unsigned short *a_rd = arow + (wave * K::TW + p) * K::AROW + 8 * (kg < 3 ? kg : 0);
    const unsigned short *b_lane = win + p * K::CSTR + 8 * kg;
    const float inv_nelems = 1.0f / static_cast<float>(C);

    for (int ty = ty_begin; ty < ty_end; ++ty) {
        const int y0 = ty * K::TH, y = y0 + wave;
        const bool more = ty + 1 < ty_end;
        // ---- prefetch the next tile's new window rows (rows y0 + 8 .. y0 + 11) into registers ----
        u2v pf[NPF];
        if (more) {
            const int sy = y0 + K::TH + kD + pf_row;
            const int base = (pf_col_ok && sy < H) ? (sy * W + pf_sx) * 2 : kDead;
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const int ch = pf_cg + NCG * k;
                const bool on = ch < K::CS && c_begin + ch < C;
                pf[k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_src, on ? base + (c_begin + ch) * plane * 2 : kDead, 0, 0);
            }
        }
        // ---- 9 vertical displacements x (4 segments x CB channel blocks) MFMAs ----
        f4v acc[K::NSEG][K::CB];
#pragma unroll
        for (int s = 0; s < K::NSEG; ++s)
#pragma unroll
            for (int cb = 0; cb < K::CB; ++cb) acc[s][cb] = f4v{0.f, 0.f, 0.f, 0.f};
        int slot = __builtin_amdgcn_readfirstlane((y0 + wave) % K::WR);   // ring slot of window row y - 4
        // Software pipeline over the displacement rows: while the MFMAs of row ey run on the A
        // operands already in registers, the band of row ey + 1 is written and read back (LDS
        // executes a wave's operations in order, so the one band-row buffer is enough).
        // The band rows a lane writes are read by OTHER lanes of its wave: the hardware runs a
        // wave's LDS operations in order, and the wavefront-scope fence + wave barrier (no
        // instruction) keeps the compiler from moving the reads above the writes, or the next
        // band's writes above these reads, whatever it can prove about the addresses.
        auto wave_lds_fence = [] {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        auto write_band = [&](int eyi) {
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < kND; ++i)
                if (!(dbg & 8)) my_arow[i] = gv[eyi][i];
            wave_lds_fence();
        };
        auto read_a = [&](u4v (&a)[K::NSEG]) {
#pragma unroll
            for (int s = 0; s < K::NSEG; ++s) {
                a[s] = *reinterpret_cast<const u4v *>(a_rd + s * 16 * K::AROW);
                if (kg == 3) a[s] = u4v{0, 0, 0, 0};   // k = 24..31: outside every band
            }
        };
        u4v a_cur[K::NSEG], a_nxt[K::NSEG];
        write_band(0);
        read_a(a_cur);
#pragma unroll
        for (int eyi = 0; eyi < kND; ++eyi) {
            if (eyi + 1 < kND) {
                write_band(eyi + 1);
                read_a(a_nxt);
            }
            if (!(dbg & 16)) {
                const unsigned short *b_rd = b_lane + slot * K::WC;
#pragma unroll
                for (int cb = 0; cb < K::CB; ++cb)
#pragma unroll
                    for (int s = 0; s < K::NSEG; ++s) {
                        const u4v bv = *reinterpret_cast<const u4v *>(b_rd + cb * 16 * K::CSTR + 16 * s);
                        acc[s][cb] = Mma<T>::run(a_cur[s], bv, acc[s][cb]);
                    }
            }
#pragma unroll
            for (int s = 0; s < K::NSEG; ++s) a_cur[s] = a_nxt[s];
            slot = slot + 1 == K::WR ? 0 : slot + 1;
        }
        // ---- the next tile's gradOutput values: in flight during the stores and the ring update ----
        if (more) load_g(y + K::TH);
        // ---- non-finite results: redo this row exactly ----
        // The band matrix holds explicit zeros and 0 x Inf = NaN inside an MFMA, so an Inf / NaN of
        // the source can spread over its whole 16-pixel segment.  It can only ADD non-finite
        // results, never hide one: a row whose accumulators are all finite is right as it is, and
        // a row with a non-finite accumulator (a diverged step) is recomputed tap by tap -- lane =
        // pixel, the window and this pixel's 81 gradOutput values are still at hand -- so that
        // NaN / Inf reach exactly the elements they reach in the vector kernels.
        // (one test on the SUM of the lane's accumulators: NaN and Inf survive addition)
        f4v tot = acc[0][0];
#pragma unroll
        for (int s = 0; s < K::NSEG; ++s)
#pragma unroll
            for (int cb = 0; cb < K::CB; ++cb)
                if (s + cb) tot += acc[s][cb];
        const float tsum = (tot[0] + tot[1]) + (tot[2] + tot[3]);
        const bool bad = (__float_as_uint(tsum) & 0x7F800000u) == 0x7F800000u;
        const bool redo = __builtin_amdgcn_ballot_w64(bad) != 0;   // wave-uniform: a wave owns a row
        if (redo && y < H) {
            // (gradOutput is read again instead of keeping the 81 registers alive across this
            // rarely taken branch: that cost the common path 8 % in spills)
            const int slot0 = (y0 + wave) % K::WR;
            for (int c = 0; c < K::CS && c_begin + c < C; ++c) {
                const unsigned short *wc = win + c * K::CSTR + lane;
                float sum = 0.f;
                int sl = slot0;
#pragma unroll 1
                for (int j = 0; j < kND; ++j) {
                    const int yy = side ? y + j - kD : y;
                    const bool rok = yy >= 0 && yy < H;
                    const int d0 = side ? kND * kND - 1 - j * kND : j * kND;
                    const int row_off = __builtin_amdgcn_readfirstlane((d0 * plane + (rok ? yy : 0) * W) * 2);
#pragma unroll
                    for (int i = 0; i < kND; ++i) {
                        const unsigned short gb = __builtin_amdgcn_raw_buffer_load_b16(
                            rsrc_go, rok ? g_voff[i] : kDead, row_off + i * g_step, 0);
                        sum = fmaf(Mma<T>::widen(gb), Mma<T>::widen(wc[sl * K::WC + i]), sum);
                    }
                    sl = sl + 1 == K::WR ? 0 : sl + 1;
                }
                const int x = x0 + lane;
                __builtin_amdgcn_raw_buffer_store_b16(
                    static_cast<unsigned short>(Mma<T>::pack2(sum * inv_nelems, 0.f)), rsrc_dst,
                    x < W ? ((c_begin + c) * plane + y * W + x) * 2 : kDead, 0, 0);
            }
        }
        // ---- D[pixel][channel] -> gradInput[c][y][x]: 4 consecutive pixels of one channel per lane ----
        if (y < H && !redo && !((dbg & 32) && acc[0][0][0] + acc[3][1][3] != 12345.f)) {
#pragma unroll
            for (int cb = 0; cb < K::CB; ++cb) {
                const int c = c_begin + cb * 16 + p;
#pragma unroll
                for (int s = 0; s < K::NSEG; ++s) {
                    const int x = x0 + 16 * s + 4 * kg;
                    const f4v d = acc[s][cb];
                    const u2v o = {Mma<T>::pack2(d[0] * inv_nelems, d[1] * inv_nelems),
                                   Mma<T>::pack2(d[2] * inv_nelems, d[3] * inv_nelems)};
                    __builtin_amdgcn_raw_buffer_store_b64(
                        o, rsrc_dst, (c < C && x < W) ? (c * plane + y * W + x) * 2 : kDead, 0, 0);
                }
            }
        }
        if (!more) break;
        // ---- ring update: the new rows replace the 4 oldest (rows y0 - 4 .. y0 - 1) ----
        __syncthreads();   // every wave is done reading this tile's window
        if (pf_cg < NCG) {
            const int nslot = (y0 + K::TH + 2 * kD + pf_row) % K::WR;   // window row (y0 + 8 + pf_row) + 4
            unsigned short *w = win + nslot * K::WC + pf_un * 4;
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const int ch = pf_cg + NCG * k;
                if (ch < K::CS && !(dbg & 4)) *reinterpret_cast<u2v *>(w + ch * K::CSTR) = pf[k];
            }
        }
        __syncthreads();
    }
#endif
}

// ----------------------------------------------------------------------------------------------
// Round 5: the same tile with a wave per 16-pixel SEGMENT (all four rows) instead of a wave per row.
// Why: the compute phase of the kernel above is bound by LDS operand reads, not by the matrix cores
// (per tile and wave 72 B reads + 36 A reads of 1 KB each against 72 MFMAs of 16 cycles: the CU's
// 128 B / clk feed four SIMDs).  A window row r serves the outputs (row, ey) with row + ey = r, and a
// wave that owns all four rows of its segment keeps the B operands of four consecutive window rows in
// registers: per displacement step ey it reads ONE new window row (2 B reads) and the four band rows
// (4 A reads) for its eight MFMAs -- 24 + 36 reads per tile instead of 72 + 36.  gradOutput is loaded
// by lane = (row, pixel of the segment); a lane's nine band values go out as five packed dwords; the
// four rows' A blocks are skewed by 32 bytes so that they start in different banks.  Same products,
// same fp32 accumulation order per output (ey ascending, one MFMA per ey): identical bits.
struct BwdSegCfg : BwdMfmaCfg {
    static constexpr int ABLK = 16 * AROW + 16;             // halves per (wave, row) block of band rows: 16 pixels + 32 B skew
    static constexpr size_t LDS_BYTES = 2 * (WIN + TH * TH * ABLK);
};

// (SIDE is a template parameter so that each side's body is straight-line code: with the side's two load patterns as
// run-time branches inside the displacement loop the compiler's wait-count pass lost track of the loads in flight and
// waited for each row of values right after requesting it -- 71 -> 92 us at 32 x 256 x 512)
template <typename T, int SIDE>
__device__ __forceinline__ void corr_bwd_seg_body(
    const T *__restrict__ x1, const T *__restrict__ x2, const T *__restrict__ gout,
    T *__restrict__ gin1, T *__restrict__ gin2, int C, int H, int W, int tiles_x, int tiles_y,
    int nslice, int nwalk) {
#if defined(__HIP_DEVICE_COMPILE__)
    using K = BwdSegCfg;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short *win = smem;                 // [CS][WR ring slots][WC] halves, channel stride CSTR
    unsigned short *arow = smem + K::WIN;       // [wave = segment][row][16 pixels][AROW] (+ skew)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    const int walks_y = (tiles_y + nwalk - 1) / nwalk;
    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    constexpr int side = SIDE; bid >>= 1;   // 0: gradInput1
    const int slice = __builtin_amdgcn_readfirstlane(bid % nslice); bid /= nslice;
    const int tx = __builtin_amdgcn_readfirstlane(bid % tiles_x); bid /= tiles_x;
    const int wy = __builtin_amdgcn_readfirstlane(bid % walks_y);
    const int b = __builtin_amdgcn_readfirstlane(bid / walks_y);
    const int ty_begin = wy * nwalk, ty_end = min(tiles_y, ty_begin + nwalk);
    const int x0 = tx * K::TW;
    const int c_begin = slice * K::CS;
    const int plane = H * W;

    const T *src = (side == 0 ? x2 : x1) + static_cast<int64_t>(b) * C * plane;
    T *dst = (side == 0 ? gin1 : gin2) + static_cast<int64_t>(b) * C * plane;
    const __amdgpu_buffer_rsrc_t rsrc_src = uniform_rsrc(src, C * plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_dst = uniform_rsrc(dst, C * plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_go =
        uniform_rsrc(gout + static_cast<int64_t>(b) * (kND * kND) * plane, kND * kND * plane * 2);

    // ---- gradOutput: lane = (row of the tile, pixel of this wave's segment) ----
    // A wave-level load instruction costs the address path ~9 cycles per CU whether its lanes fetch 2 or 4 bytes
    // (tools/ubench/vmem_rate.hip), and 81 halfword loads per lane and tile made this path the kernel's largest single
    // cost.  Two neighbouring lanes (pixels 2m, 2m + 1) therefore fetch DWORDS -- both pixels of one plane each, two
    // different planes -- and swap halves (one DPP move + one v_perm at the time the values are used):
    //   side 0 (no shifts): planes (0,1) (2,3) (4,5) (6,7) (8,-) of a displacement row: 5 loads instead of 9;
    //   side 1 (plane ex is read at column x + ex): the even shifts pair up the same way -- (0,2) (4,6) (8,-) -- the odd
    //   ones would pair lanes (2m - 1, 2m) across the segment's ends and stay halfword loads: 7 instead of 9.
    const int grow = lane >> 4, gp = lane & 15;
    const int gx = x0 + 16 * wave + gp;
    const bool odd = gp & 1;
    const int g_step = side ? -plane * 2 : plane * 2;   // plane step per ex
    // Per-lane byte offsets: column + the plane's distance from the LOWEST plane of the displacement row (the scalar offset
    // of all its loads), so that every offset is non-negative and one SGPR serves a row.  An invalid column is 2^30 (a batch
    // item is smaller: out of range whatever is added), an invalid row 2^31.
    constexpr int NRAW = 7;
    constexpr int kDeadCol = 0x40000000;
    const int pstep = plane * 2;
    int g_off[NRAW];
    {
        const int pc = gx & ~1;                           // first column of the pair's dword at shift 0
        auto col = [&](int c, int planes) { return (c >= 0 && c < W) ? c * 2 + planes * pstep : kDeadCol; };
        if (!side) {
#pragma unroll
            for (int k = 0; k < 4; ++k) g_off[k] = col(pc, 2 * k + (odd ? 1 : 0));
            g_off[4] = odd ? kDeadCol : col(pc, 8);
            g_off[5] = g_off[6] = kDeadCol;
        } else {       // plane ex of the row lies (8 - ex) planes above its lowest
            g_off[0] = odd ? col(pc - 2, 6) : col(pc - 4, 8);     // planes ex = 2 | 0
            g_off[1] = odd ? col(pc + 2, 2) : col(pc, 4);         // planes ex = 6 | 4
            g_off[2] = odd ? kDeadCol : col(pc + 4, 0);           // plane ex = 8
#pragma unroll
            for (int k = 0; k < 4; ++k) g_off[3 + k] = col(gx + 2 * k + 1 - kD, 7 - 2 * k);   // planes ex = 1, 3, 5, 7: halfwords
        }
    }
    unsigned raw[kND][5];            // the dwords (side 1: the first three)
    unsigned short rawh[kND][4];     // side 1: the halfwords of the odd shifts
    auto load_g_row = [&](int j, int y) {     // the nine values of displacement row j; y: this lane's row
        const int yy = side ? y + j - kD : y;
        const bool rok = yy >= 0 && yy < H && y < H;
        const int lowest = side ? kND * kND - 1 - j * kND - (kND - 1) : j * kND;
        const int base_s = __builtin_amdgcn_readfirstlane(lowest * pstep);
        const int row_off = rok ? yy * W * 2 : kDead;
        if (!side) {
#pragma unroll
            for (int k = 0; k < 5; ++k) raw[j][k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_go, g_off[k] + row_off, base_s, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) raw[j][k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_go, g_off[k] + row_off, base_s, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) rawh[j][k] = __builtin_amdgcn_raw_buffer_load_b16(rsrc_go, g_off[3 + k] + row_off, base_s, 0);
        }
    };
    // (this lane's value, the value of the partner's plane) from the dwords the two lanes of a pair fetched
    const unsigned swap_sel = odd ? 0x03020706u : 0x05040100u;
    auto exchange = [&](unsigned own) {
        const unsigned partner = static_cast<unsigned>(__builtin_amdgcn_mov_dpp(static_cast<int>(own), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        return __builtin_amdgcn_perm(partner, own, swap_sel);
    };
#pragma unroll
    for (int j = 0; j < kND; ++j) load_g_row(j, ty_begin * K::TH + grow);   // in flight during the window copy

    // ---- first tile: the whole 12-row window -> ring; zero the A rows ----
    unsigned short *my_blk = arow + (wave * K::TH + grow) * K::ABLK;   // this lane's (wave, row) block
    {
        constexpr int POS = K::WR * K::UPR;
        static_assert(POS <= K::THREADS, "one unit position per thread");
        const int y0 = ty_begin * K::TH;
        const int row = tid / K::UPR, un = tid % K::UPR;
        const int sy = y0 - kD + row, sx = x0 - kD + 4 * un;
        const bool ok = tid < POS && un < (K::TW + 2 * kD) / 4 && sy >= 0 && sy < H && sx >= 0 && sx < W;
        const int w_voff = ok ? (sy * W + sx) * 2 : kDead;
        const int slot = (y0 + row) % K::WR;                // ring slot of window row (sy + 4)
        u2v v[K::CS];
#pragma unroll
        for (int ch = 0; ch < K::CS; ++ch) {
            v[ch] = u2v{0, 0};
            if (c_begin + ch < C)                           // uniform
                v[ch] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_src, w_voff, (c_begin + ch) * plane * 2, 0);
        }
        if (tid < POS) {
            unsigned short *w = win + slot * K::WC + un * 4;
#pragma unroll
            for (int ch = 0; ch < K::CS; ++ch) *reinterpret_cast<u2v *>(w + ch * K::CSTR) = v[ch];
        }
        unsigned short *mine = my_blk + gp * K::AROW;
#pragma unroll
        for (int i = 0; i < K::AROW / 8; ++i) *reinterpret_cast<u4v *>(mine + 8 * i) = u4v{0, 0, 0, 0};
    }
    __syncthreads();

    // next tile's 4 new rows: thread -> (unit position of the 4 x 20 units, every 3rd channel)
    constexpr int NPOS = K::TH * K::UPR, NCG = K::THREADS / NPOS, NPF = (K::CS + NCG - 1) / NCG;
    const int pf_pos = tid % NPOS, pf_cg = tid / NPOS;      // pf_cg == NCG: idle thread
    const int pf_row = pf_pos / K::UPR, pf_un = pf_pos % K::UPR;
    const int pf_sx = x0 - kD + 4 * pf_un;
    const bool pf_col_ok = pf_cg < NCG && pf_un < (K::TW + 2 * kD) / 4 && pf_sx >= 0 && pf_sx < W;

    const int p = lane & 15, kg = lane >> 4;
    // band slots gp .. gp + 8 of this lane's A row, as five dwords starting at the even slot gp & ~1 (the half below an odd
    // gp and the half above an even gp + 8 lie outside every band: they stay zero)
    unsigned *my_band = reinterpret_cast<unsigned *>(my_blk + gp * K::AROW + (gp & ~1));
    const unsigned short *a_rd = arow + wave * K::TH * K::ABLK + p * K::AROW + 8 * (kg < 3 ? kg : 0);
    const unsigned short *b_lane = win + p * K::CSTR + 8 * kg + 16 * wave;
    const float inv_nelems = 1.0f / static_cast<float>(C);

    for (int ty = ty_begin; ty < ty_end; ++ty) {
        const int y0 = ty * K::TH;
        const bool more = ty + 1 < ty_end;
        u2v pf[NPF];
        if (more) {
            const int sy = y0 + K::TH + kD + pf_row;
            const int base = (pf_col_ok && sy < H) ? (sy * W + pf_sx) * 2 : kDead;
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const int ch = pf_cg + NCG * k;
                const bool on = ch < K::CS && c_begin + ch < C;
                pf[k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_src, on ? base + (c_begin + ch) * plane * 2 : kDead, 0, 0);
            }
        }
        f4v acc[K::TH][K::CB];
#pragma unroll
        for (int r = 0; r < K::TH; ++r)
#pragma unroll
            for (int cb = 0; cb < K::CB; ++cb) acc[r][cb] = f4v{0.f, 0.f, 0.f, 0.f};
        auto wave_lds_fence = [] {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        auto write_band = [&](int eyi) {
            unsigned pk[5];       // (value 2k, value 2k + 1) of this lane's pixel
            if (!side) {
#pragma unroll
                for (int k = 0; k < 5; ++k) pk[k] = exchange(raw[eyi][k]);
            } else {
                const unsigned xa = exchange(raw[eyi][0]), xb = exchange(raw[eyi][1]);   // (v0, v2), (v4, v6)
                pk[0] = __builtin_amdgcn_perm(rawh[eyi][0], xa, 0x05040100u);
                pk[1] = __builtin_amdgcn_perm(rawh[eyi][1], xa, 0x05040302u);
                pk[2] = __builtin_amdgcn_perm(rawh[eyi][2], xb, 0x05040100u);
                pk[3] = __builtin_amdgcn_perm(rawh[eyi][3], xb, 0x05040302u);
                pk[4] = exchange(raw[eyi][2]);
            }
            wave_lds_fence();
            unsigned prev = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) {       // an odd pixel's dwords start one slot lower: (value 2k - 1, value 2k)
                my_band[k] = odd ? __builtin_amdgcn_alignbit(pk[k], prev, 16) : pk[k];
                prev = pk[k];
            }
            wave_lds_fence();
        };
        auto read_a = [&](u4v (&a)[K::TH]) {
#pragma unroll
            for (int r = 0; r < K::TH; ++r) {
                a[r] = *reinterpret_cast<const u4v *>(a_rd + r * K::ABLK);
                if (kg == 3) a[r] = u4v{0, 0, 0, 0};   // k = 24..31: outside every band
            }
        };
        // B operands of window rows t .. t + 3 (ring slots), both channel blocks: a sliding register window
        const int slot0 = __builtin_amdgcn_readfirstlane(y0 % K::WR);    // ring slot of window row 0 (image row y0 - 4)
        auto read_b = [&](int r, u4v (&bw)[K::CB]) {
            int sl = slot0 + r;
            sl = sl >= K::WR ? sl - K::WR : sl;
            const unsigned short *b_rd = b_lane + sl * K::WC;
#pragma unroll
            for (int cb = 0; cb < K::CB; ++cb) bw[cb] = *reinterpret_cast<const u4v *>(b_rd + cb * 16 * K::CSTR);
        };
        u4v bwin[K::TH][K::CB];
        u4v a_cur[K::TH], a_nxt[K::TH];
        // The next tile's gradOutput values are requested row by row, each as soon as its registers are free (its band has
        // been written): every row of values gets a whole tile period to arrive.  (Requested together behind the MFMA loop,
        // as the kernel above does, the trip to memory was exposed once per tile.)
        const int ly = y0 + grow;                                    // this lane's row as a gradOutput / redo pixel
        write_band(0);
        if (more) load_g_row(0, ly + K::TH);
        read_a(a_cur);
#pragma unroll
        for (int r = 0; r < K::TH - 1; ++r) read_b(r, bwin[r]);
#pragma unroll
        for (int eyi = 0; eyi < kND; ++eyi) {
            read_b(eyi + K::TH - 1, bwin[(eyi + K::TH - 1) % K::TH]);
            if (eyi + 1 < kND) {
                write_band(eyi + 1);
                if (more) load_g_row(eyi + 1, ly + K::TH);
                read_a(a_nxt);
            }
#pragma unroll
            for (int r = 0; r < K::TH; ++r)
#pragma unroll
                for (int cb = 0; cb < K::CB; ++cb)
                    acc[r][cb] = Mma<T>::run(a_cur[r], bwin[(r + eyi) % K::TH][cb], acc[r][cb]);
#pragma unroll
            for (int r = 0; r < K::TH; ++r) a_cur[r] = a_nxt[r];
        }
        // ---- non-finite results: redo this segment exactly (see the kernel above) ----
        f4v tot = acc[0][0];
#pragma unroll
        for (int r = 0; r < K::TH; ++r)
#pragma unroll
            for (int cb = 0; cb < K::CB; ++cb)
                if (r + cb) tot += acc[r][cb];
        const float tsum = (tot[0] + tot[1]) + (tot[2] + tot[3]);
        const bool bad = (__float_as_uint(tsum) & 0x7F800000u) == 0x7F800000u;
        const bool redo = __builtin_amdgcn_ballot_w64(bad) != 0;   // wave-uniform
        if (redo) {
            const int x = gx;
            for (int c = 0; c < K::CS && c_begin + c < C; ++c) {
                const unsigned short *wc = win + c * K::CSTR + 16 * wave + gp;
                float sum = 0.f;
                int sl = (y0 + grow) % K::WR;
#pragma unroll 1
                for (int j = 0; j < kND; ++j) {
                    const int yy = side ? ly + j - kD : ly;
                    const bool rok = yy >= 0 && yy < H && ly < H;
                    const int d0 = side ? kND * kND - 1 - j * kND : j * kND;
                    const int plane_off = __builtin_amdgcn_readfirstlane(d0 * plane * 2);
#pragma unroll
                    for (int i = 0; i < kND; ++i) {
                        const int xx = side ? gx + i - kD : gx;
                        const unsigned short gb = __builtin_amdgcn_raw_buffer_load_b16(
                            rsrc_go, (rok && xx >= 0 && xx < W) ? (yy * W + xx) * 2 : kDead, plane_off + i * g_step, 0);
                        sum = fmaf(Mma<T>::widen(gb), Mma<T>::widen(wc[sl * K::WC + i]), sum);
                    }
                    sl = sl + 1 == K::WR ? 0 : sl + 1;
                }
                __builtin_amdgcn_raw_buffer_store_b16(
                    static_cast<unsigned short>(Mma<T>::pack2(sum * inv_nelems, 0.f)), rsrc_dst,
                    (x < W && ly < H) ? ((c_begin + c) * plane + ly * W + x) * 2 : kDead, 0, 0);
            }
        }
        // ---- D[pixel][channel] -> gradInput[c][y][x]: 4 consecutive pixels of one channel per lane ----
        if (!redo) {
#pragma unroll
            for (int cb = 0; cb < K::CB; ++cb) {
                const int c = c_begin + cb * 16 + p;
#pragma unroll
                for (int r = 0; r < K::TH; ++r) {
                    const int x = x0 + 16 * wave + 4 * kg, y = y0 + r;
                    const f4v d = acc[r][cb];
                    const u2v o = {Mma<T>::pack2(d[0] * inv_nelems, d[1] * inv_nelems),
                                   Mma<T>::pack2(d[2] * inv_nelems, d[3] * inv_nelems)};
                    __builtin_amdgcn_raw_buffer_store_b64(
                        o, rsrc_dst, (c < C && x < W && y < H) ? (c * plane + y * W + x) * 2 : kDead, 0, 0);
                }
            }
        }
        if (!more) break;
        // ---- ring update: the new rows replace the 4 oldest (rows y0 - 4 .. y0 - 1) ----
        __syncthreads();   // every wave is done reading this tile's window
        if (pf_cg < NCG) {
            const int nslot = (y0 + K::TH + 2 * kD + pf_row) % K::WR;   // window row (y0 + 8 + pf_row) + 4
            unsigned short *w = win + nslot * K::WC + pf_un * 4;
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const int ch = pf_cg + NCG * k;
                if (ch < K::CS) *reinterpret_cast<u2v *>(w + ch * K::CSTR) = pf[k];
            }
        }
        __syncthreads();
    }
#endif
}

template <typename T>
__global__ __launch_bounds__(BwdSegCfg::THREADS, 2) void corr_bwd_d4_mfma_seg_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, const T *__restrict__ gout,
    T *__restrict__ gin1, T *__restrict__ gin2, int C, int H, int W, int tiles_x, int tiles_y,
    int nslice, int nwalk) {
    if (xcd_chunk(blockIdx.x, gridDim.x) & 1)    // uniform: consecutive workgroups are the two sides of one tile column
        corr_bwd_seg_body<T, 1>(x1, x2, gout, gin1, gin2, C, H, W, tiles_x, tiles_y, nslice, nwalk);
    else
        corr_bwd_seg_body<T, 0>(x1, x2, gout, gin1, gin2, C, H, W, tiles_x, tiles_y, nslice, nwalk);
}

// ============================================================================
// forward
// ============================================================================
// out[(dy,dx)][y][x] = leaky(1/C * sum_c x1[c][y][x] * x2[c][y+dy][x+dx]): for an output row, a
// 16-pixel segment and one dy, D[p][q] = sum_c x1[c][p] * x2[c][q] over a 16-column block of
// the x2 row is ONE v_mfma_f32_16x16x32 per 32 channels (M = pixels, N = window columns, K =
// channels); the nine wanted entries of a row of D are the diagonals q - p - 4 + 8 h = dx of
// two blocks h = 0, 1 (columns x0 - 4 + 8 h ..).  No operand holds explicit zeros, so NaN / Inf
// reach exactly the outputs they reach in the VALU kernels.
// The contraction runs over channels, so a lane's 8 consecutive k must be 8 consecutive
// CHANNELS of one pixel: the tiles are transposed to pixel-major / channel-minor on their way
// into LDS (4 x 4 blocks in registers; the reference does the same as a separate pass over
// global memory, correlation_cuda_kernel.cu:13-27).  A pixel's 16-byte chunks are XOR-swizzled
// by its column so that the 16 lanes of an operand read start in different banks.
// A wave owns one row: per dy 8 MFMAs; the wanted diagonals are scaled, LeakyReLU'd, rounded
// and passed through a per-wave LDS tile to become whole 128 / 64-byte plane rows.
// NKB = 32-channel blocks held at once (C <= 32 NKB; 1, 2 or 4); the tile is 4 rows x 64 / NKB pixels.
// (Walking down a column of tiles as the backward does -- x2 rows in a ring, the next tile's rows
// prefetched -- was measured and dropped: 32x256x512 55.6 -> 53.5 us, but 32x128x256 16.4 -> 18.2
// and 64x64x128 9.7 -> 11.3: the forward's traffic is dominated by its 81-plane output, not by
// the window halo.)
template <int NKB_, int DS_ = (NKB_ == 4 ? 3 : 2)>
struct FwdMfmaCfg {
    static constexpr int NKB = NKB_, NSEG = 4 / NKB, TW = 16 * NSEG, TH = 4;
    static constexpr int PIXB = 64 * NKB;                      // bytes per pixel (all channels)
    static constexpr int WR = TH + 2 * kD, WCOL = TW + 2 * kD; // window rows / columns
    static constexpr int GRP = 4 * PIXB + 16;                  // 4 pixels + 16 bytes: see pix()
    static constexpr int WROW = (WCOL / 4) * GRP, XROW = (TW / 4) * GRP;   // row pitches in bytes
    static constexpr int WIN_B = WR * WROW, X1_B = TH * XROW;
    static constexpr int TP = TW + 1;                          // T row pitch: odd, so a diagonal of D spreads over the banks
    static constexpr int T_B = (kND * TP + 64) * 4;            // per wave: 9 planes x TW pixels fp32 + 64 dump slots
    // DS waves share a row of the tile, each taking 9 / DS of the vertical displacements: more waves per staged window (the
    // same MFMAs and stores; identical bits).  NKB = 4 (65 .. 128 channels; round 5): the window alone is 75 KB -- one
    // workgroup per CU -- and three waves per row make twelve per window.  Measured (4 pairs, fp16, us, DS = 1 / 2 / 3):
    // 128 x 64 x 128 18.1 / 16.3 / 15.2 (the vector kernel: 22.8), 64 x 128 x 256 27.1 / 22.9 / 22.6, 64 x 64 x 128 10.2 / 8.2 / 8.4,
    // 32 x 256 x 512 53.7 / 49.6 / 59.9, 32 x 128 x 256 16.7 / 15.6 / 17.2
    static constexpr int DS = DS_;                             // waves per row
    static constexpr int DPW = (kND + DS - 1) / DS;            // vertical displacements per wave
    static constexpr int THREADS = 64 * TH * DS;
    static constexpr size_t LDS_BYTES = WIN_B + (X1_B > TH * DS * T_B ? X1_B : TH * DS * T_B);   // the T tiles reuse the x1 tile's LDS
    static constexpr int MINB = NKB == 4 ? 1 : 2;              // workgroups per CU (round 6: 4 x 32 tiles at 3 per CU for C <= 32 -- within 3 % either way)
    static_assert(NKB == 1 || NKB == 2 || NKB == 4, "channel blocks");
    // Byte offset, inside a row, of 16-byte chunk ci (8 channels) of the pixel at column col.
    // Two access patterns must both spread over the banks: the operand reads (16 lanes = 16
    // consecutive columns, same chunk) and the transposing writes (16 lanes = columns 4 apart).
    // 16 bytes of padding per 4 pixels staggers the 4-pixel groups, the XOR staggers the pixels
    // inside a group; without them the staging writes alone cost 14 us of a 64 us launch.
    static __device__ __forceinline__ int pix(int col, int ci) {
        const int sw = (NKB == 1 ? (col >> 1) & 1 : col & 3) << 1;
        return (col >> 2) * GRP + (col & 3) * PIXB + ((ci ^ sw) << 4);
    }
};

template <typename K, typename T>
__global__ __launch_bounds__(K::THREADS, K::MINB) void corr_fwd_d4_mfma_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, T *__restrict__ out, int C, int H, int W,
    int tiles_x, int tiles_y, float slope, int64_t out_bstride, int dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef CERB_ABLATE
    dbg = 0;   // timing ablations exist in -DCERB_ABLATE builds only (tools/ablate_mfma.py)
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *win = lds, *x1t = lds + K::WIN_B;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int tx = __builtin_amdgcn_readfirstlane(bid % tiles_x); bid /= tiles_x;
    const int ty = __builtin_amdgcn_readfirstlane(bid % tiles_y);
    const int b = __builtin_amdgcn_readfirstlane(bid / tiles_y);
    const int x0 = tx * K::TW, y0 = ty * K::TH;
    const int plane = H * W;
    const __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(x1 + static_cast<int64_t>(b) * C * plane, C * plane * 2);
    const __amdgpu_buffer_rsrc_t r2 = uniform_rsrc(x2 + static_cast<int64_t>(b) * C * plane, C * plane * 2);

    // ---- global (NCHW) -> LDS (pixel-major, channel-minor): units of 4 pixels x 4 channels ----
    {
        constexpr int CG = 8 * K::NKB;                                   // 4-channel groups
        constexpr int WU = K::WR * (K::WCOL / 4) * CG;                   // window units
        constexpr int XU = K::TH * (K::TW / 4) * CG;                     // x1 tile units
        constexpr int NU = (WU + XU + K::THREADS - 1) / K::THREADS;
        u2v v[NU][4];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            if (dbg & 1) { v[i][0] = v[i][1] = v[i][2] = v[i][3] = u2v{0x3c003c00u, 0x3c003c00u}; continue; }
            const int u = tid + K::THREADS * i;
            const bool isw = u < WU;
            const int uu = isw ? u : u - WU;
            constexpr int WG4 = K::WCOL / 4, XG4 = K::TW / 4;
            const int g4 = isw ? uu % WG4 : uu % XG4;                    // 4-pixel group in the row
            const int rest = isw ? uu / WG4 : uu / XG4;
            const int row = isw ? rest % K::WR : rest % K::TH;
            const int cg = isw ? rest / K::WR : rest / K::TH;
            const int gy = isw ? y0 - kD + row : y0 + row, gxx = isw ? x0 - kD + 4 * g4 : x0 + 4 * g4;
            const bool ok = u < WU + XU && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * cg + k;
                const int off = (ok && c < C) ? (c * plane + gy * W + gxx) * 2 : kDead;
                v[i][k] = isw ? __builtin_amdgcn_raw_buffer_load_b64(r2, off, 0, 0)
                              : __builtin_amdgcn_raw_buffer_load_b64(r1, off, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + K::THREADS * i;
            if (u >= WU + XU || (dbg & 2)) continue;
            const bool isw = u < WU;
            const int uu = isw ? u : u - WU;
            constexpr int WG4 = K::WCOL / 4, XG4 = K::TW / 4;
            const int g4 = isw ? uu % WG4 : uu % XG4;
            const int rest = isw ? uu / WG4 : uu / XG4;
            const int row = isw ? rest % K::WR : rest % K::TH;
            const int cg = isw ? rest / K::WR : rest / K::TH;
            unsigned char *base = isw ? win + row * K::WROW : x1t + row * K::XROW;
            // v[i][k] = channel 4cg+k, pixels (0,1 | 2,3) -> per pixel j: channels (0,1 | 2,3)
            const unsigned a0 = v[i][0].x, a1 = v[i][1].x, a2 = v[i][2].x, a3 = v[i][3].x;
            const unsigned b0 = v[i][0].y, b1 = v[i][1].y, b2 = v[i][2].y, b3 = v[i][3].y;
            const u2v px[4] = {
                u2v{(a0 & 0xFFFFu) | (a1 << 16), (a2 & 0xFFFFu) | (a3 << 16)},
                u2v{(a0 >> 16) | (a1 & 0xFFFF0000u), (a2 >> 16) | (a3 & 0xFFFF0000u)},
                u2v{(b0 & 0xFFFFu) | (b1 << 16), (b2 & 0xFFFFu) | (b3 << 16)},
                u2v{(b0 >> 16) | (b1 & 0xFFFF0000u), (b2 >> 16) | (b3 & 0xFFFF0000u)}};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 4 * g4 + j;
                *reinterpret_cast<u2v *>(base + K::pix(col, cg >> 1) + (cg & 1) * 8) = px[j];
            }
        }
    }
    __syncthreads();

    // ---- this wave's row ----
    const int p = lane & 15, kg = lane >> 4;
    const int row = wave % K::TH, part = wave / K::TH;   // (uniform: wave is)
    const int y = y0 + row;
    u4v a[K::NSEG][K::NKB];
#pragma unroll
    for (int s = 0; s < K::NSEG; ++s)
#pragma unroll
        for (int kb = 0; kb < K::NKB; ++kb) {
            const int col = 16 * s + p;
            a[s][kb] = *reinterpret_cast<const u4v *>(x1t + row * K::XROW + K::pix(col, 4 * kb + kg));
        }
    __syncthreads();   // every wave holds its x1 operands in registers: the tile's LDS becomes the T tiles
    float *tt = reinterpret_cast<float *>(x1t + wave * K::T_B);
    const float inv_nelems = 1.0f / static_cast<float>(C);
    const int64_t obs = out_bstride ? out_bstride : static_cast<int64_t>(kND * kND) * plane;
    const __amdgpu_buffer_rsrc_t ro = uniform_rsrc(out + b * obs, kND * kND * plane * 2);
    // D[m][q]: a lane holds q = p and m = 4 kg + j; dx = q - m - 4 + 8 h.  Block 0 serves the
    // pixels m < 8 (lanes kg < 2), block 1 the pixels m >= 8 (kg >= 2), and then
    // dx + 4 = p - 4 (kg & 1) - j for both: per j ONE predicate and ONE LDS slot per lane,
    // fixed for the whole kernel (a lane without a wanted entry writes a dump slot).
    const bool low = kg < 2;
    int t_slot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dxi = p - 4 * (kg & 1) - j;
        t_slot[j] = (dxi >= 0 && dxi < kND) ? dxi * K::TP + 4 * kg + j : kND * K::TP + p;   // (+ 16 s below: 64 dump slots)
    }
    constexpr int DPI = 64 / K::TW;                       // dx planes per store instruction
    const int st_px = lane % K::TW, st_dx = lane / K::TW;
    const int st_voff = (y < H && x0 + st_px < W) ? (st_dx * plane + y * W + x0 + st_px) * 2 : kDead;
#pragma unroll 1
    for (int dyi = part * K::DPW; dyi < min(kND, (part + 1) * K::DPW); ++dyi) {
        const unsigned char *wrow = win + (row + dyi) * K::WROW;
        f4v acc[K::NSEG][2];
#pragma unroll
        for (int s = 0; s < K::NSEG; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                acc[s][h] = f4v{0.f, 0.f, 0.f, 0.f};
                const int col = 16 * s + 8 * h + p;       // window column of this lane's q
#pragma unroll
                for (int kb = 0; kb < K::NKB; ++kb) {
                    const u4v bv = *reinterpret_cast<const u4v *>(wrow + K::pix(col, 4 * kb + kg));
                    acc[s][h] = Mma<T>::run(a[s][kb], bv, acc[s][h]);
                }
            }
        if (dbg & 4) {   // no T tile, no stores: one store keeps the MFMAs alive
            if (dyi == min(kND, (part + 1) * K::DPW) - 1 && acc[0][0][0] + acc[K::NSEG - 1][1][3] == 123.f) tt[0] = 1.f;
            continue;
        }
        // the T tile is written by one set of lanes and read back by another (same wave): fence as
        // in the backward's band rows (in-order LDS per wave; nothing may cross at compile time)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < K::NSEG; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) tt[t_slot[j] + 16 * s] = low ? acc[s][0][j] : acc[s][1][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (dbg & 8) continue;   // no read-back, no stores
        // whole plane rows: lane = (dx sub-plane, pixel); scale, LeakyReLU, one rounding
#pragma unroll
        for (int i = 0; i < (kND + DPI - 1) / DPI; ++i) {
            const int dxi = i * DPI + st_dx;
            const float q = tt[min(dxi, kND - 1) * K::TP + st_px] * inv_nelems;
            const float v = q > 0.f ? q : q * slope;
            const unsigned short bits = static_cast<unsigned short>(Mma<T>::pack2(v, v));   // one rounding instruction
            __builtin_amdgcn_raw_buffer_store_b16(bits, ro, dxi < kND ? st_voff : kDead,
                                                  __builtin_amdgcn_readfirstlane((dyi * kND) * plane * 2) +
                                                      (i * DPI) * plane * 2,
                                                  0);
        }
    }
#endif
}

// ---- forward, second form (round 6): the tiles as they lie in memory + the transposing LDS read -------------------------
// corr_fwd_d4_mfma_kernel above spends 20 of its 52 us (32 x 256 x 512, 4 pairs) bringing the tiles into LDS pixel-major:
// 20 loads per thread into registers, 4 x 4 transposes (v_perm), 80 ds_write_b64 per thread with the bank conflicts of a
// transposing write (SQ_LDS_BANK_CONFLICT: half of all LDS cycles of the launch).  gfx950 has the transposition in the LDS
// READ: ds_read_b64_tr_b16 hands each lane of a 16-lane group one COLUMN of a 4-row x 16-column block of 16-bit values.
// With rows = channels and columns = pixels that is an MFMA operand (a lane's k = consecutive channels of ITS pixel)
// straight from NCHW rows -- so the tiles enter LDS as they lie in memory, by LDS-DMA (buffer_load_dwordx4 ... lds: no
// registers, no VALU, no ds_write), in 16-byte cells of 8 pixels:
//   window : [32 channel slots][12 rows][9 cells], the first cell at column x0 - 4; slot stride 1760 B
//   x1 tile: [32 channel slots][ 4 rows][8 cells]; slot stride 544 B
// * Same operands as the first form: a lane's k = 8 kg + e is channel 8 kg + e (two transposing reads: e = 0..3, 4..7),
//   same MFMAs, same T tile, same stores -- identical bits (test).
// * Banks: the 32 lanes of a half read 8 channels x 32 B; (a / 4) mod 64 makes that conflict-free when the 8 channels start
//   in different 32-byte eighths of a 256-byte bank row.  A half's channels are {0-3, 8-11} (or + 4, + 16, + 20): channel c
//   lives in SLOT c with bits 2 and 3 swapped, and the slot stride is an odd number of 32-byte units.
// * A cell is copied whole (inside the image), arrives as zeros (outside: an out-of-range offset -- the zero padding), or
//   -- the cell that holds column 0 or W of a border tile (cells start at 4 mod 8) -- is left to a fix-up pass: its lanes
//   are masked out of the copy and a thread writes its inside half + zeros.  W % 8 == 0 (else: the first form).
// * T tile: row pitch 69 floats (5 mod 8, odd): the (up to) 32 lanes of a half that write a wanted entry hit 32
//   different banks (pitch 65: lanes kg = 0 and kg = 1 of one column met on one bank); lanes without one all write ONE
//   dump slot per segment (one address: no conflict among them).
template <int DS_ = 2>
struct FwdTrCfg {
    static constexpr int TW = 64, TH = 4, NSEG = TW / 16, DS = DS_, NCH = 32;
    static constexpr int WR = TH + 2 * kD;                            // window rows
    static constexpr int WCELL = (TW + 2 * kD) / 8, WPITCH = WCELL * 16;   // 9 cells = 144 B per window row
    static constexpr int CSLOT = WR * WPITCH + 32;                    // 1760 B: 55 units of 32 B
    static constexpr int XCELL = TW / 8, XPITCH = XCELL * 16;         // 8 cells = 128 B per x1 row
    static constexpr int XSLOT = TH * XPITCH + 32;                    // 544 B: 17 units
    static constexpr int WIN_B = NCH * CSLOT, X1_B = NCH * XSLOT;
    static constexpr int WIN_INST = WIN_B / 1024, X1_INST = X1_B / 1024, NDMA = WIN_INST + X1_INST;   // 55 + 17 copy instructions
    static constexpr int TP = 69, T_B = (kND * TP + 52) * 4;          // + the slots (one per segment) the lanes without a wanted entry write
    static constexpr int DPW = (kND + DS - 1) / DS;
    static constexpr int THREADS = 64 * TH * DS;
    static constexpr size_t LDS_BYTES = WIN_B + (X1_B > TH * DS * T_B ? X1_B : TH * DS * T_B);
    static_assert(WIN_B % 1024 == 0 && X1_B % 1024 == 0, "whole copy instructions");
    static_assert((CSLOT / 32) % 2 == 1 && (XSLOT / 32) % 2 == 1, "odd slot strides");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

typedef short s4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4v *lds_s4v_ptr;
typedef __attribute__((address_space(3))) void *lds_dma_ptr;
// 8 channels x this lane's pixel: the two transposing reads (channels 8 kg + 0..3 at `p`, 8 kg + 4..7 at `p + hi`)
__device__ __forceinline__ u4v tr_operand(const unsigned char *p, int hi) {
    const u2v lo = __builtin_bit_cast(u2v, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(p)));
    const u2v up = __builtin_bit_cast(u2v, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr)(p + hi)));
    return u4v{lo.x, lo.y, up.x, up.y};
}

template <typename K, typename T>
__global__ __launch_bounds__(K::THREADS, 2) void corr_fwd_d4_mfma_tr_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, T *__restrict__ out, int C, int H, int W,
    int tiles_x, int tiles_y, float slope, int64_t out_bstride, int dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef CERB_ABLATE
    dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *win = lds, *x1t = lds + K::WIN_B;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int tx = __builtin_amdgcn_readfirstlane(bid % tiles_x); bid /= tiles_x;
    const int ty = __builtin_amdgcn_readfirstlane(bid % tiles_y);
    const int b = __builtin_amdgcn_readfirstlane(bid / tiles_y);
    const int x0 = tx * K::TW, y0 = ty * K::TH;
    const int plane = H * W;
    const __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(x1 + static_cast<int64_t>(b) * C * plane, C * plane * 2);
    const __amdgpu_buffer_rsrc_t r2 = uniform_rsrc(x2 + static_cast<int64_t>(b) * C * plane, C * plane * 2);
    auto channel_of = [](int slot) { return (slot & 0x13) | ((slot & 4) << 1) | ((slot & 8) >> 1); };   // (its own inverse)

    // ---- global -> LDS: copy instruction k fills bytes 1024 k .. 1024 k + 1023; lane l owns cell 64 k + l ----
    {
        constexpr int NW = K::THREADS / 64;
#pragma unroll
        for (int i = 0; i < (K::NDMA + NW - 1) / NW; ++i) {
            const int k = wave + NW * i;                       // (uniform)
            if (k >= K::NDMA || (dbg & 1)) break;
            const bool isw = k < K::WIN_INST;
            int slot, row, col, gy, gx;
            bool real;
            if (isw) {
                const int cell = 64 * k + lane;
                slot = cell / (K::CSLOT / 16);
                const int r = cell - slot * (K::CSLOT / 16);
                real = r < K::WR * K::WCELL;
                row = r / K::WCELL; col = r - row * K::WCELL;
                gy = y0 - kD + row; gx = x0 - kD + 8 * col;
            } else {
                const int cell = 64 * (k - K::WIN_INST) + lane;
                slot = cell / (K::XSLOT / 16);
                const int r = cell - slot * (K::XSLOT / 16);
                real = r < K::TH * K::XCELL;
                row = r / K::XCELL; col = r - row * K::XCELL;
                gy = y0 + row; gx = x0 + 8 * col;
            }
            const int ch = channel_of(slot);
            const bool rowok = real && ch < C && gy >= 0 && gy < H;
            const bool inside = rowok && gx >= 0 && gx + 8 <= W;
            const bool straddle = rowok && !inside && gx + 8 > 0 && gx < W;   // an image edge inside the cell: the fix-up's
            const int voff = inside ? (ch * plane + gy * W + gx) * 2 : kDead;
            if (!straddle)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(isw ? r2 : r1, (lds_dma_ptr)(lds + k * 1024), 16, voff, 0, 0, 0);
        }
        // border tiles: the window cell that holds column 0 (side 0) or column W (side 1) of each row and channel
        if ((x0 == 0 || x0 + K::TW + kD > W) && !(dbg & 1)) {
            for (int it = tid; it < 2 * K::NCH * K::WR; it += K::THREADS) {
                const int side = it / (K::NCH * K::WR), rem = it - side * (K::NCH * K::WR);
                const int slot = rem / K::WR, row = rem - slot * K::WR;
                const int ch = channel_of(slot), gy = y0 - kD + row;
                const int col = side == 0 ? 0 : (W - x0) >> 3;
                const int gx = x0 - kD + 8 * col;
                const bool has = side == 0 ? x0 == 0 : (col < K::WCELL && gx < W && gx + 8 > W);
                if (has && ch < C && gy >= 0 && gy < H) {
                    const u2v d = __builtin_amdgcn_raw_buffer_load_b64(r2, (ch * plane + gy * W + (side == 0 ? 0 : gx)) * 2, 0, 0);
                    *reinterpret_cast<u4v *>(win + slot * K::CSLOT + row * K::WPITCH + col * 16) =
                        side == 0 ? u4v{0u, 0u, d.x, d.y} : u4v{d.x, d.y, 0u, 0u};
                }
            }
        }
    }
    __syncthreads();   // (waits for this wave's copies: vmcnt(0), then the barrier)

    // ---- this wave's row ----
    const int p = lane & 15, kg = lane >> 4;
    const int row = wave % K::TH, part = wave / K::TH;   // (uniform: wave is)
    const int y = y0 + row;
    // lane 4 q + r of a 16-lane group addresses row q (channel 8 kg + q -> slot q + 4 (kg & 1) + 16 (kg >> 1); + 8: channels + 4),
    // columns 4 r .. 4 r + 3 of the block
    const int slot0 = (p >> 2) + 4 * (kg & 1) + 16 * (kg >> 1);
    u4v a[K::NSEG];
    {
        const unsigned char *xa = x1t + slot0 * K::XSLOT + row * K::XPITCH + 8 * (p & 3);
#pragma unroll
        for (int s = 0; s < K::NSEG; ++s) a[s] = tr_operand(xa + 32 * s, 8 * K::XSLOT);
    }
    __syncthreads();   // every wave holds its x1 operands in registers: the tile's LDS becomes the T tiles
    float *tt = reinterpret_cast<float *>(x1t + wave * K::T_B);
    const float inv_nelems = 1.0f / static_cast<float>(C);
    const int64_t obs = out_bstride ? out_bstride : static_cast<int64_t>(kND * kND) * plane;
    const __amdgpu_buffer_rsrc_t ro = uniform_rsrc(out + b * obs, kND * kND * plane * 2);
    // D[m][q]: a lane holds q = p and m = 4 kg + j; dx = q - m - 4 + 8 h.  Block 0 serves the pixels m < 8 (lanes kg < 2),
    // block 1 the pixels m >= 8, and then dx + 4 = p - 4 (kg & 1) - j for both (see the first form)
    const bool low = kg < 2;
    int t_slot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dxi = p - 4 * (kg & 1) - j;
        t_slot[j] = (dxi >= 0 && dxi < kND) ? dxi * K::TP + 4 * kg + j : kND * K::TP;   // (+ 16 s below)
    }
    const int st_voff = (y < H && x0 + lane < W) ? (y * W + x0 + lane) * 2 : kDead;
    const unsigned char *wb = win + slot0 * K::CSLOT + row * K::WPITCH + 8 * (p & 3);
#pragma unroll 1
    for (int dyi = part * K::DPW; dyi < min(kND, (part + 1) * K::DPW); ++dyi) {
        const unsigned char *wrow = wb + dyi * K::WPITCH;
        f4v acc[K::NSEG][2];
#pragma unroll
        for (int s = 0; s < K::NSEG; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u4v bv = tr_operand(wrow + (16 * s + 8 * h) * 2, 8 * K::CSLOT);
                acc[s][h] = Mma<T>::run(a[s], bv, f4v{0.f, 0.f, 0.f, 0.f});
            }
        if (dbg & 4) {   // no T tile, no stores: one store keeps the MFMAs alive
            if (dyi == min(kND, (part + 1) * K::DPW) - 1 && acc[0][0][0] + acc[K::NSEG - 1][1][3] == 123.f) tt[0] = 1.f;
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < K::NSEG; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                tt[t_slot[j] + 16 * s] = low ? acc[s][0][j] : acc[s][1][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (dbg & 8) continue;   // no read-back, no stores
        // whole plane rows: lane = pixel; scale, LeakyReLU, one rounding
#pragma unroll
        for (int dxi = 0; dxi < kND; ++dxi) {
            const float q = tt[dxi * K::TP + lane] * inv_nelems;
            const float v = q > 0.f ? q : q * slope;
            const unsigned short bits = static_cast<unsigned short>(Mma<T>::pack2(v, v));
            __builtin_amdgcn_raw_buffer_store_b16(bits, ro, st_voff,
                                                  __builtin_amdgcn_readfirstlane((dyi * kND) * plane * 2) + dxi * plane * 2, 0);
        }
    }
#endif
}

// ---- forward, third form (round 6): the second form walking down a column of 4 x 32 tiles ------------------------------
// Ablation of the second form (32 x 256 x 512, 4 pairs, 48 us): without the copies 31, without the output 28, without both
// 15 -- the parts ADD, and each memory part runs at the memory system's rate while it runs (145 MB window + tile reads,
// 67 of them from HBM, in ~15 us; 85 MB of stores in ~17): every workgroup of the launch copies, then computes, then
// stores, in step with all the others.  Here a workgroup keeps the window ROWS in a ring of 16 and walks down its column:
// a further tile needs 4 new rows (window traffic 3.4 x -> 1.6 x the source) and their copies -- asynchronous, no
// registers -- are issued BEFORE the current tile is computed and stored, so that the three parts overlap inside every
// workgroup.  One barrier per tile.
//   window : [16 ring rows][32 channel slots][6 cells = 48 pixels from x0 - 8]   (3072 B per row; a slot = 3 units of 32 B)
//   x1 tile: 2 x [32 channel slots][4 rows][4 cells] + 2 cells                    (288 B per slot = 9 units)
// Cells start at multiples of 8 pixels of the IMAGE (x0 is a multiple of 32): with W % 8 == 0 a cell is inside or outside
// as a whole -- no fix-up pass.  Same operands, MFMAs, T tile values and rounding as the other two forms: identical bits.
template <int NKB_, int DS_, int XBUF_>
struct FwdWalkCfg {
    static constexpr int NKB = NKB_, DS = DS_, XBUF = XBUF_;                   // 32-channel blocks; waves per row; x1 tile buffers
    static constexpr int TW = 32, TH = 4, NSEG = TW / 16, NCH = 32 * NKB, RR = 16;
    static constexpr int WCELL = 6, WSLOT = WCELL * 16, WROW = NCH * WSLOT;   // 96 B; 3072 B per 32 channels
    static constexpr int WIN_B = RR * WROW;
    static constexpr int GRP_INST = TH * WROW / 1024;                          // copy instructions per group of 4 rows
    static constexpr int XSLOT = TH * (TW / 8) * 16 + 32, X1_B = NCH * XSLOT;  // 288 B; 9216 B per 32 channels
    static constexpr int X1_INST = X1_B / 1024;
    static constexpr int TP = 37;                                              // T row pitch in floats: 5 mod 8
    static constexpr int T_DUMP = 352, T_B = 1664;                             // floats 352 .. 399: the lanes without a wanted entry (see the kernel)
    static constexpr int DPW = (kND + DS - 1) / DS;
    static constexpr int NWAVE = TH * DS, THREADS = 64 * NWAVE;
    static constexpr int NI_W = (GRP_INST + NWAVE - 1) / NWAVE, NI_X = (X1_INST + NWAVE - 1) / NWAVE;   // per wave
    static constexpr int ST_PER_DY = (kND + 3) / 4;                            // store instructions per displacement row: 4 planes each
    static constexpr size_t LDS_BYTES = WIN_B + XBUF * X1_B + NWAVE * T_B;
    static constexpr int MINB = 2 * LDS_BYTES <= 160 * 1024 ? 2 : 1;          // workgroups per CU
    static_assert((TH * WROW) % 1024 == 0 && X1_B % 1024 == 0, "whole copy instructions");
    static_assert((WSLOT / 32) % 2 == 1 && (XSLOT / 32) % 2 == 1, "odd slot strides");
    static_assert(LDS_BYTES <= 160 * 1024, "fits a CU");
    static_assert(DPW * ST_PER_DY < 50 && NSEG == 2 && ST_PER_DY == 3, "the counted wait fits vmcnt; the loop body is written out for two segments");
    static_assert(T_B % 128 == 0 && T_DUMP % 32 == 0 && T_DUMP >= kND * TP + 16 && (T_DUMP + 48) * 4 <= T_B && (WIN_B + XBUF * X1_B) % 128 == 0,
                  "a T tile starts at bank 0; the dump slots behind the live ones");
};

template <int N> __device__ __forceinline__ void mfma_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most n (wave-uniform) vector-memory operations of this wave still outstanding
__device__ __forceinline__ void mfma_wait_vmcnt_upto(int n) {
    switch (n) {
#define W(k) case k: mfma_wait_vmcnt<k>(); break;
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19)
        W(20) W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31) W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39)
        W(40) W(41) W(42) W(43) W(44) W(45) W(46) W(47) W(48) W(49)
#undef W
        default: mfma_wait_vmcnt<50>(); break;
    }
}

// LDS accesses the compiler must not see as such: after an LDS-DMA it makes every ds_read / ds_write it knows of wait for
// the copy (it cannot tell the ring rows being filled from the ones being read), which would serialise exactly what this
// kernel overlaps.  Inline asm, in-order per wave, lgkmcnt by hand (the waited values are tied to the wait).
template <int OFF> __device__ __forceinline__ u2v asm_tr_read(unsigned addr) {
    u2v d;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
    return d;
}
typedef float f2v __attribute__((ext_vector_type(2)));
template <int OFF0> __device__ __forceinline__ f2v asm_lds_read2(unsigned addr) {      // dwords OFF0, OFF0 + 1 behind addr
    f2v d;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d) : "v"(addr), "n"(OFF0), "n"(OFF0 + 1));
    return d;
}
template <int OFF> __device__ __forceinline__ void asm_lds_write(unsigned addr, float v) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void asm_lgkm_wait(u2v &a, u2v &b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int... I, typename F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// MAXFORM: 0 < slope <= 1, LeakyReLU as max(v, v * slope) -- the same value as v > 0 ? v : v * slope for every input
// (slope 0 is not: -Inf * 0 = NaN), two instructions instead of three per value
template <typename K, typename T, bool MAXFORM>
__global__ __launch_bounds__(K::THREADS, K::MINB) void corr_fwd_d4_mfma_walk_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, T *__restrict__ out, int C, int H, int W,
    int tiles_x, int tiles_y, int nwalk, float slope, int64_t out_bstride, int dbg) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef CERB_ABLATE
    dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *win = lds, *x1t = lds + K::WIN_B;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    const int walks_y = (tiles_y + nwalk - 1) / nwalk;
    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int tx = __builtin_amdgcn_readfirstlane(bid % tiles_x); bid /= tiles_x;
    const int wy = __builtin_amdgcn_readfirstlane(bid % walks_y);
    const int b = __builtin_amdgcn_readfirstlane(bid / walks_y);
    const int x0 = tx * K::TW, ystart = wy * nwalk * K::TH;
    const int ntile = min(nwalk, tiles_y - wy * nwalk);            // (uniform)
    const int plane = H * W;
    const __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(x1 + static_cast<int64_t>(b) * C * plane, C * plane * 2);
    const __amdgpu_buffer_rsrc_t r2 = uniform_rsrc(x2 + static_cast<int64_t>(b) * C * plane, C * plane * 2);
    auto channel_of = [](int slot) { return (slot & ~0xc) | ((slot & 4) << 1) | ((slot & 8) >> 1); };   // (its own inverse)

    // ---- the copy plan: fixed per lane; a tile only moves the rows ----
    // window group (4 rows): instruction k fills bytes 1024 k.. of the group; this wave issues k = wave + NWAVE i
    int w_col[K::NI_W], w_row[K::NI_W];   // source byte offset without the row term (kDead: nothing to copy); row inside the group
#pragma unroll
    for (int i = 0; i < K::NI_W; ++i) {
        const int cell = 64 * (wave + K::NWAVE * i) + lane;
        const int rowin = cell / (K::NCH * K::WCELL), rem = cell - rowin * (K::NCH * K::WCELL);
        const int slot = rem / K::WCELL, cc = rem - slot * K::WCELL;
        const int ch = channel_of(slot), gx = x0 - 8 + 8 * cc;
        w_row[i] = rowin;
        w_col[i] = (ch < C && gx >= 0 && gx + 8 <= W) ? (ch * plane + gx) * 2 : kDead;
    }
    // x1 tile: instruction k = (wave + NWAVE / 2) % NWAVE + NWAVE i (the waves with fewer window copies first)
    int x_col[K::NI_X], x_row[K::NI_X];
#pragma unroll
    for (int i = 0; i < K::NI_X; ++i) {
        const int cell = 64 * ((wave + K::NWAVE / 2) % K::NWAVE + K::NWAVE * i) + lane;
        const int slot = cell / (K::XSLOT / 16), r = cell - slot * (K::XSLOT / 16);
        const int ch = channel_of(slot), gx = x0 + 8 * (r & 3);
        x_row[i] = r >> 2;
        x_col[i] = (r < 16 && ch < C && gx + 8 <= W) ? (ch * plane + gx) * 2 : kDead;
    }
    auto issue_rows = [&](int gy0, int ring_row) {      // image rows gy0 .. gy0 + 3 -> ring rows ring_row .. + 3 (uniform arguments)
        if (dbg & 1) return;
#pragma unroll
        for (int i = 0; i < K::NI_W; ++i) {
            const int k = wave + K::NWAVE * i;
            if (k >= K::GRP_INST) break;
            const int gy = gy0 + w_row[i];
            const int voff = (w_col[i] != kDead && gy >= 0 && gy < H) ? w_col[i] + gy * W * 2 : kDead;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_dma_ptr)(win + ring_row * K::WROW + k * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto issue_x1 = [&](int gy0, int buf) {
        if (dbg & 1) return;
#pragma unroll
        for (int i = 0; i < K::NI_X; ++i) {
            const int k = (wave + K::NWAVE / 2) % K::NWAVE + K::NWAVE * i;
            if (k >= K::X1_INST) break;
            const int gy = gy0 + x_row[i];
            const int voff = (x_col[i] != kDead && gy < H) ? x_col[i] + gy * W * 2 : kDead;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_dma_ptr)(x1t + buf * K::X1_B + k * 1024), 16, voff, 0, 0, 0);
        }
    };

    // ---- this wave's row / displacement rows; operand addresses ----
    const int p = lane & 15, kg = lane >> 4;
    const int row = wave % K::TH, part = wave / K::TH;
    const int dy_lo = part * K::DPW, dy_hi = min(kND, (part + 1) * K::DPW);
    const int slot0 = (p >> 2) + 4 * (kg & 1) + 16 * (kg >> 1);
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)lds));
    const unsigned xa = lds0 + K::WIN_B + slot0 * K::XSLOT + row * (K::TW * 2) + 8 * (p & 3);
    const unsigned wb = lds0 + slot0 * K::WSLOT + 8 * (p & 3) + 8;          // (+ 8: window column 4 is x0 - 4)
    const unsigned tt = lds0 + K::WIN_B + K::XBUF * K::X1_B + wave * K::T_B;
    const float inv_nelems = 1.0f / static_cast<float>(C);
    const int64_t obs = out_bstride ? out_bstride : static_cast<int64_t>(kND * kND) * plane;
    const __amdgpu_buffer_rsrc_t ro = uniform_rsrc(out + b * obs, kND * kND * plane * 2);
    const bool low = kg < 2;
    // T tile: float dxi * TP + m holds displacement dxi of pixel m.  Pitch 37 (5 mod 8): the lanes of a 32-lane half that
    // hold a wanted entry write 32 different banks; a lane without one writes the dump slot of the bank it WOULD have hit.
    unsigned t_slot[4];        // byte addresses (segment s: + 64)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dxi = p - 4 * (kg & 1) - j;
        const int nat = dxi * K::TP + 4 * kg + j;
        t_slot[j] = tt + 4 * ((dxi >= 0 && dxi < kND) ? nat : K::T_DUMP + (nat & 31));
    }
    // read-back / stores: lane = (plane 4 i + sub, pixels 2 pp, 2 pp + 1): ds_read2_b32 (lanes 0-15 even banks, 16-31 odd), one dword store
    const int st_pp = lane & 15, st_sub = lane >> 4;
    const int st_col = x0 + 2 * st_pp < W ? (st_sub * plane + x0 + 2 * st_pp) * 2 : kDead;
    const unsigned t_rd = tt + 4 * (st_sub * K::TP + 2 * st_pp);
    const unsigned t_rd_last = tt + 4 * ((kND - 1) * K::TP + 2 * st_pp);

    // ---- prologue: the first tile's 12 rows and its x1 tile ----
    issue_rows(ystart - kD, 0);
    issue_rows(ystart - kD + 4, 4);
    issue_rows(ystart - kD + 8, 8);
    issue_x1(ystart, 0);
    int stores_behind = 0;     // this wave's stores issued after its youngest copies
    constexpr int NOP = 4 * K::NKB;    // operand (pairs of transposing reads) per displacement row: (s, h, kb)
    for (int t = 0; t < ntile; ++t) {
        const int y0 = ystart + K::TH * t, y = y0 + row;
        mfma_wait_vmcnt_upto(stores_behind);     // in order: everything older than those stores -- the copies -- has landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (K::XBUF == 2 && t + 1 < ntile) {     // the next tile's rows replace the ring rows tile t - 1 read first
            issue_rows(y0 + K::TH + kD, (4 * t + 12) & (K::RR - 1));
            issue_x1(y0 + K::TH, (t + 1) & 1);
        }
        u4v a[K::NSEG][K::NKB];
        {
            const unsigned xat = xa + (K::XBUF == 2 ? (t & 1) * K::X1_B : 0);
            u2v lo[K::NSEG * K::NKB], hi[K::NSEG * K::NKB];
            static_for(std::make_integer_sequence<int, K::NSEG * K::NKB>{}, [&](auto I) {
                constexpr int s = I / K::NKB, kb = I % K::NKB;
                lo[I] = asm_tr_read<32 * s + kb * 32 * K::XSLOT>(xat);
                hi[I] = asm_tr_read<32 * s + kb * 32 * K::XSLOT + 8 * K::XSLOT>(xat);
            });
            static_for(std::make_integer_sequence<int, K::NSEG * K::NKB>{}, [&](auto I) {
                asm_lgkm_wait<0>(lo[I], hi[I]);
                a[I / K::NKB][I % K::NKB] = u4v{lo[I].x, lo[I].y, hi[I].x, hi[I].y};
            });
        }
        if (K::XBUF == 1) {                      // one x1 buffer: it is free once every wave holds its operands
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 1 < ntile) {
                issue_rows(y0 + K::TH + kD, (4 * t + 12) & (K::RR - 1));
                issue_x1(y0 + K::TH, 0);
            }
        }
        const int st_voff = (st_col != kDead && y < H) ? st_col + y * W * 2 : kDead;
        stores_behind = 0;
#pragma unroll 1
        for (int dyi = dy_lo; dyi < dy_hi; ++dyi) {
            const unsigned wrow = wb + ((4 * t + row + dyi) & (K::RR - 1)) * K::WROW;
            // operand (s, h, kb) at window columns 4 + 16 s + 8 h .., channels 32 kb ..: two transposing reads each, all in
            // flight, an MFMA as its pair arrives (kb ascending into one accumulator: the first form's order)
            u2v blo[NOP], bhi[NOP];
            static_for(std::make_integer_sequence<int, NOP>{}, [&](auto I) {
                constexpr int o = I / K::NKB, kb = I % K::NKB;
                blo[I] = asm_tr_read<16 * o + kb * 32 * K::WSLOT>(wrow);
                bhi[I] = asm_tr_read<16 * o + kb * 32 * K::WSLOT + 8 * K::WSLOT>(wrow);
            });
            f4v acc[K::NSEG][2];
            static_for(std::make_integer_sequence<int, NOP>{}, [&](auto I) {
                constexpr int o = I / K::NKB, kb = I % K::NKB, s = o >> 1, h = o & 1;
                asm_lgkm_wait<2 * (NOP - 1 - I)>(blo[I], bhi[I]);
                acc[s][h] = Mma<T>::run(a[s][kb], u4v{blo[I].x, blo[I].y, bhi[I].x, bhi[I].y},
                                        kb == 0 ? f4v{0.f, 0.f, 0.f, 0.f} : acc[s][h]);
            });
            if (dbg & 4) {
                if (dyi == dy_hi - 1 && acc[0][0][0] + acc[K::NSEG - 1][1][3] == 123.f) asm_lds_write<0>(tt, 1.f);
                continue;
            }
            // the wanted diagonals -> the T tile (in-order LDS per wave: the reads below see them)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm_lds_write<0>(t_slot[j], low ? acc[0][0][j] : acc[0][1][j]);
                asm_lds_write<64>(t_slot[j], low ? acc[1][0][j] : acc[1][1][j]);
            }
            if (dbg & 8) continue;
            // whole plane rows, four planes per instruction; scale, LeakyReLU, one rounding
            f2v q[K::ST_PER_DY];
            q[0] = asm_lds_read2<0>(t_rd);
            q[1] = asm_lds_read2<4 * K::TP>(t_rd);
            q[2] = asm_lds_read2<0>(t_rd_last);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]));
#pragma unroll
            for (int i = 0; i < K::ST_PER_DY; ++i) {
                const f2v sc = q[i] * f2v{inv_nelems, inv_nelems};
                f2v v;
                if constexpr (MAXFORM) {
                    const f2v neg = sc * f2v{slope, slope};
                    asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(sc.x), "v"(neg.x));
                    asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(sc.y), "v"(neg.y));
                } else {
                    v = f2v{sc.x > 0.f ? sc.x : sc.x * slope, sc.y > 0.f ? sc.y : sc.y * slope};
                }
                __builtin_amdgcn_raw_buffer_store_b32(Mma<T>::pack2(v.x, v.y), ro, 4 * i + st_sub < kND ? st_voff : kDead,
                                                      __builtin_amdgcn_readfirstlane((dyi * kND) * plane * 2) + (4 * i) * plane * 2, 0);
            }
            stores_behind += K::ST_PER_DY;
        }
    }
#endif
}

template <typename K, typename T>
int launch_fwd_mfma_walk(const char *name, const void *in1, const void *in2, void *outp, const CorrGeom &g, float slope,
                         int64_t obs, hipStream_t s) {
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    // tiles per walk: as many as still leave one round of resident workgroups
    const int64_t tiles = static_cast<int64_t>(g.B) * tiles_x * tiles_y, round = 256 * K::MINB;
    int nwalk = static_cast<int>(std::min<int64_t>(tiles_y, std::max<int64_t>(1, (tiles + round - 1) / round)));
    if (const int forced = option(OPT_CORR_BWD_CSLICE)) nwalk = std::max(1, std::min(forced, tiles_y));
    const int64_t blocks = static_cast<int64_t>(g.B) * tiles_x * ((tiles_y + nwalk - 1) / nwalk);
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    note_kernel(0, name);
    auto go = [&](auto maxform) {
        constexpr bool M = decltype(maxform)::value;
        static std::atomic<uint64_t> lds_done{0};
        if (const int rc = ensure_lds(corr_fwd_d4_mfma_walk_kernel<K, T, M>, K::LDS_BYTES, &lds_done)) return rc;
        hipLaunchKernelGGL((corr_fwd_d4_mfma_walk_kernel<K, T, M>), dim3(static_cast<unsigned>(blocks)), dim3(K::THREADS),
                           K::LDS_BYTES, s, static_cast<const T *>(in1), static_cast<const T *>(in2),
                           static_cast<T *>(outp), g.C, g.H, g.W, tiles_x, tiles_y, nwalk, slope, obs, debug_mask());
        return 0;
    };
    if (const int rc = (slope > 0.f && slope <= 1.f) ? go(std::true_type{}) : go(std::false_type{})) return rc;
    return launch_status();
}

template <typename K, typename T>
int launch_fwd_mfma_tr(const char *name, const void *in1, const void *in2, void *outp, const CorrGeom &g,
                       float slope, int64_t obs, hipStream_t s) {
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int64_t blocks = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    static std::atomic<uint64_t> lds_done{0};
    if (const int rc = ensure_lds(corr_fwd_d4_mfma_tr_kernel<K, T>, K::LDS_BYTES, &lds_done)) return rc;
    note_kernel(0, name);
    hipLaunchKernelGGL((corr_fwd_d4_mfma_tr_kernel<K, T>), dim3(static_cast<unsigned>(blocks)), dim3(K::THREADS),
                       K::LDS_BYTES, s, static_cast<const T *>(in1), static_cast<const T *>(in2),
                       static_cast<T *>(outp), g.C, g.H, g.W, tiles_x, tiles_y, slope, obs, debug_mask());
    return launch_status();
}

template <typename K, typename T>
int launch_fwd_mfma(const char *name, const void *in1, const void *in2, void *outp, const CorrGeom &g,
                    float slope, int64_t obs, hipStream_t s) {
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int64_t blocks = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    static std::atomic<uint64_t> lds_done{0};
    if (const int rc = ensure_lds(corr_fwd_d4_mfma_kernel<K, T>, K::LDS_BYTES, &lds_done)) return rc;
    note_kernel(0, name);
    hipLaunchKernelGGL((corr_fwd_d4_mfma_kernel<K, T>), dim3(static_cast<unsigned>(blocks)), dim3(K::THREADS),
                       K::LDS_BYTES, s, static_cast<const T *>(in1), static_cast<const T *>(in2),
                       static_cast<T *>(outp), g.C, g.H, g.W, tiles_x, tiles_y, slope, obs, debug_mask());
    return launch_status();
}

template <typename T>
int fwd_pick(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope, int64_t obs,
             hipStream_t s) {
    // round 6: W % 8 == 0 and C <= 64 take the column walk (LDS-DMA, transposing reads); corr_fwd_variant 20 keeps the
    // register-staged form of rounds 4-5 (also: other widths, 65 .. 128 channels), 26 the walk's stand-still form (C <= 32)
    const int v = option(OPT_CORR_FWD_VARIANT);
    const bool cells = g.W % 8 == 0 && v != 20;
    if (g.C <= 32) {
        if (cells && v == 26)
            return launch_fwd_mfma_tr<FwdTrCfg<2>, T>("corr_fwd_d4_mfma_tr_4x64", in1, in2, out, g, slope, obs, s);
        if (cells)
            return launch_fwd_mfma_walk<FwdWalkCfg<1, 2, 2>, T>("corr_fwd_d4_mfma_walk_4x32", in1, in2, out, g, slope, obs, s);
        return launch_fwd_mfma<FwdMfmaCfg<1>, T>("corr_fwd_d4_mfma_4x64", in1, in2, out, g, slope, obs, s);
    }
    if (g.C <= 64) {
        if (cells)   // 64 channel slots: one workgroup of 12 waves (3 per row) per CU
            return launch_fwd_mfma_walk<FwdWalkCfg<2, 3, 2>, T>("corr_fwd_d4_mfma_walk_4x32_c64", in1, in2, out, g, slope, obs, s);
        return launch_fwd_mfma<FwdMfmaCfg<2>, T>("corr_fwd_d4_mfma_4x32", in1, in2, out, g, slope, obs, s);
    }
    if (g.C <= 128)
        return launch_fwd_mfma<FwdMfmaCfg<4>, T>("corr_fwd_d4_mfma_4x16", in1, in2, out, g, slope, obs, s);
    return CERB_EUNSUPPORTED;
}

template <typename T>
int launch(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
           const CorrGeom &g, hipStream_t s, bool seg) {
    using K = BwdSegCfg;
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int nslice = (g.C + K::CS - 1) / K::CS;
    // tiles a workgroup walks down: as many as still leave ~512 workgroups = ONE round of 2 per
    // CU, every workgroup walking.  Measured (4 pairs, fp16, us by tiles per walk 1 / 2 / 4 / 8):
    // 32x128x256 26.9 / 23.6 / 30.6 / 52.9, 64x128x256 48.1 / 44.7 / 39.6 / 55.7, 32x256x512
    // 90.8 / 84.2 / 78.3 / 72.8: the best is where the launch is one round.
    const int64_t cols = static_cast<int64_t>(g.B) * tiles_x * nslice * 2;
    int nwalk = static_cast<int>(std::min<int64_t>(tiles_y, std::max<int64_t>(1, cols * tiles_y / 512)));
    if (const int forced = option(OPT_CORR_BWD_CSLICE)) nwalk = std::max(1, std::min(forced, tiles_y));
    const int64_t blocks = cols * ((tiles_y + nwalk - 1) / nwalk);
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    if (seg) {
        static std::atomic<uint64_t> lds_done_seg{0};
        if (const int rc = ensure_lds(corr_bwd_d4_mfma_seg_kernel<T>, K::LDS_BYTES, &lds_done_seg)) return rc;
        note_kernel(1, "corr_bwd_d4_mfma_seg_4x64");
        hipLaunchKernelGGL((corr_bwd_d4_mfma_seg_kernel<T>), dim3(static_cast<unsigned>(blocks)), dim3(K::THREADS),
                           K::LDS_BYTES, s, static_cast<const T *>(in1), static_cast<const T *>(in2),
                           static_cast<const T *>(gout), static_cast<T *>(gin1), static_cast<T *>(gin2), g.C,
                           g.H, g.W, tiles_x, tiles_y, nslice, nwalk);
        return launch_status();
    }
    static std::atomic<uint64_t> lds_done{0};
    if (const int rc = ensure_lds(corr_bwd_d4_mfma_kernel<T>, BwdMfmaCfg::LDS_BYTES, &lds_done)) return rc;
    note_kernel(1, "corr_bwd_d4_mfma_4x64");
    hipLaunchKernelGGL((corr_bwd_d4_mfma_kernel<T>), dim3(static_cast<unsigned>(blocks)), dim3(K::THREADS),
                       BwdMfmaCfg::LDS_BYTES, s, static_cast<const T *>(in1), static_cast<const T *>(in2),
                       static_cast<const T *>(gout), static_cast<T *>(gin1), static_cast<T *>(gin2), g.C,
                       g.H, g.W, tiles_x, tiles_y, nslice, nwalk, debug_mask());
    return launch_status();
}

}  // namespace

// 16-bit storage, pad = d = 4, k = 1, s1 = s2 = 1, W % 4 == 0, 8-byte aligned tensors, a batch
// item below 2^30 bytes (32-bit buffer offsets): checked by the caller (corr_d4.hip)
int corr_mfma_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
                       const CorrGeom &g, int dtype, hipStream_t s) {
    switch (dtype) {
        case CERB_F16: return launch<__half>(in1, in2, gout, gin1, gin2, g, s, option(OPT_CORR_BWD_VARIANT) != 11);
        case CERB_BF16: return launch<hip_bfloat16>(in1, in2, gout, gin1, gin2, g, s, option(OPT_CORR_BWD_VARIANT) != 11);
        default: return CERB_EUNSUPPORTED;
    }
}

// forward, 16-bit storage, C <= 128 (CERB_EUNSUPPORTED otherwise: the caller keeps its VALU kernels)
int corr_mfma_forward(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope,
                      int64_t obs, int dtype, hipStream_t s) {
    switch (dtype) {
        case CERB_F16: return fwd_pick<__half>(in1, in2, out, g, slope, obs, s);
        case CERB_BF16: return fwd_pick<hip_bfloat16>(in1, in2, out, g, slope, obs, s);
        default: return CERB_EUNSUPPORTED;
    }
}

}  // namespace cerb
