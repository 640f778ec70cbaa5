// corr_mfma.hip -- correlation backward for 16-bit storage (fp16 / bf16) on the matrix cores.
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
__global__ __launch_bounds__(K::THREADS, K::NKB == 4 ? 1 : 2) void corr_fwd_d4_mfma_kernel(
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
    if (g.C <= 32)
        return launch_fwd_mfma<FwdMfmaCfg<1>, T>("corr_fwd_d4_mfma_4x64", in1, in2, out, g, slope, obs, s);
    if (g.C <= 64)
        return launch_fwd_mfma<FwdMfmaCfg<2>, T>("corr_fwd_d4_mfma_4x32", in1, in2, out, g, slope, obs, s);
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
