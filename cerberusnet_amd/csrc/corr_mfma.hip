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
// one side; a wave owns one row: D = 4 segments x 2 channel blocks x 4 VGPRs.  LDS: the
// 12 x 72 source window of the 32 channels (62 KB, channel stride = 16 B mod 128 B: the 16
// channels of a B read start in different banks) + the A rows (12 KB): 2 workgroups / CU.
//
// Non-finite inputs: a band matrix has explicit zeros, and 0 * Inf = NaN inside the MFMA: an
// Inf / NaN in the source reaches every pixel of its 16-pixel segment whose row window holds
// it (up to 19 pixels away horizontally instead of 4).  Finite data -- every training step that
// has not already diverged -- is unaffected; the VALU kernels (corr_bwd_variant = 1) keep the
// reference's exact NaN reach.
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
    static_assert((CS * WR * UPR) % THREADS == 0, "window units per thread");
};

template <typename T>
__global__ __launch_bounds__(BwdMfmaCfg::THREADS, 2) void corr_bwd_d4_mfma_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, const T *__restrict__ gout,
    T *__restrict__ gin1, T *__restrict__ gin2, int C, int H, int W, int tiles_x, int tiles_y,
    int nslice) {
#if defined(__HIP_DEVICE_COMPILE__)
    using K = BwdMfmaCfg;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short *win = smem;                 // [CS][WR][WC] halves, channel stride CSTR
    unsigned short *arow = smem + K::WIN;       // [TH][TW][AROW]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int side = __builtin_amdgcn_readfirstlane(bid & 1); bid >>= 1;   // 0: gradInput1
    const int slice = __builtin_amdgcn_readfirstlane(bid % nslice); bid /= nslice;
    const int tx = __builtin_amdgcn_readfirstlane(bid % tiles_x); bid /= tiles_x;
    const int ty = __builtin_amdgcn_readfirstlane(bid % tiles_y);
    const int b = __builtin_amdgcn_readfirstlane(bid / tiles_y);
    const int x0 = tx * K::TW, y0 = ty * K::TH;
    const int c_begin = slice * K::CS;
    const int plane = H * W;

    const T *src = (side == 0 ? x2 : x1) + static_cast<int64_t>(b) * C * plane;
    T *dst = (side == 0 ? gin1 : gin2) + static_cast<int64_t>(b) * C * plane;
    const __amdgpu_buffer_rsrc_t rsrc_src = uniform_rsrc(src, C * plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_dst = uniform_rsrc(dst, C * plane * 2);
    const __amdgpu_buffer_rsrc_t rsrc_go =
        uniform_rsrc(gout + static_cast<int64_t>(b) * (kND * kND) * plane, kND * kND * plane * 2);

    // ---- this wave's row: gradOutput addressing (lane = pixel) ----
    const int y = y0 + wave;
    const int gx = x0 + lane;
    // side 0: g[e][y][x] = gO[e][y][x];  side 1: g[e][y][x] = gO[80 - e][y + ey][x + ex]
    auto g_load = [&](int eyi, int exi) -> unsigned short {
        const int e = eyi * kND + exi;
        const int d = side ? kND * kND - 1 - e : e;
        const int yy = side ? y + eyi - kD : y, xx = side ? gx + exi - kD : gx;
        const bool rok = yy >= 0 && yy < H && y < H;                         // wave-uniform
        const int soff = __builtin_amdgcn_readfirstlane(rok ? (d * plane + yy * W) * 2 : 0);
        const int voff = (rok && xx >= 0 && xx < W) ? xx * 2 : kDead;
        return __builtin_amdgcn_raw_buffer_load_b16(rsrc_go, voff, soff, 0);
    };
    // all 81 values of this lane's pixel, in flight during the window copy: the displacement
    // loop below then never waits on global memory
    unsigned short gv[kND][kND];
#pragma unroll
    for (int j = 0; j < kND; ++j)
#pragma unroll
        for (int i = 0; i < kND; ++i) gv[j][i] = g_load(j, i);

    // ---- the source window of the slice's channels -> LDS; zero the A rows ----
    {
        constexpr int UNITS = K::CS * K::WR * K::UPR, PER = UNITS / K::THREADS, BATCH = 30;
        static_assert(PER % BATCH == 0, "window copy batches");
#pragma unroll 1
        for (int i0 = 0; i0 < PER; i0 += BATCH) {
            u2v v[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int u = tid + K::THREADS * (i0 + i);
                const int ch = u / (K::WR * K::UPR), rem = u % (K::WR * K::UPR);
                const int row = rem / K::UPR, un = rem % K::UPR;
                const int sy = y0 - kD + row, sx = x0 - kD + 4 * un;
                const bool ok = un < (K::TW + 2 * kD) / 4 && sy >= 0 && sy < H && sx >= 0 && sx < W &&
                                c_begin + ch < C;
                v[i] = __builtin_amdgcn_raw_buffer_load_b64(
                    rsrc_src, ok ? ((c_begin + ch) * plane + sy * W + sx) * 2 : kDead, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int u = tid + K::THREADS * (i0 + i);
                const int ch = u / (K::WR * K::UPR), rem = u % (K::WR * K::UPR);
                *reinterpret_cast<u2v *>(win + ch * K::CSTR + rem * 4) = v[i];
            }
        }
        unsigned short *mine = arow + (wave * K::TW + lane) * K::AROW;
#pragma unroll
        for (int i = 0; i < K::AROW / 8; ++i) *reinterpret_cast<u4v *>(mine + 8 * i) = u4v{0, 0, 0, 0};
    }
    __syncthreads();

    // ---- 9 vertical displacements x (4 segments x CB channel blocks) MFMAs ----
    const int p = lane & 15, kg = lane >> 4;
    f4v acc[K::NSEG][K::CB];
#pragma unroll
    for (int s = 0; s < K::NSEG; ++s)
#pragma unroll
        for (int cb = 0; cb < K::CB; ++cb) acc[s][cb] = f4v{0.f, 0.f, 0.f, 0.f};
    unsigned short *my_arow = arow + (wave * K::TW + lane) * K::AROW + p;   // band slots p .. p + 8
    const unsigned short *a_rd = arow + (wave * K::TW + p) * K::AROW + 8 * (kg < 3 ? kg : 0);
    const unsigned short *b_rd = win + p * K::CSTR + wave * K::WC + 8 * kg;
#pragma unroll
    for (int eyi = 0; eyi < kND; ++eyi) {
        // this lane's pixel: the nine g values of the row into its band slots
#pragma unroll
        for (int i = 0; i < kND; ++i) my_arow[i] = gv[eyi][i];
        // (same wave wrote the rows it reads: program order + the compiler's lgkmcnt wait)
        u4v a[K::NSEG];
#pragma unroll
        for (int s = 0; s < K::NSEG; ++s) {
            a[s] = *reinterpret_cast<const u4v *>(a_rd + s * 16 * K::AROW);
            if (kg == 3) a[s] = u4v{0, 0, 0, 0};   // k = 24..31: outside every band
        }
#pragma unroll
        for (int cb = 0; cb < K::CB; ++cb)
#pragma unroll
            for (int s = 0; s < K::NSEG; ++s) {
                const u4v bv = *reinterpret_cast<const u4v *>(b_rd + cb * 16 * K::CSTR + eyi * K::WC + 16 * s);
                acc[s][cb] = Mma<T>::run(a[s], bv, acc[s][cb]);
            }
    }

    // ---- D[pixel][channel] -> gradInput[c][y][x]: 4 consecutive pixels of one channel per lane ----
    if (y >= H) return;
    const float inv_nelems = 1.0f / static_cast<float>(C);
#pragma unroll
    for (int cb = 0; cb < K::CB; ++cb) {
        const int c = c_begin + cb * 16 + p;
#pragma unroll
        for (int s = 0; s < K::NSEG; ++s) {
            const int x = x0 + 16 * s + 4 * kg;
            const f4v d = acc[s][cb];
            const u2v o = {Mma<T>::pack2(d[0] * inv_nelems, d[1] * inv_nelems),
                           Mma<T>::pack2(d[2] * inv_nelems, d[3] * inv_nelems)};
            __builtin_amdgcn_raw_buffer_store_b64(o, rsrc_dst, (c < C && x < W) ? (c * plane + y * W + x) * 2 : kDead,
                                                  0, 0);
        }
    }
#endif
}

template <typename T>
int launch(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
           const CorrGeom &g, hipStream_t s) {
    using K = BwdMfmaCfg;
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int nslice = (g.C + K::CS - 1) / K::CS;
    const int64_t blocks = static_cast<int64_t>(g.B) * tiles_x * tiles_y * nslice * 2;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    static std::atomic<int> lds_set{0};
    if (!lds_set.load(std::memory_order_acquire)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(corr_bwd_d4_mfma_kernel<T>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(K::LDS_BYTES));
        if (e != hipSuccess) return static_cast<int>(e);
        lds_set.store(1, std::memory_order_release);
    }
    note_kernel(1, "corr_bwd_d4_mfma_4x64");
    hipLaunchKernelGGL((corr_bwd_d4_mfma_kernel<T>), dim3(static_cast<unsigned>(blocks)), dim3(K::THREADS),
                       K::LDS_BYTES, s, static_cast<const T *>(in1), static_cast<const T *>(in2),
                       static_cast<const T *>(gout), static_cast<T *>(gin1), static_cast<T *>(gin2), g.C,
                       g.H, g.W, tiles_x, tiles_y, nslice);
    return launch_status();
}

}  // namespace

// 16-bit storage, pad = d = 4, k = 1, s1 = s2 = 1, W % 4 == 0, 8-byte aligned tensors, a batch
// item below 2^30 bytes (32-bit buffer offsets): checked by the caller (corr_d4.hip)
int corr_mfma_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
                       const CorrGeom &g, int dtype, hipStream_t s) {
    switch (dtype) {
        case CERB_F16: return launch<__half>(in1, in2, gout, gin1, gin2, g, s);
        case CERB_BF16: return launch<hip_bfloat16>(in1, in2, gout, gin1, gin2, g, s);
        default: return CERB_EUNSUPPORTED;
    }
}

}  // namespace cerb
