// corr_d4.hip -- tuned CDNA4 (gfx950) kernels for the one correlation
// configuration every CerberusNet model uses: pad = max_displacement = 4,
// kernel 1, stride1 = stride2 = 1 (pwcnet_sfd.py:131-133), fp32.
//
//   out[n][(dy+4)*9+(dx+4)][y][x] = 1/C * sum_c x1[n][c][y][x] * x2[n][c][y+dy][x+dx]
//   gI1[n][c][y][x] = 1/C * sum_d gO[n][d][y][x]       * x2[n][c][y+dy][x+dx]
//   gI2[n][c][y][x] = 1/C * sum_d gO[n][d][y-dy][x-dx] * x1[n][c][y-dy][x-dx]
//
// What the reference does (correlation_cuda_kernel.cu) and what replaces it:
//   * NHWC zero-padded copies of both inputs in forward AND again in backward
//     (.cu:13-27, 266-293, 345-376)          -> none: NCHW is read directly,
//     W-contiguous 16-byte loads, the halo is predicated to zero.
//   * one 32-thread block per output pixel, 81 barrier-separated serial
//     lane-0 reductions (.cu:55-93)          -> a workgroup owns a spatial
//     tile; the (tile + 2d)^2 window of x2 is staged in LDS once per channel
//     chunk (double buffered, next chunk prefetched into registers while the
//     current one is consumed); each lane owns a 4-pixel strip and slides a
//     12-float x2 row segment across the 9 horizontal displacements in
//     registers (36 FMAs per 4 LDS reads); the 9 vertical displacements are
//     the 9 wavefronts of the workgroup.
//   * channel reduction by shared memory + one lane (.cu:78-83) -> channels
//     are split over S lane groups of a wavefront where the map is small and
//     reduced with wave64 shuffles (ds_bpermute), no LDS round trip.
//   * backward: 2*B sequential launches of H*W*C one-warp blocks with
//     H*W-strided gradOutput reads (.cu:386-426) -> one launch; a lane keeps
//     all 81 gradOutput values of its 2 pixels in registers (the flipped /
//     shifted gather for gI2 is done once per tile) and streams channels
//     through the LDS window: 162 FMAs per 45 ds_read_b64, no atomics, no
//     cross-lane reduction; channels are additionally split over workgroups.
//
// No MFMA: this is a gather-reduce (a banded product wastes >7x the flops as a
// dense GEMM and fp32 MFMA runs at the vector rate anyway).
#include <atomic>
#include <type_traits>

#include "common.h"

namespace cerb {
namespace {

constexpr int kD = 4;            // max displacement
constexpr int kND = 2 * kD + 1;  // 9 displacements per axis
constexpr int kP = 4;            // pixels per lane in forward (one float4)

__host__ __device__ constexpr int pad_to_residue(int x, int res) {
    return x + ((res - x % 64) + 64) % 64;
}

// ============================================================================
// forward
// ============================================================================
// S    channel groups per wavefront (lanes cg*NS .. cg*NS+NS-1 own channels
//      [cg*C/S, (cg+1)*C/S)), reduced with shuffles at the end
// TSX  4-pixel strips per tile row        RB  row blocks per lane
// CC   channels per group per LDS chunk   ROT strip rotation per row (bank fix)
// RS/RS1 row strides (floats) of the x2 / x1 LDS tiles, PRES plane-stride
// residue mod 64 -- chosen with tools/lds_conflicts.py so that every
// ds_read_b128 of the main loop is conflict free.
template <int S_, int TSX_, int RB_, int CC_, int ROT_, int RS_, int RS1_, int PRES_, int WPS_ = 3,
          int NSET_ = 1>
struct FwdCfg {
    static constexpr int S = S_, TSX = TSX_, RB = RB_, CC = CC_, ROT = ROT_;
    static constexpr int NS = 64 / S;        // strips per channel group
    static constexpr int NR = NS / TSX;      // tile rows per row block
    static constexpr int TH = NR * RB;       // tile height
    static constexpr int TW = TSX * kP;      // tile width
    static constexpr int HR = TH + 2 * kD;   // halo rows
    static constexpr int HW4 = TSX + 2;      // halo width in float4
    static constexpr int RS = RS_, RS1 = RS1_;
    static constexpr int PS = pad_to_residue(HR * RS, PRES_);
    static constexpr int PS1 = pad_to_residue(TH * RS1, PRES_);
    static constexpr int NCH = S * CC;       // channel planes per chunk
    static constexpr int THREADS = 64 * kND;
    static constexpr int N2 = NCH * HR * HW4;  // float4 slots of the x2 halo chunk
    static constexpr int N1 = NCH * TH * TSX;  // float4 slots of the x1 chunk
    static constexpr int NSLOT = (N2 + N1 + THREADS - 1) / THREADS;
    static constexpr int BUF = NCH * (PS + PS1);  // floats per LDS buffer
    static constexpr size_t LDS_BYTES = 2 * sizeof(float) * BUF;
    // waves per SIMD requested from the register allocator: two 9-wave workgroups per CU
    // need 5 (<= 96 VGPRs); the 2-row-block variant keeps 72 accumulators and asks for 3
    static constexpr int WPS = WPS_;
    static constexpr int NSET = NSET_;  // chunks of global loads kept in flight (register sets)
    static_assert(NS % TSX == 0, "strips must tile rows");
    static_assert(RS % 4 == 0 && RS1 % 4 == 0 && PS % 4 == 0 && PS1 % 4 == 0, "16B alignment");
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// ---- global-memory element types ------------------------------------------------------
// fp16 / bf16 are STORAGE formats: values are widened when staged into LDS / registers,
// every product and sum is fp32 (the reference accumulates fp16 in fp16),
// results are rounded once on the way out.  LDS always holds fp32.
template <typename T> struct Gmem;
template <> struct Gmem<float> {
    static __device__ __forceinline__ float load1(const float *p) { return *p; }
    static __device__ __forceinline__ float2 load2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
    static __device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void store1(float *p, float v) { *p = v; }
    static __device__ __forceinline__ void store2(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
    static __device__ __forceinline__ void store4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    // streaming store (written once, not re-read by the kernel): leaves L2 to the halo lines
    static __device__ __forceinline__ void stream2(float *p, float a, float b) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(f2v{a, b}, reinterpret_cast<f2v *>(p));
    }
};
template <> struct Gmem<__half> {
    static __device__ __forceinline__ void stream2(__half *p, float a, float b) { store2(p, a, b); }
    static __device__ __forceinline__ float load1(const __half *p) { return __half2float(*p); }
    static __device__ __forceinline__ float2 load2(const __half *p) { return __half22float2(*reinterpret_cast<const __half2 *>(p)); }
    static __device__ __forceinline__ float4 load4(const __half *p) {
        const uint2 raw = *reinterpret_cast<const uint2 *>(p);
        const float2 lo = __half22float2(*reinterpret_cast<const __half2 *>(&raw.x));
        const float2 hi = __half22float2(*reinterpret_cast<const __half2 *>(&raw.y));
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    static __device__ __forceinline__ void store1(__half *p, float v) { *p = __float2half(v); }
    static __device__ __forceinline__ void store2(__half *p, float a, float b) { *reinterpret_cast<__half2 *>(p) = __floats2half2_rn(a, b); }
    static __device__ __forceinline__ void store4(__half *p, float4 v) {
        uint2 raw;
        *reinterpret_cast<__half2 *>(&raw.x) = __floats2half2_rn(v.x, v.y);
        *reinterpret_cast<__half2 *>(&raw.y) = __floats2half2_rn(v.z, v.w);
        *reinterpret_cast<uint2 *>(p) = raw;
    }
};
template <> struct Gmem<hip_bfloat16> {
    static __device__ __forceinline__ void stream2(hip_bfloat16 *p, float a, float b) { store2(p, a, b); }
    static __device__ __forceinline__ float widen(unsigned short b) { return __uint_as_float(static_cast<unsigned int>(b) << 16); }
    static __device__ __forceinline__ unsigned short narrow(float v) {
        const hip_bfloat16 h(v);  // round to nearest even, NaN stays NaN
        unsigned short b;
        __builtin_memcpy(&b, &h, 2);
        return b;
    }
    static __device__ __forceinline__ float load1(const hip_bfloat16 *p) { return widen(*reinterpret_cast<const unsigned short *>(p)); }
    static __device__ __forceinline__ float2 load2(const hip_bfloat16 *p) {
        const unsigned int raw = *reinterpret_cast<const unsigned int *>(p);
        return make_float2(__uint_as_float(raw << 16), __uint_as_float(raw & 0xFFFF0000u));
    }
    static __device__ __forceinline__ float4 load4(const hip_bfloat16 *p) {
        const uint2 raw = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xFFFF0000u),
                           __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xFFFF0000u));
    }
    static __device__ __forceinline__ void store1(hip_bfloat16 *p, float v) { *reinterpret_cast<unsigned short *>(p) = narrow(v); }
    // two floats -> packed bf16 pair: gfx950's v_cvt_pk_bf16_f32 (round to nearest even, NaN
    // preserved), one instruction instead of the ~10 of the software rounding per value
    static __device__ __forceinline__ unsigned int narrow2(float a, float b) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned int, __builtin_convertvector(f2v{a, b}, bf2v));
    }
    static __device__ __forceinline__ void store2(hip_bfloat16 *p, float a, float b) {
        *reinterpret_cast<unsigned int *>(p) = narrow2(a, b);
    }
    static __device__ __forceinline__ void store4(hip_bfloat16 *p, float4 v) {
        uint2 raw;
        raw.x = narrow2(v.x, v.y);
        raw.y = narrow2(v.z, v.w);
        *reinterpret_cast<uint2 *>(p) = raw;
    }
};

typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v pkfma(float2v a, float2v b, float2v c) {
    return __builtin_elementwise_fma(a, b, c);  // v_pk_fma_f32
}
__device__ __forceinline__ float2v ld2v(const float *p) { return *reinterpret_cast<const float2v *>(p); }
// volatile: keeps hipcc's load/store optimizer from fusing neighbouring 8-byte LDS reads
// into ds_read2_b64, which runs at HALF the LDS rate of ds_read_b64 on gfx950
// (MI355X_MICROARCH.md LDS table: 8 vs 2 cycles per wave-instruction for 2x/1x 512 B)
typedef const volatile __attribute__((address_space(3))) float2v *lds_f2_volatile_ptr;
__device__ __forceinline__ float2v ld2v_nomerge(const float *p) {
    return *(lds_f2_volatile_ptr)(p);  // explicit LDS address space: stays a ds_read_b64
}

// 16 bytes of zeros in device memory: the source of every halo / padding slot, so
// that staging loads are UNCONDITIONAL.  (A load under a branch makes hipcc's
// waitcnt pass fall back to vmcnt(0) at the next use, which serialises the
// prefetch pipeline -- seen in the ISA of the first version.)
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

template <typename K, typename T, bool VEC>
__global__ __launch_bounds__(K::THREADS, K::WPS) void corr_fwd_d4_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, T *__restrict__ out, int C,
    int H, int W, int tiles_x, int tiles_y, float slope, int64_t out_bstride, int dbg) {
    // Timing-ablation builds only (-DCERB_ABLATE; results are WRONG when the mask is set):
    // 1 = store only displacement 0, 2 = load only the first chunk, 4 = skip the FMAs,
    // 8 = no LDS commit, 16 = no barrier, 32 = no epilogue, 64 = return at once.
    // The product build compiles the mask to the constant 0.
#ifndef CERB_ABLATE
    dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = K::S, TSX = K::TSX, RB = K::RB, CC = K::CC, NSET = K::NSET;
    if (dbg & 64) return;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;  // vertical displacement index dy + 4
    const int lane = tid & 63;
    const int cg = lane / K::NS;
    const int si = lane % K::NS;
    const int r = si / TSX;
    const int sx = (K::ROT == 0) ? si % TSX : (si % TSX + TSX - (K::ROT * r) % TSX) % TSX;

    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx * K::TW, y0 = ty * K::TH;

    const int plane = H * W;
    const int Cg = C / S;  // channels per group (launcher guarantees C % S == 0)
    const int nchunks = (Cg + CC - 1) / CC;
    const T *x1b = x1 + static_cast<int64_t>(b) * C * plane;
    const T *x2b = x2 + static_cast<int64_t>(b) * C * plane;

    // ---- per-slot staging descriptors (fixed for the whole kernel) ----
    int goff[K::NSLOT];           // element offset (+4) of chunk 0 in the batch item, <0: zeros
    int loff[K::NSLOT];           // LDS float offset inside a buffer, <0: slot unused
    int gx0[K::NSLOT];            // first x of the slot (scalar path bounds)
    int chi[K::NSLOT];            // channel index inside the chunk
#pragma unroll
    for (int j = 0; j < K::NSLOT; ++j) {
        int id = tid + j * K::THREADS;
        goff[j] = -1; loff[j] = -1; gx0[j] = 0; chi[j] = 0;
        if (id < K::N2) {
            const int pl = id / (K::HR * K::HW4);
            const int rem = id % (K::HR * K::HW4);
            const int row = rem / K::HW4, c4 = rem % K::HW4;
            const int i = pl / S, g = pl % S;
            const int gy = y0 - kD + row, gx = x0 - kD + 4 * c4;
            loff[j] = pl * K::PS + row * K::RS + 4 * c4;
            gx0[j] = gx; chi[j] = i;
            const bool in = gy >= 0 && gy < H && (VEC ? (gx >= 0 && gx < W) : (gx > -4 && gx < W));
            if (in) goff[j] = (g * Cg + i) * plane + gy * W + gx + 4;  // +4: scalar path gx > -4
        } else if (id < K::N2 + K::N1) {
            id -= K::N2;
            const int pl = id / (K::TH * TSX);
            const int rem = id % (K::TH * TSX);
            const int row = rem / TSX, c4 = rem % TSX;
            const int i = pl / S, g = pl % S;
            const int gy = y0 + row, gx = x0 + 4 * c4;
            loff[j] = K::NCH * K::PS + pl * K::PS1 + row * K::RS1 + 4 * c4;
            gx0[j] = gx; chi[j] = i;
            const bool in = gy < H && gx < W;
            if (in) goff[j] = (g * Cg + i) * plane + gy * W + gx + 4;
        }
    }
    float4 stage[NSET][K::NSLOT];

    auto prefetch = [&](int k, float4(&st)[K::NSLOT]) {
#pragma unroll
        for (int j = 0; j < K::NSLOT; ++j) {
            const bool live = goff[j] >= 0 && (k * CC + chi[j] < Cg);
            const T *base = (tid + j * K::THREADS < K::N2) ? x2b : x1b;
            if (VEC) {
                // branch-free: dead slots (halo outside the image, channels past the
                // end, chunks past the last one) read the zero block instead
                const T *src = live ? base + (goff[j] - 4) + k * CC * plane
                                    : reinterpret_cast<const T *>(g_zero16);
                st[j] = Gmem<T>::load4(src);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) {
                    const T *src = base + (goff[j] - 4) + k * CC * plane;
                    const int gx = gx0[j];
                    if (gx >= 0 && gx < W) v.x = Gmem<T>::load1(src);
                    if (gx + 1 >= 0 && gx + 1 < W) v.y = Gmem<T>::load1(src + 1);
                    if (gx + 2 >= 0 && gx + 2 < W) v.z = Gmem<T>::load1(src + 2);
                    if (gx + 3 >= 0 && gx + 3 < W) v.w = Gmem<T>::load1(src + 3);
                }
                st[j] = v;
            }
        }
    };
    auto commit = [&](float *buf, const float4(&st)[K::NSLOT]) {
#pragma unroll
        for (int j = 0; j < K::NSLOT; ++j)
            if (loff[j] >= 0) st4(buf + loff[j], st[j]);
    };

    // accumulators: for pixel p the 9 horizontal displacements are kept as 4
    // register pairs + 1 single, paired so that every packed FMA reads an ALIGNED
    // pair of the x2 row segment: p even -> pairs (0,1)(2,3)(4,5)(6,7), single 8;
    // p odd -> single 0, pairs (1,2)(3,4)(5,6)(7,8).
    float2v accp[RB][kP][4];
    float accs[RB][kP];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            accs[rb][p] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) accp[rb][p][j] = float2v{0.f, 0.f};
        }

    auto compute = [&](const float *buf) {
        const float *X2 = buf + cg * K::PS + (r + wave) * K::RS + 4 * sx;
        const float *X1 = buf + K::NCH * K::PS + cg * K::PS1 + r * K::RS1 + 4 * sx;
#pragma unroll
        for (int i = 0; i < CC; ++i) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float4 a = ld4(X1 + i * S * K::PS1 + rb * K::NR * K::RS1);
                const float *bp = X2 + i * S * K::PS + rb * K::NR * K::RS;
                const float4 b0 = ld4(bp), b1 = ld4(bp + 4), b2 = ld4(bp + 8);
                const float av[4] = {a.x, a.y, a.z, a.w};
                const float2v bw[6] = {float2v{b0.x, b0.y}, float2v{b0.z, b0.w},
                                       float2v{b1.x, b1.y}, float2v{b1.z, b1.w},
                                       float2v{b2.x, b2.y}, float2v{b2.z, b2.w}};
#pragma unroll
                for (int p = 0; p < kP; ++p) {
                    const int off = p & 1;
                    const float2v aa = float2v{av[p], av[p]};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        accp[rb][p][j] = pkfma(aa, bw[(p + off) / 2 + j], accp[rb][p][j]);
                    const float bs = off ? bw[(p - 1) / 2].y : bw[(p + 8) / 2].x;
                    accs[rb][p] = fmaf(av[p], bs, accs[rb][p]);
                }
            }
        }
    };

    // ---- software pipeline: NSET chunks of global loads in flight ----
#pragma unroll
    for (int s = 0; s < NSET; ++s) prefetch(s, stage[s]);
    for (int k0 = 0; k0 < nchunks; k0 += NSET) {
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
            const int k = k0 + s;
            if (k < nchunks) {
                float *buf = smem + (k & 1) * K::BUF;
                if (!(dbg & 8)) commit(buf, stage[s]);
                if (!(dbg & 16)) __syncthreads();
                if (!(dbg & 2)) prefetch(k + NSET, stage[s]);  // unconditional (zeros past the end)
                if (!(dbg & 4)) compute(buf);
            }
        }
    }

    if (dbg & 32) {  // ablation: no epilogue
        if (tid == 0) Gmem<T>::store1(out + blockIdx.x, accs[0][0] + stage[0][0].x);
        return;
    }
    // ---- unpack to acc[rb][d][p] ----
    float acc[RB][kND][kP];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            const int off = p & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[rb][2 * j + off][p] = accp[rb][p][j].x;
                acc[rb][2 * j + off + 1][p] = accp[rb][p][j].y;
            }
            acc[rb][off ? 0 : 8][p] = accs[rb][p];
        }

    // ---- channel-group reduction with wave64 shuffles ----
    if (S > 1) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int d = 0; d < kND; ++d)
#pragma unroll
                for (int p = 0; p < kP; ++p) {
                    float v = acc[rb][d][p];
#pragma unroll
                    for (int m = K::NS; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
                    acc[rb][d][p] = v;
                }
    }

    // ---- epilogue: 1/C, fused LeakyReLU, coalesced stores ----
    if (cg != 0) return;
    // sum / nelems as in the reference (.cu:85,91); a reciprocal multiply is exact
    // whenever C is a power of two (every CerberusNet level) and within 1 ulp otherwise
    const float inv_nelems = 1.0f / static_cast<float>(C);
    const int64_t obs = out_bstride ? out_bstride : static_cast<int64_t>(kND * kND) * plane;
    T *ob = out + b * obs + static_cast<int64_t>(wave * kND) * plane;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int y = y0 + r + rb * K::NR;
        const int x = x0 + 4 * sx;
        if (y >= H || x >= W) continue;
#pragma unroll
        for (int d = 0; d < kND; ++d) {
            if ((dbg & 1) && d) continue;
            float v[4];
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                const float q = acc[rb][d][p] * inv_nelems;
                v[p] = q > 0.f ? q : q * slope;
            }
            T *dst = ob + static_cast<int64_t>(d) * plane + y * W + x;
            if (VEC) {
                if constexpr (sizeof(T) == 4) {
                    // streamed once, never re-read by this kernel: keep it out of the way of
                    // the x2 halo lines that neighbouring tiles want to find in L2
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(f4v{v[0], v[1], v[2], v[3]},
                                                reinterpret_cast<f4v *>(dst));
                } else {
                    Gmem<T>::store4(dst, make_float4(v[0], v[1], v[2], v[3]));
                }
            } else {
#pragma unroll
                for (int p = 0; p < kP; ++p)
                    if (x + p < W) Gmem<T>::store1(dst + p, v[p]);
            }
        }
    }
}

// ============================================================================
// forward, LDS-DMA variant (fp32, vector path only)
// ============================================================================
// Same tiles, lane mapping and FMA loop as corr_fwd_d4_kernel<S,TSX,RB=1>, different data
// movement: a TENTH wavefront is a dedicated loader.  It streams channel chunks into a ring
// of NB LDS buffers with buffer_load ... lds (LDS-DMA: no VGPR round trip, no ds_write), NB-1
// chunks ahead, and only ever executes {DMA issue, s_waitcnt vmcnt, s_barrier}; the nine
// compute wavefronts only execute {s_barrier, ds_read, FMA}.  hipcc places a conservative
// vmcnt(0) before every ds_read that follows an LDS-DMA in the SAME wave (tools/ubench/
// glds_test.hip) -- with the roles split across waves that wait is free: the compute waves
// have no vector-memory operation in flight.
// LDS image of a chunk (plane pl = i*S + g: channel i of the chunk, lane group g):
//   x2 windows: NCH planes x HR rows x HW4 16-byte slots, row stride RS = 4*HW4 floats, plane
//               stride PS (residue as in the staged kernels: conflict-free ds_read_b128);
//               one DMA instruction = RPI whole rows of one plane.
//   x1 tiles  : per chunk channel i ONE instruction of 64 slots: lane L stores the strip it
//               will itself read, at float offset 4*L -- linear, conflict free; the strip
//               rotation of the x2 side is applied to the DMA source address.
template <int S_, int TSX_, int CC_, int NB_, int ROT_, int PRES_>
struct FwdDmaCfg {
    static constexpr int S = S_, TSX = TSX_, CC = CC_, NB = NB_, ROT = ROT_;
    static constexpr int NS = 64 / S;                 // strips per channel group
    static constexpr int TH = NS / TSX, TW = TSX * kP;
    static constexpr int HR = TH + 2 * kD, HW4 = TSX + 2, RS = HW4 * 4;
    static constexpr int PS = pad_to_residue(HR * RS, PRES_);
    static constexpr int NCH = S * CC;                // channel planes per chunk
    static constexpr int RPI = 64 / HW4;              // window rows per DMA instruction
    static constexpr int NI2 = (HR + RPI - 1) / RPI;  // DMA instructions per x2 plane
    static constexpr int NINST = NCH * NI2 + CC;      // DMA wave-instructions per chunk
    static constexpr int BUF = NCH * PS + CC * 256;   // floats per ring buffer
    static constexpr int THREADS = 64 * (kND + 1);
    static constexpr size_t LDS_BYTES = sizeof(float) * NB * BUF;
    static_assert(NS % TSX == 0, "strips must tile rows");
    static_assert(NINST * (NB - 2) <= 63, "vmcnt is 6 bits");
    static_assert(NB >= 3 && NB <= 4, "ring depth");
    static_assert(PS % 4 == 0, "16B alignment");
};

typedef __attribute__((address_space(3))) void *lds_void_ptr;
typedef const __attribute__((address_space(1))) void *gbl_void_ptr;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename K>
__global__ __launch_bounds__(K::THREADS, K::S == 1 ? 5 : 3) void corr_fwd_d4_dma_kernel(
    const float *__restrict__ x1, const float *__restrict__ x2, float *__restrict__ out, int C,
    int H, int W, int tiles_x, int tiles_y, float slope, int64_t out_bstride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = K::S, TSX = K::TSX, CC = K::CC, NB = K::NB;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // lane -> (channel group, tile row, strip); the loader uses the same mapping for x1
    const int cg = lane / K::NS;
    const int si = lane % K::NS;
    const int r = si / TSX;
    const int sx = (K::ROT == 0) ? si % TSX : (si % TSX + TSX - (K::ROT * r) % TSX) % TSX;

    int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int tx = __builtin_amdgcn_readfirstlane(bid % tiles_x); bid /= tiles_x;
    const int ty = __builtin_amdgcn_readfirstlane(bid % tiles_y);
    const int b = __builtin_amdgcn_readfirstlane(bid / tiles_y);
    const int x0 = tx * K::TW, y0 = ty * K::TH;
    const int plane = H * W;
    const int Cg = C / S;  // channels per group (launcher guarantees C % S == 0)
    const int nchunks = (Cg + CC - 1) / CC;

#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only
    if (wave == kND) {
        // ------------------------------ loader wavefront ------------------------------
        // Buffer-resource DMA: an offset at or beyond num_records reads zeros, so padding and
        // the channel tail need no branch and no zero block; the plane / chunk advance goes
        // into the scalar offset, which leaves NI2 + 1 per-lane byte offsets for the whole
        // kernel (the row groups of an x2 window, and the lane's own x1 strip).
        constexpr int kDead = static_cast<int>(0x80000000u);
        const int item_bytes = C * plane * 4;
        const __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(x1 + static_cast<int64_t>(b) * C * plane, item_bytes);
        const __amdgpu_buffer_rsrc_t r2 = uniform_rsrc(x2 + static_cast<int64_t>(b) * C * plane, item_bytes);
        const bool x2lane = lane < K::RPI * K::HW4;
        const int lrow = lane / K::HW4, c4 = lane % K::HW4;
        const int gx2 = x0 - kD + 4 * c4;
        const bool xok = gx2 >= 0 && gx2 < W;
        int v2[K::NI2];
#pragma unroll
        for (int rg = 0; rg < K::NI2; ++rg) {
            const int row = rg * K::RPI + lrow;
            const int gy = y0 - kD + row;
            v2[rg] = (xok && row < K::HR && gy >= 0 && gy < H) ? (gy * W + gx2) * 4 : kDead;
        }
        const int gy1 = y0 + r, gx1 = x0 + 4 * sx;
        // x1: the lane's own strip of its own group's channel (group offset in the lane part)
        const int v1 = (gy1 < H && gx1 < W) ? (cg * Cg * plane + gy1 * W + gx1) * 4 : kDead;
        auto issue = [&](int k) {
            float *buf = smem + (k % NB) * K::BUF;
            // scalar byte offsets of the chunk's planes, fixed before the (divergent) lane
            // mask below so that they stay in SGPRs
            int soff[K::NCH];
            bool chok[CC];
#pragma unroll
            for (int i = 0; i < CC; ++i) {
                chok[i] = k * CC + i < Cg;
#pragma unroll
                for (int g = 0; g < S; ++g)
                    soff[i * S + g] = __builtin_amdgcn_readfirstlane((g * Cg + k * CC + i) * plane * 4);
            }
            if (x2lane) {
#pragma unroll
                for (int pl = 0; pl < K::NCH; ++pl)
#pragma unroll
                    for (int rg = 0; rg < K::NI2; ++rg)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            r2, (lds_void_ptr)(buf + pl * K::PS + rg * (K::RPI * K::RS)), 16,
                            chok[pl / S] ? v2[rg] : kDead, soff[pl], 0, 0);
            }
#pragma unroll
            for (int i = 0; i < CC; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    r1, (lds_void_ptr)(buf + K::NCH * K::PS + i * 256), 16, chok[i] ? v1 : kDead,
                    soff[i * S], 0, 0);   // group 0's plane offset; the group part is in v1
        };
#pragma unroll
        for (int k = 0; k < NB - 1; ++k)
            if (k < nchunks) issue(k);
        for (int k = 0; k < nchunks; ++k) {
            // this wave's DMAs for chunk k have landed once at most the younger chunks remain
            const int younger = min(NB - 2, nchunks - 1 - k);
            if (younger >= 2 && NB >= 4) wait_vmcnt<(NB >= 4 ? 2 : 1) * K::NINST>();
            else if (younger == 1) wait_vmcnt<K::NINST>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // chunk k visible to the compute waves; k-1 consumed
            if (k + NB - 1 < nchunks) issue(k + NB - 1);  // into the buffer chunk k-1 used
        }
        return;
    }
#endif

    // -------------------------------- compute wavefronts --------------------------------
    float2v accp[kP][4];
    float accs[kP];
#pragma unroll
    for (int p = 0; p < kP; ++p) {
        accs[p] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) accp[p][j] = float2v{0.f, 0.f};
    }
    for (int k = 0; k < nchunks; ++k) {
        __builtin_amdgcn_s_barrier();
        const float *buf = smem + (k % NB) * K::BUF;
        const float *X2 = buf + cg * K::PS + (r + wave) * K::RS + 4 * sx;
        const float *X1 = buf + K::NCH * K::PS + 4 * lane;
#pragma unroll
        for (int i = 0; i < CC; ++i) {
            const float4 a = ld4(X1 + i * 256);
            const float *bp = X2 + i * S * K::PS;
            const float4 b0 = ld4(bp), b1 = ld4(bp + 4), b2 = ld4(bp + 8);
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float2v bw[6] = {float2v{b0.x, b0.y}, float2v{b0.z, b0.w}, float2v{b1.x, b1.y},
                                   float2v{b1.z, b1.w}, float2v{b2.x, b2.y}, float2v{b2.z, b2.w}};
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                const int off = p & 1;
                const float2v aa = float2v{av[p], av[p]};
#pragma unroll
                for (int j = 0; j < 4; ++j) accp[p][j] = pkfma(aa, bw[(p + off) / 2 + j], accp[p][j]);
                const float bs = off ? bw[(p - 1) / 2].y : bw[(p + 8) / 2].x;
                accs[p] = fmaf(av[p], bs, accs[p]);
            }
        }
        // all LDS reads of this chunk have returned before the next barrier releases the loader
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- unpack, channel-group reduction with wave64 shuffles ----
    float acc[kND][kP];
#pragma unroll
    for (int p = 0; p < kP; ++p) {
        const int off = p & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[2 * j + off][p] = accp[p][j].x;
            acc[2 * j + off + 1][p] = accp[p][j].y;
        }
        acc[off ? 0 : 8][p] = accs[p];
    }
    if (S > 1) {
#pragma unroll
        for (int d = 0; d < kND; ++d)
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                float v = acc[d][p];
#pragma unroll
                for (int m = K::NS; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
                acc[d][p] = v;
            }
    }
    if (cg != 0) return;
    const float inv_nelems = 1.0f / static_cast<float>(C);
    const int64_t obs = out_bstride ? out_bstride : static_cast<int64_t>(kND * kND) * plane;
    float *ob = out + b * obs + static_cast<int64_t>(wave * kND) * plane;
    const int y = y0 + r, x = x0 + 4 * sx;
    if (y >= H || x >= W) return;
#pragma unroll
    for (int d = 0; d < kND; ++d) {
        float v[4];
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            const float q = acc[d][p] * inv_nelems;
            v[p] = q > 0.f ? q : q * slope;
        }
        typedef float f4v __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4v{v[0], v[1], v[2], v[3]},
                                    reinterpret_cast<f4v *>(ob + static_cast<int64_t>(d) * plane + y * W + x));
    }
}

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
// 16-bit storage is only instantiated for the vector (aligned, W % 4 == 0) path; other
// shapes of those dtypes take the generic kernels.
template <typename K, typename T>
int launch_fwd(const char *name, const void *in1, const void *in2, void *outp, const CorrGeom &g,
               float slope, int64_t obs, bool vec, hipStream_t s) {
    const T *x1 = static_cast<const T *>(in1), *x2 = static_cast<const T *>(in2);
    T *out = static_cast<T *>(outp);
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int64_t blocks = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    int rc;
    static std::atomic<uint64_t> lds_v{0}, lds_s{0};
    const int dbg = debug_mask();
    if (vec) {
        note_kernel(0, name);
        if ((rc = ensure_lds(corr_fwd_d4_kernel<K, T, true>, K::LDS_BYTES, &lds_v))) return rc;
        hipLaunchKernelGGL((corr_fwd_d4_kernel<K, T, true>), dim3(static_cast<unsigned>(blocks)),
                           dim3(K::THREADS), K::LDS_BYTES, s, x1, x2, out, g.C, g.H, g.W, tiles_x,
                           tiles_y, slope, obs, dbg);
    } else if constexpr (sizeof(T) == 4) {
        note_kernel(0, name);
        if ((rc = ensure_lds(corr_fwd_d4_kernel<K, T, false>, K::LDS_BYTES, &lds_s))) return rc;
        hipLaunchKernelGGL((corr_fwd_d4_kernel<K, T, false>), dim3(static_cast<unsigned>(blocks)),
                           dim3(K::THREADS), K::LDS_BYTES, s, x1, x2, out, g.C, g.H, g.W, tiles_x,
                           tiles_y, slope, obs, dbg);
    } else {
        return CERB_EUNSUPPORTED;
    }
    return launch_status();
}

//                     S  TSX RB CC ROT RS  RS1 PRES WPS NSET
using FwdA2 = FwdCfg<1, 16, 2, 4, 2, 72, 72, 0, 3, 1>;   // 8x64 tile
using FwdA1 = FwdCfg<1, 16, 1, 8, 2, 72, 72, 0, 5, 1>;   // 4x64 tile, 8-channel chunks
using FwdB1 = FwdCfg<2, 16, 1, 4, 2, 72, 72, 0, 3, 2>;   // 2x64 tile, 2 channel groups
using FwdC1 = FwdCfg<4, 16, 1, 4, 0, 72, 64, 0, 3, 2>;   // 1x64 tile, 4 channel groups
using FwdD1 = FwdCfg<8, 8, 1, 4, 0, 40, 32, 32, 3, 1>;   // 1x32 tile, 8 channel groups
using FwdE1 = FwdCfg<16, 4, 1, 2, 0, 24, 16, 16, 3, 2>;  // 1x16 tile, 16 channel groups
using FwdA1b = FwdCfg<1, 16, 1, 4, 2, 72, 72, 0, 3, 1>;  // 4x64 tile, small chunks, 3 WGs/CU
using FwdA1c = FwdCfg<1, 16, 1, 4, 2, 72, 72, 0, 3, 2>;  // 4x64 tile, 2 chunks in flight

using FwdDma4 = FwdDmaCfg<1, 16, 4, 4, 2, 0>;    // 4x64 tile, 4-channel chunks, ring of 4
using FwdDmaB = FwdDmaCfg<2, 16, 4, 3, 2, 0>;    // 2x64 tile, 2 channel groups
using FwdDmaC = FwdDmaCfg<4, 16, 2, 3, 0, 0>;    // 1x64 tile, 4 channel groups
using FwdDmaD = FwdDmaCfg<8, 8, 2, 3, 0, 32>;    // 1x32 tile, 8 channel groups
using FwdDmaE = FwdDmaCfg<16, 4, 2, 3, 0, 16>;   // 1x16 tile, 16 channel groups

template <typename K>
int launch_fwd_dma(const char *name, const void *in1, const void *in2, void *outp,
                   const CorrGeom &g, float slope, int64_t obs, hipStream_t s) {
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int64_t blocks = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    static std::atomic<uint64_t> lds_done{0};
    int rc;
    if ((rc = ensure_lds(corr_fwd_d4_dma_kernel<K>, K::LDS_BYTES, &lds_done))) return rc;
    note_kernel(0, name);
    hipLaunchKernelGGL((corr_fwd_d4_dma_kernel<K>), dim3(static_cast<unsigned>(blocks)),
                       dim3(K::THREADS), K::LDS_BYTES, s, static_cast<const float *>(in1),
                       static_cast<const float *>(in2), static_cast<float *>(outp), g.C, g.H, g.W,
                       tiles_x, tiles_y, slope, obs);
    return launch_status();
}

template <typename K>
int64_t fwd_tiles(const CorrGeom &g) {
    return static_cast<int64_t>(g.B) * ((g.W + K::TW - 1) / K::TW) * ((g.H + K::TH - 1) / K::TH);
}

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

bool fast_config(const CorrGeom &g, int dtype) {
    return (dtype == CERB_F32 || dtype == CERB_F16 || dtype == CERB_BF16) && g.pad == kD &&
           g.maxd == kD && g.ksize == 1 && g.s1 == 1 && g.s2 == 1 &&
           static_cast<int64_t>(g.C) * g.H * g.W < (1ll << 30) &&
           static_cast<int64_t>(kND * kND) * g.H * g.W < (1ll << 30);  // 32-bit offsets
}

// The LDS-DMA kernels address a batch item through 32-bit buffer offsets and use 2^31 as the
// "out of range" offset: a batch item (and its 81-plane gradOutput) must stay below 2 GiB.
// Larger items keep the register-staged kernels (64-bit pointers, up to 2^30 elements).
bool dma_ok(const CorrGeom &g) {
    return static_cast<int64_t>(g.C) * g.H * g.W < (1ll << 29) &&
           static_cast<int64_t>(kND * kND) * g.H * g.W < (1ll << 29);
}

// a 4-element group must be naturally aligned: 16 B (fp32) or 8 B (16-bit storage)
bool aligned_group(const void *p, int dtype) {
    return (reinterpret_cast<uintptr_t>(p) & (dtype == CERB_F32 ? 15 : 7)) == 0;
}

template <typename T>
int fwd_dispatch(const void *x1, const void *x2, void *o, const CorrGeom &g, float slope,
                 int64_t obs, bool vec, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        // 16-bit storage, 16 < C <= 64: the matrix-core kernel (corr_mfma.hip; 14 forces it); variants 1-8 keep the VALU kernels
        const int v = option(OPT_CORR_FWD_VARIANT);
        if (vec && dma_ok(g) && g.C <= 64 && option(OPT_CORR_NO_MFMA) == 0 && (v == 14 || (v == 0 && g.C > 16)))   // <= 16 channels fill half an MFMA: no gain
            return corr_mfma_forward(x1, x2, o, g, slope, obs,
                                     std::is_same<T, __half>::value ? CERB_F16 : CERB_BF16, s);
    }
    {
        // coarse levels (W <= 64): independent waves, no loader, one barrier (corr_coarse.hip); 15 forces it, 16 keeps it off
        // (4 pairs: 7.0 vs 11.6 us at 256 x 16 x 32, 8.1 vs 10.8 us at 128 x 32 x 64; the tile kernels catch up
        // once a launch has more than ~2500 (row, displacement row) workgroups).  16-bit storage: the same kernel
        // with the loads widened (the matrix-core kernel above keeps 16 < C <= 64)
        const int v = option(OPT_CORR_FWD_VARIANT);
        const bool coarse_auto = g.W <= 64 && static_cast<int64_t>(g.B) * g.H * kND <= 2560;   // 8 pairs of 128 x 32 x 64: 12.5 vs 13.0 us
        if (vec && dma_ok(g) && (v == 15 || (v == 0 && coarse_auto))) {
            const int dt = sizeof(T) == 4 ? CERB_F32 : std::is_same<T, __half>::value ? CERB_F16 : CERB_BF16;
            const int rc = corr_coarse_forward(x1, x2, o, g, slope, obs, dt, s);
            if (rc != CERB_EUNSUPPORTED) return rc;
        }
    }
    switch (option(OPT_CORR_FWD_VARIANT)) {  // tuning / test hook
#ifdef CERB_EXPERIMENTS   // measured and rejected; never picked by the dispatcher (test builds only)
        case 1: return launch_fwd<FwdA2, T>("corr_fwd_d4_8x64", x1, x2, o, g, slope, obs, vec, s);
        case 2: return launch_fwd<FwdA1, T>("corr_fwd_d4_4x64", x1, x2, o, g, slope, obs, vec, s);
#endif
        case 3: if (g.C % 2 == 0) return launch_fwd<FwdB1, T>("corr_fwd_d4_2x64_s2", x1, x2, o, g, slope, obs, vec, s); break;
        case 4: if (g.C % 4 == 0) return launch_fwd<FwdC1, T>("corr_fwd_d4_1x64_s4", x1, x2, o, g, slope, obs, vec, s); break;
        case 5: if (g.C % 8 == 0) return launch_fwd<FwdD1, T>("corr_fwd_d4_1x32_s8", x1, x2, o, g, slope, obs, vec, s); break;
        case 6: if (g.C % 16 == 0) return launch_fwd<FwdE1, T>("corr_fwd_d4_1x16_s16", x1, x2, o, g, slope, obs, vec, s); break;
        case 7: return launch_fwd<FwdA1b, T>("corr_fwd_d4_4x64_cc4", x1, x2, o, g, slope, obs, vec, s);
#ifdef CERB_EXPERIMENTS
        case 8: return launch_fwd<FwdA1c, T>("corr_fwd_d4_4x64_cc4x2", x1, x2, o, g, slope, obs, vec, s);
#endif
        case 9:
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g))
                    return launch_fwd_dma<FwdDma4>("corr_fwd_d4_dma_4x64", x1, x2, o, g, slope, obs, s);
            }
            break;
        case 17:   // the persistent, cross-item pipelined forward (corr_fwd_pipe.hip)
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g)) {
                    const int rc = corr_fwd_pipe(x1, x2, o, g, slope, obs, s);
                    if (rc != CERB_EUNSUPPORTED) return rc;
                }
            }
            break;
        case 10: case 11: case 12: case 13:
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g)) {
                    const int v = option(OPT_CORR_FWD_VARIANT);
                    if (v == 10 && g.C % 2 == 0)
                        return launch_fwd_dma<FwdDmaB>("corr_fwd_d4_dma_2x64_s2", x1, x2, o, g, slope, obs, s);
                    if (v == 11 && g.C % 4 == 0)
                        return launch_fwd_dma<FwdDmaC>("corr_fwd_d4_dma_1x64_s4", x1, x2, o, g, slope, obs, s);
                    if (v == 12 && g.C % 8 == 0)
                        return launch_fwd_dma<FwdDmaD>("corr_fwd_d4_dma_1x32_s8", x1, x2, o, g, slope, obs, s);
                    if (v == 13 && g.C % 16 == 0)
                        return launch_fwd_dma<FwdDmaE>("corr_fwd_d4_dma_1x16_s16", x1, x2, o, g, slope, obs, s);
                }
            }
            break;
        default: break;
    }
    // Smallest channel split that still yields >= 256 workgroups (one per CU); the tile
    // sweep on MI355X (tools/tune_corr.py, profiles/) picked exactly this order.
    const int64_t want = 256;
    // fp32 vector path: the LDS-DMA kernels (loader wavefront + ring of LDS buffers); same
    // tiles, lane mapping and summation order as the register-staged ones -> identical bits.
    // Level 3: 19-21 vs 22 us; levels 0 / 1 / 2: 11.6 / 11.0 / 12.0 vs 12.7 / 13.2 / 13.3 us.
    const bool dma = sizeof(T) == 4 && vec && dma_ok(g);
    if (fwd_tiles<FwdA1b>(g) >= want || g.C % 2 != 0) {
        if constexpr (sizeof(T) == 4) {
            if (dma) return launch_fwd_dma<FwdDma4>("corr_fwd_d4_dma_4x64", x1, x2, o, g, slope, obs, s);
        }
        return launch_fwd<FwdA1b, T>("corr_fwd_d4_4x64_cc4", x1, x2, o, g, slope, obs, vec, s);
    }
    if (fwd_tiles<FwdB1>(g) >= want || g.C % 4 != 0) {
        if constexpr (sizeof(T) == 4) {
            if (dma) return launch_fwd_dma<FwdDmaB>("corr_fwd_d4_dma_2x64_s2", x1, x2, o, g, slope, obs, s);
        }
        return launch_fwd<FwdB1, T>("corr_fwd_d4_2x64_s2", x1, x2, o, g, slope, obs, vec, s);
    }
    if (fwd_tiles<FwdC1>(g) >= want || g.C % 8 != 0) {
        if constexpr (sizeof(T) == 4) {
            if (dma) return launch_fwd_dma<FwdDmaC>("corr_fwd_d4_dma_1x64_s4", x1, x2, o, g, slope, obs, s);
        }
        return launch_fwd<FwdC1, T>("corr_fwd_d4_1x64_s4", x1, x2, o, g, slope, obs, vec, s);
    }
    if (fwd_tiles<FwdD1>(g) >= want || g.C % 16 != 0) {
        if constexpr (sizeof(T) == 4) {
            if (dma) return launch_fwd_dma<FwdDmaD>("corr_fwd_d4_dma_1x32_s8", x1, x2, o, g, slope, obs, s);
        }
        return launch_fwd<FwdD1, T>("corr_fwd_d4_1x32_s8", x1, x2, o, g, slope, obs, vec, s);
    }
    if constexpr (sizeof(T) == 4) {
        if (dma) return launch_fwd_dma<FwdDmaE>("corr_fwd_d4_dma_1x16_s16", x1, x2, o, g, slope, obs, s);
    }
    return launch_fwd<FwdE1, T>("corr_fwd_d4_1x16_s16", x1, x2, o, g, slope, obs, vec, s);
}

template <typename T>
int bwd_dispatch(const void *x1, const void *x2, const void *go, void *g1, void *g2,
                 const CorrGeom &g, bool vec, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        // 16-bit storage: the matrix-core kernel (corr_mfma.hip); variants 1-3 keep the VALU kernels
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
        const int64_t coarse_wgs = static_cast<int64_t>(g.B) * 2 * g.H * (g.C / (g.W == 64 ? 16 : 32));
        if (vec && dma_ok(g) && (v == 14 || ((v == 0 || v == 13) && g.W <= 64 && coarse_wgs <= 4096))) {   // 13 = auto, minus the strip kernel on 64-wide maps
            const int rc = corr_coarse_backward(x1, x2, go, g1, g2, g, s);
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
        const int v = option(OPT_CORR_BWD_VARIANT);
        const int rows_per_wg = g.W == 256 ? 2 : g.W == 128 ? 4 : 8;
        const int64_t strip_wgs = static_cast<int64_t>(g.B) * (g.H / rows_per_wg) * (g.C / 32) * 2;
        const bool strip_auto = (g.W == 256 && strip_wgs >= 192) || (g.W == 128 && strip_wgs >= 192) ||
                                (g.W == 64 && strip_wgs >= 128 && v != 13);
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

int corr_d4_forward(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope,
                    int64_t obs, int dtype, hipStream_t s) {
    if (!fast_config(g, dtype)) return CERB_EUNSUPPORTED;
    const bool vec = g.W % 4 == 0 && aligned_group(in1, dtype) && aligned_group(in2, dtype) &&
                     aligned_group(out, dtype) && (obs % 4 == 0);
    switch (dtype) {
        case CERB_F32: return fwd_dispatch<float>(in1, in2, out, g, slope, obs, vec, s);
        case CERB_F16: return fwd_dispatch<__half>(in1, in2, out, g, slope, obs, vec, s);
        case CERB_BF16: return fwd_dispatch<hip_bfloat16>(in1, in2, out, g, slope, obs, vec, s);
        default: return CERB_EUNSUPPORTED;
    }
}

int corr_d4_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
                     const CorrGeom &g, int dtype, hipStream_t s) {
    if (!fast_config(g, dtype)) return CERB_EUNSUPPORTED;
    const bool vec = g.W % 4 == 0 && aligned_group(in1, dtype) && aligned_group(in2, dtype) &&
                     aligned_group(gout, dtype) && aligned_group(gin1, dtype) &&
                     aligned_group(gin2, dtype);
    switch (dtype) {
        case CERB_F32: return bwd_dispatch<float>(in1, in2, gout, gin1, gin2, g, vec, s);
        case CERB_F16: return bwd_dispatch<__half>(in1, in2, gout, gin1, gin2, g, vec, s);
        case CERB_BF16: return bwd_dispatch<hip_bfloat16>(in1, in2, gout, gin1, gin2, g, vec, s);
        default: return CERB_EUNSUPPORTED;
    }
}

}  // namespace cerb
