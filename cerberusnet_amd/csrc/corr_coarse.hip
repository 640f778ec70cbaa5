// corr_coarse.hip -- correlation forward / backward for the COARSE pyramid levels (d = 4, W <= 64; round 6: both on
// any EVEN W up to 64, the widths between 16 / 32 / 64 on the lanes of the next one, half a strip at the end of a row
// that is 2 mod 4 wide).
//
//   out[dy*9+dx][y][x] = leaky(1/C sum_c x1[c][y][x] * x2[c][y+dy-4][x+dx-4])
//   (reference: correlation_cuda_kernel.cu:29-95; same sums, other order)
//
// A coarse level is a few thousand pixels by hundreds of channels (4 x 256 x 16 x 32 at the
// top of the 1024 x 512 pyramid): 4 MB in, 0.7 MB out, 42 M multiply-adds -- nothing a 256-CU
// part can be busy with for long.  The tile kernels of corr_d4.hip take 11 us there, and 7.7 us
// of that is still there with 16 channels (tools/floor_probe.py): one loader wave per workgroup issues
// every LDS-DMA of its tile, the compute waves sit behind a barrier per chunk, 144 ds_bpermute
// close the channel groups, and only 128 workgroups exist.  A kernel of independent waves that
// each issue 16 or 32 loads and store once runs in 2.4-3.2 us here, launch included
// (tools/ubench/launch_floor.hip).  So this kernel has no loader, no ring and ONE barrier:
//   * a workgroup = one output row (b, y) x one displacement row dy; its NW = 4 waves split the
//     channels, and the 64 lanes of a wave are G = 64 / SPR channel groups x SPR four-pixel
//     strips (SPR = W / 4), so a lane owns C / (NW * G) channels of one strip;
//   * x1[c][y] and x2[c][y+dy-4] go global -> registers, one 16-byte load each per lane and
//     channel, the first 8 channels' worth requested before anything is waited for;
//   * the left / right neighbour strips of the 12-float window come from v_mov_b32_dpp
//     row_shr:K / row_shl:K with bound_ctrl: a 16-lane DPP row holds ONE image row of one
//     (SPR = 16) or K = 16 / SPR INTERLEAVED channel groups (lane i of a row: strip i / K, group
//     i % K), so a strip's neighbour is K lanes away, the row's ends are the image's ends, and the
//     zero the shift fills in is the reference's padding.  8 moves per 36 FMAs, no masks, no LDS;
//   * the NW * G partial sums of the row's 9 x W outputs meet in LDS (36 KB), one barrier, and
//     9 * SPR lanes add them in a fixed order (bit-reproducible), scale, apply the LeakyReLU
//     and store 16 bytes each;
//   * a displacement row that leaves the image (y+dy-4 outside [0, H)) only sums x1 * 0 over the
//     channels: the zeros -- or the NaN of a non-finite x1 -- the reference's padded x2 produces.
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <utility>

#include "common.h"

namespace cerb {
namespace {

constexpr int kD = 4;
constexpr int kND = 2 * kD + 1;
[[maybe_unused]] constexpr int kDead = static_cast<int>(0x80000000u);   // buffer offset that is out of range: reads 0

typedef float f4 __attribute__((ext_vector_type(4)));

// n / d for n * d < 2^32 by one multiply-high (the magic number comes from the host): the block -> work item
// decode has no division by a run-time value on the way to a wave's first load
struct FastDiv {
    uint32_t m, d;
    explicit FastDiv(uint32_t dv) : m(static_cast<uint32_t>(((1ull << 32) + dv - 1) / dv)), d(dv) {}
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1 ? n : __umulhi(n, m); }
};

#if defined(__HIP_DEVICE_COMPILE__)
// four consecutive pixels of storage type T at a byte offset of a buffer resource, as loaded (Px4<T>::raw), and widened
// to fp32 where they are used -- NOT where they are requested: the conversion would wait for the load (16-bit storage is
// widened on use and rounded to nearest even once, on store: the arithmetic is fp32 throughout)
template <typename T> struct Px4 { typedef f4 raw; };
template <> struct Px4<__half> { typedef unsigned raw __attribute__((ext_vector_type(2))); };
template <> struct Px4<hip_bfloat16> { typedef unsigned raw __attribute__((ext_vector_type(2))); };
template <typename T>
__device__ __forceinline__ typename Px4<T>::raw load_px4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    else return __builtin_bit_cast(typename Px4<T>::raw, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
template <typename T>
__device__ __forceinline__ f4 widen_px4(typename Px4<T>::raw raw) {
    if constexpr (sizeof(T) == 4) {
        return raw;
    } else if constexpr (std::is_same<T, __half>::value) {
        // (by 16-bit halves: hipcc 7.2 drops the second dword when a vector ELEMENT is bit-cast to a _Float16 pair)
        auto h = [](unsigned bits) { return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(bits))); };
        return f4{h(raw[0]), h(raw[0] >> 16), h(raw[1]), h(raw[1] >> 16)};
    } else {
        return f4{__builtin_bit_cast(float, raw[0] << 16), __builtin_bit_cast(float, raw[0] & 0xffff0000u),
                  __builtin_bit_cast(float, raw[1] << 16), __builtin_bit_cast(float, raw[1] & 0xffff0000u)};
    }
}
template <typename T>
__device__ __forceinline__ void store_px4(T *p, f4 v) {
    if constexpr (sizeof(T) == 4) {
        __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
    } else {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        typedef float f2v __attribute__((ext_vector_type(2)));
        u2 raw;
        if constexpr (std::is_same<T, __half>::value) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            raw[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[0], v[1]}, h2));
            raw[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[2], v[3]}, h2));
        } else {
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            raw[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[0], v[1]}, b2));
            raw[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[2], v[3]}, b2));
        }
        __builtin_nontemporal_store(raw, reinterpret_cast<u2 *>(p));
    }
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// two consecutive pixels (the half strips of a width that is 2 mod 4: 8-byte / 4-byte aligned)
template <typename T>
__device__ __forceinline__ void store_px2(T *p, float a, float b) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    if constexpr (sizeof(T) == 4) {
        __builtin_nontemporal_store(f2v{a, b}, reinterpret_cast<f2v *>(p));
    } else if constexpr (std::is_same<T, __half>::value) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(__builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, h2)), reinterpret_cast<unsigned *>(p));
    } else {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(__builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, b2)), reinterpret_cast<unsigned *>(p));
    }
}
#endif

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// SPR  4-pixel strips per image row (W = 4 * SPR, SPR | 16)
// NW   waves per workgroup (channel slices)
// CB   channels per load batch and lane; two batches in flight
// RAG  1: the image row is narrower than the lanes' 4 * SPR columns (any W % 4 == 0 up to that): the true width is a
//      run-time value, the strips past the row's end load nothing -- their zeros are the zero padding their left
//      neighbour's shift picks up -- and store nothing (round 6)
//      2: the same for W % 4 == 2 (1216 x 352 frames: the coarsest level is 38 wide): the row's last strip is HALF a strip.
//      Its 16-byte load brings the next row's first two pixels along: they are zeroed where the strip is the shifted
//      operand (two selects per channel), and the outputs leave as two 8-byte halves (rows are 8-byte aligned, not 16)
template <int SPR_, int NW_, int CB_, int RAG_ = 0>
struct CoarseFwdCfg {
    static constexpr bool RAG = RAG_ != 0, PART = RAG_ == 2;
    static constexpr int SPR = SPR_, W = 4 * SPR_, NW = NW_, CB = CB_;
    static constexpr int KI = 16 / SPR_;          // channel groups interleaved inside a 16-lane DPP row
    static constexpr int G = 64 / SPR_;           // channel groups per wave
    static constexpr int CMULT = NW_ * G * CB_;   // C must be a multiple of this
    static constexpr int THREADS = 64 * NW_;
    static constexpr int NF = kND * SPR_;         // 16-byte groups in the 9 x W outputs of a workgroup
    static constexpr size_t LDS_BYTES = static_cast<size_t>(NW_) * NF * 16;
    static_assert(16 % SPR_ == 0, "an image row must fit a 16-lane DPP row");
    static_assert(LDS_BYTES <= 64 * 1024, "no LDS opt-in");
};

#ifdef CERB_STAMP
// diagnostic build only (-DCERB_STAMP): s_memrealtime (100 MHz) of every wave at its phase boundaries
__device__ unsigned long long g_coarse_stamps[2048][8][8];
#define COARSE_STAMP(k)                                                                                 \
    do {                                                                                                \
        if ((threadIdx.x & 63) == 0 && blockIdx.y * gridDim.x + blockIdx.x < 2048) g_coarse_stamps[blockIdx.y * gridDim.x + blockIdx.x][threadIdx.x >> 6][k] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define COARSE_STAMP(k) do {} while (0)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ float dpp0(float v) {   // v of the lane CTRL names; 0 where that lane is outside the 16-lane row
    float r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
    asm volatile("" : "+v"(r));   // keep it a move: a DPP operand on the FMA itself halves the FMA rate (dpp_probe.hip)
    return r;
}
// in program order, one accumulator each: hipcc's v_pk_fma_f32 pairing costs a v_mov per operand pair and is
// no faster on gfx950 (tools/ubench/fma_rate.hip)
__device__ __forceinline__ void fmac(float &acc, float a, float b) {
    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}
// a + b where lane l of the result is a[l] + b[l] summed over the two 16-lane rows of a row pair: rows
// {0, 1} of the result hold a's rows 0+1 and b's rows 0+1, rows {2, 3} hold a's 2+3 and b's 2+3
// (v_permlane16_swap: odd rows of the first operand <-> even rows of the second)
__device__ __forceinline__ float rows_pair_sum(float a, float b) {
    // inline asm: hipcc 7.2 folds the builtin's two results into one register (v_add v, a, a) when they are added
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// s, t as produced by rows_pair_sum from (a, b) and (c, d): rows 0..3 of the result = the sums over all four
// rows of a, b, c, d (v_permlane32_swap: upper half of the first operand <-> lower half of the second)
__device__ __forceinline__ float halves_sum(float s, float t) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s), "+v"(t));
    return s + t;
}
template <int QP>
__device__ __forceinline__ float quad_add(float v) {   // v + v of the lane the quad permutation names
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), QP, 0xf, 0xf, true));
}
#endif

template <typename K, typename T>
__global__ __launch_bounds__(K::THREADS, K::NW >= 8 ? 5 : 3) void corr_fwd_d4_coarse_kernel(
    const T *__restrict__ x1, const T *__restrict__ x2, T *__restrict__ out, int C, int H, int Wimg,
    int cpl, float slope, int64_t out_bstride, unsigned per_xcd, unsigned nitems, FastDiv by_h) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) f4 part[];
    constexpr int SPR = K::SPR, CB = K::CB;
    const int W = K::RAG ? Wimg : K::W;        // the image's width (K::W: the lanes' capacity and the pitch of the LDS rows)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i16 = lane & 15;
    const int sx = i16 / K::KI;
    const int g = (lane >> 4) * K::KI + i16 % K::KI;

    COARSE_STAMP(0);
    // 8 * per_xcd workgroups: consecutive ids go round the 8 XCDs, so XCD x takes the items
    // [x * per_xcd, (x + 1) * per_xcd) of the (image, row, dy) list -- at 4 pairs half an image: neighbouring
    // rows, which share their x2 rows, meet in one L2 (fabric reads 11.1 -> 5.3 MB at 256 x 16 x 32)
    const unsigned item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= nitems) return;
    const unsigned row = item / kND;                           // (b, y)
    const int dy = __builtin_amdgcn_readfirstlane(item - row * kND);
    const int b = __builtin_amdgcn_readfirstlane(by_h.div(row));
    const int y = __builtin_amdgcn_readfirstlane(row - b * H);
    const int y2 = y + dy - kD;
    const int plane = H * W;
    const int64_t obs = out_bstride ? out_bstride : static_cast<int64_t>(kND * kND) * plane;
    T *orow = out + b * obs + static_cast<int64_t>(dy * kND) * plane + y * W;
    constexpr int E = sizeof(T);   // bytes per stored element

    const bool inside = y2 >= 0 && y2 < H;   // else the whole displacement row reads padding

    const int item_bytes = C * plane * E;
    const __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(x1 + static_cast<int64_t>(b) * C * plane, item_bytes);
    const __amdgpu_buffer_rsrc_t r2 = uniform_rsrc(x2 + static_cast<int64_t>(b) * C * plane, item_bytes);
    const int c0 = (wave * K::G + g) * cpl;
    const bool strip_live = !K::RAG || 4 * sx < W;
    const bool strip_whole = !K::PART || 4 * sx + 4 <= W;     // (PART: else pixels 2, 3 of the strip belong to the next row)
    const int v1 = strip_live ? ((c0 * H + y) * W + 4 * sx) * E : kDead;
    const int v2 = strip_live ? ((c0 * H + y2) * W + 4 * sx) * E : kDead;
    const int nb = cpl / CB;

    float acc[kND][4];
    typename Px4<T>::raw xa[2][CB], xw[2][CB];
    // every batch is requested unconditionally (past the last one: out of range, zeros, no traffic) so
    // that the compiler's count of requests in flight is exact and a wait never covers the batch
    // requested just before it
    auto load = [&](int set, int k) {
        const int soff = __builtin_amdgcn_readfirstlane(k * CB * plane * E);
        const bool live = k < nb;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            xa[set][i] = load_px4<T>(r1, (live && strip_live) ? v1 + i * plane * E : kDead, soff);
            xw[set][i] = load_px4<T>(r2, (live && inside && strip_live) ? v2 + i * plane * E : kDead, soff);
        }
    };
    float zacc[4] = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int set, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;   // the first channel writes the accumulators (no zeroing pass)
        if (!inside) {
            // x1 * 0 summed over the channels, as the reference's zero-padded x2 gives it: 0, or NaN where x1 is not finite
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                const f4 a = widen_px4<T>(xa[set][i]);
#pragma unroll
                for (int p = 0; p < 4; ++p) zacc[p] = __builtin_fmaf(a[p], 0.f, zacc[p]);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const f4 a = widen_px4<T>(xa[set][i]);
            f4 w = widen_px4<T>(xw[set][i]);
            if constexpr (K::PART) {
                w[2] = strip_whole ? w[2] : 0.f;
                w[3] = strip_whole ? w[3] : 0.f;
            }
            float win[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                win[j] = dpp0<0x110 + K::KI>(w[j]);       // row_shr: the strip to the left
                win[4 + j] = w[j];
                win[8 + j] = dpp0<0x100 + K::KI>(w[j]);   // row_shl: the strip to the right
            }
#pragma unroll
            for (int d = 0; d < kND; ++d)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (FIRST && i == 0) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(acc[d][p]) : "v"(a[p]), "v"(win[p + d]));
                    else fmac(acc[d][p], a[p], win[p + d]);
                }
        }
    };
    load(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    COARSE_STAMP(1);
#ifdef CERB_STAMP
    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    COARSE_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    COARSE_STAMP(3);
#endif
    compute(0, std::true_type{});
    __builtin_amdgcn_sched_barrier(0);
    load(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(1, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    load(1, 3);
    __builtin_amdgcn_sched_barrier(0);
    for (int k = 2; k < nb; k += 2) {
        compute(0, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        load(0, k + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(1, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        load(1, k + 3);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!inside) {
#pragma unroll
        for (int d = 0; d < kND; ++d)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[d][p] = zacc[p];
    }

    COARSE_STAMP(4);
    // channel groups of the wave: interleaved neighbours by a quad permutation, then the four DPP rows by
    // two lane-swap rounds -- row p of red[d] ends up with the sums of acc[d][p], i.e. pixel 4 * sx + p
    if constexpr (K::KI >= 2) {
#pragma unroll
        for (int d = 0; d < kND; ++d)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                acc[d][p] = quad_add<0xb1>(acc[d][p]);                          // [1,0,3,2]
                if constexpr (K::KI == 4) acc[d][p] = quad_add<0x4e>(acc[d][p]);   // [2,3,0,1]
            }
    }
    {
        float *dst = reinterpret_cast<float *>(part) + wave * (kND * K::W) + 4 * sx + (lane >> 4);
#pragma unroll
        for (int d = 0; d < kND; ++d) {
            const float red = halves_sum(rows_pair_sum(acc[d][0], acc[d][1]), rows_pair_sum(acc[d][2], acc[d][3]));
            if (i16 % K::KI == 0) dst[d * K::W] = red;
        }
    }
    __syncthreads();
    COARSE_STAMP(5);
    const float inv = 1.0f / static_cast<float>(C);
    for (int t = tid; t < K::NF; t += K::THREADS) {
        f4 s0 = part[t];
#pragma unroll
        for (int w = 1; w < K::NW; ++w) s0 += part[w * K::NF + t];
        f4 q = s0 * inv;
#pragma unroll
        for (int p = 0; p < 4; ++p) q[p] = q[p] > 0.f ? q[p] : q[p] * slope;
        if constexpr (K::PART) {
            T *o = orow + (t / SPR) * plane + (t % SPR) * 4;
            if ((t % SPR) * 4 < W) store_px2<T>(o, q[0], q[1]);
            if ((t % SPR) * 4 + 4 <= W) store_px2<T>(o + 2, q[2], q[3]);
        } else {
            if (!K::RAG || (t % SPR) * 4 < W) store_px4<T>(orow + (t / SPR) * plane + (t % SPR) * 4, q);
        }
    }
    COARSE_STAMP(6);
#ifdef CERB_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    COARSE_STAMP(7);
#endif
#endif
}

template <typename K, typename T>
int launch_coarse_fwd(const char *name, const void *in1, const void *in2, void *outp, const CorrGeom &g,
                      float slope, int64_t obs, hipStream_t s) {
    const int64_t nitems = static_cast<int64_t>(g.B) * g.H * kND;
    if (nitems * g.H >= (1ll << 32) || nitems + 8 > 0x7fffffff) return CERB_EUNSUPPORTED;   // FastDiv range, grid size
    const unsigned per_xcd = static_cast<unsigned>((nitems + 7) / 8);
    note_kernel(0, name);
    hipLaunchKernelGGL((corr_fwd_d4_coarse_kernel<K, T>), dim3(8 * per_xcd), dim3(K::THREADS),
                       K::LDS_BYTES, s, static_cast<const T *>(in1), static_cast<const T *>(in2),
                       static_cast<T *>(outp), g.C, g.H, g.W, g.C / (K::NW * K::G), slope, obs, per_xcd,
                       static_cast<unsigned>(nitems), FastDiv(static_cast<uint32_t>(g.H)));
    return launch_status();
}

// ============================================================================
// backward
// ============================================================================
//   gI1[c][y][x] = 1/C sum_{dy,dx} gO[dy,dx][y][x]                  * x2[c][y+dy-4][x+dx-4]
//   gI2[c][y][x] = 1/C sum_{dy,dx} gO[8-dy,8-dx][y+dy-4][x+dx-4]    * x1[c][y+dy-4][x+dx-4]
//   (reference: correlation_cuda_kernel.cu:97-172 and :174-242; same sums, other order)
// Same lanes as the forward (G channel groups x SPR strips, CPL channels per lane), but here the
// channels are independent and the sum runs over the 81 displacements: a workgroup = one output
// row of one gradient (b, side, y) x G * CPL channels, its THREE waves take three displacement rows
// each (acc stays in registers across them) and meet once in LDS.  Per displacement row a wave
//   * requests the x row (CPL 16-byte loads per lane) and the nine gradOutput rows of that dy by
//     LDS-DMA into a wave-private set (64 lanes = 64 / SPR rows per instruction, so the G channel
//     groups share ONE copy instead of loading it G times); two sets: the next dy travels while
//     this one is consumed;
//   * side 2 reads gradOutput at the SOURCE pixel (x+dx-4): the shift is applied on the global side
//     of the DMA (a dword-aligned 16-byte read per lane, as in corr_strip.hip) and the |dx-4| taps
//     that fall outside the image row are zeroed in LDS once the rows have landed (20 dwords, one
//     ds_write_b32 by 20 lanes);
//   * nine ds_read_b128 bring the rows' values for the lane's strip, then 36 FMAs + 8 DPP moves
//     per channel.
// No barrier before the final one, no division by a run-time value.
template <int SPR_, int CPL_, int RAG_ = 0>
struct CoarseBwdCfg {
    static constexpr bool RAG = RAG_ != 0, PART = RAG_ == 2;   // as in CoarseFwdCfg: a row narrower than the lanes' 4 * SPR columns; W % 4 == 2
    static constexpr int SPR = SPR_, W = 4 * SPR_, CPL = CPL_;
    static constexpr int KI = 16 / SPR_, G = 64 / SPR_, CSET = G * CPL_;
    static constexpr int NWV = 3, NDYW = 3, THREADS = 64 * NWV;
    static constexpr int RPI = 64 / SPR_;                       // gradOutput rows per DMA instruction
    static constexpr int NDMA = (kND + RPI - 1) / RPI;          // DMA instructions per displacement row
    static constexpr int SET = NDMA * 1024;                     // bytes of one staged displacement row (9 x W floats + slack)
    static constexpr int WAVE_LDS = 2 * SET > CPL_ * 1024 ? 2 * SET : CPL_ * 1024;   // the partial sums reuse the wave's sets
    static constexpr size_t LDS_BYTES = static_cast<size_t>(NWV) * WAVE_LDS;
    static_assert(16 % SPR_ == 0, "an image row must fit a 16-lane DPP row");
    static_assert(NWV * NDYW == kND, "the waves cover the nine displacement rows");
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) void *lds_void_ptr;
template <int OFF>
__device__ __forceinline__ void lds_read16(f4 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
#endif

template <typename K>
__global__ __launch_bounds__(K::THREADS, 4) void corr_bwd_d4_coarse_kernel(
    const float *__restrict__ x1, const float *__restrict__ x2, const float *__restrict__ gout,
    float *__restrict__ g1, float *__restrict__ g2, int C, int H, int Wimg, unsigned per_xcd, unsigned nitems, FastDiv by_ncs,
    FastDiv by_h) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SPR = K::SPR, CPL = K::CPL;
    const int W = K::RAG ? Wimg : K::W;        // the image's width (K::W: the lanes' capacity and the pitch of the staged rows)
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i16 = lane & 15;
    const int sx = i16 / K::KI;
    const int g = (lane >> 4) * K::KI + i16 % K::KI;
    // 8 * per_xcd workgroups; XCD x takes the items [x * per_xcd, (x + 1) * per_xcd) of the (image, gradient,
    // row, channel set) list -- at 4 pairs one gradient of one image: its x tensor is read through one L2 only
    // (fabric reads 19.3 -> 5.6 MB at 256 x 16 x 32)
    const unsigned item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= nitems) return;
    const unsigned yr = by_ncs.div(item);                      // (b, side, y)
    const int cs = item - yr * by_ncs.d;
    const unsigned bs = by_h.div(yr);
    const int y = yr - bs * H;
    const int side = bs & 1, b = bs >> 1;
    const int plane = H * W;

    auto run = [&](auto side_c) {
        constexpr int SIDE = decltype(side_c)::value;
        const float *xin = SIDE == 0 ? x2 : x1;
        float *gdst = SIDE == 0 ? g1 : g2;
        const __amdgpu_buffer_rsrc_t rx = uniform_rsrc(xin + static_cast<int64_t>(b) * C * plane, C * plane * 4);
        const __amdgpu_buffer_rsrc_t rg = uniform_rsrc(gout + static_cast<int64_t>(b) * (kND * kND) * plane, kND * kND * plane * 4);
        const int c0 = cs * K::CSET + g * CPL;
        const unsigned wave_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem)) + wv * K::WAVE_LDS;
        const unsigned lane_cell = wave_base + sx * 16;          // the lane's strip inside a staged row
        const int lr = lane / SPR, st = lane % SPR;              // DMA lanes: row inside the instruction, strip
        // side 2: dword q of the 20 out-of-row taps of a displacement row (dx < 4: the first 4 - dx dwords
        // of the row, dx > 4: the last dx - 4)
        unsigned patch_off = 0;
        if constexpr (SIDE == 1) {
            const int q = lane;
            int dx, p, right;
            if (q < 10) { right = 0; dx = q < 4 ? 0 : q < 7 ? 1 : q < 9 ? 2 : 3; p = q - (q < 4 ? 0 : q < 7 ? 4 : q < 9 ? 7 : 9); }
            else { const int r = q - 10; right = 1; dx = r < 1 ? 5 : r < 3 ? 6 : r < 6 ? 7 : 8; p = r < 1 ? 3 : r < 3 ? r + 1 : r < 6 ? r - 2 : r - 6; }
            patch_off = dx * (SPR * 16) + (right ? (W - 4) * 4 : 0) + p * 4;      // the image row's true end
        }

        float acc[CPL][4];
#pragma unroll
        for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[i][p] = 0.f;
        f4 xs[2][CPL];
        bool ok[K::NDYW];
#pragma unroll
        for (int k = 0; k < K::NDYW; ++k) ok[k] = static_cast<unsigned>(y + K::NDYW * wv + k - kD) < static_cast<unsigned>(H);

        // every wave issues the same NDMA + CPL requests per displacement row, unconditionally (rows outside
        // the image: out of range, zeros, no traffic), so the waits below count exactly
        auto stage = [&](int k, int set) {
            const int dy = K::NDYW * wv + k, xr = y + dy - kD;
#pragma unroll
            for (int q = 0; q < K::NDMA; ++q) {
                const int dx = q * K::RPI + lr;
                const int pl = SIDE == 0 ? dy * kND + dx : (kND - 1 - dy) * kND + (kND - 1 - dx);
                const int row = SIDE == 0 ? y : xr;
                const int vo = (pl * H + row) * (W * 4) + st * 16 + (SIDE ? (dx - kD) * 4 : 0);
                // (side 0 reads gradOutput of its own row even when the x row lies outside the image: see compute)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void_ptr)(smem + (wv * K::WAVE_LDS + set * K::SET + q * 1024) / 4), 16,
                                                         ((SIDE == 0 || ok[k]) && dx < kND && (!K::RAG || st * 4 < W)) ? vo : kDead, 0, 0, 0);
            }
            const int vx = ((c0 * H + xr) * W + 4 * sx) * 4;
#pragma unroll
            for (int i = 0; i < CPL; ++i)
                xs[set][i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, (ok[k] && (!K::RAG || sx * 4 < W)) ? vx + i * plane * 4 : kDead, 0, 0));
        };
        auto compute = [&](int k, int set, auto pending_c) {
            constexpr int PENDING = decltype(pending_c)::value;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory");
            const unsigned base = set * K::SET;
            if constexpr (SIDE == 1) {
                if (lane < 20) asm volatile("ds_write_b32 %0, %1" ::"v"(wave_base + base + patch_off), "v"(0.f) : "memory");
            }
            f4 gq[kND];
            static_for<0, kND>([&](auto dc) {
                constexpr int dx = decltype(dc)::value;
                lds_read16<dx * (SPR * 16)>(gq[dx], lane_cell + base);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gq[0]), "+v"(gq[1]), "+v"(gq[2]), "+v"(gq[3]), "+v"(gq[4]), "+v"(gq[5]),
                         "+v"(gq[6]), "+v"(gq[7]), "+v"(gq[8]));
            if (ok[k]) {
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    f4 w = xs[set][i];
                    if constexpr (K::PART) {          // the row's last strip is half a strip: its pixels 2, 3 are the next row's
                        w[2] = sx * 4 + 4 <= W ? w[2] : 0.f;
                        w[3] = sx * 4 + 4 <= W ? w[3] : 0.f;
                    }
                    float win[12];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        win[j] = dpp0<0x110 + K::KI>(w[j]);
                        win[4 + j] = w[j];
                        win[8 + j] = dpp0<0x100 + K::KI>(w[j]);
                    }
#pragma unroll
                    for (int d = 0; d < kND; ++d)
#pragma unroll
                        for (int p = 0; p < 4; ++p) fmac(acc[i][p], win[p + d], gq[d][p]);
                }
            } else if constexpr (SIDE == 0) {
                // the x2 row is padding: gradOutput * 0, the NaN of a non-finite gradOutput included
                // (correlation_cuda_kernel.cu:150-165 multiplies with the zero-padded copy)
                float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < kND; ++d)
#pragma unroll
                    for (int p = 0; p < 4; ++p) z[p] = __builtin_fmaf(gq[d][p], 0.f, z[p]);
#pragma unroll
                for (int i = 0; i < CPL; ++i)
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[i][p] += z[p];
            }
        };
        // the counted waits below are only right while ONE stage() is exactly NDMA LDS-DMA requests + CPL
        // 16-byte x loads per wave, all unconditional, and compute() issues no vector-memory operation at all (its
        // ds_write / ds_read count in lgkmcnt).  Anyone adding a load to stage() or compute(), or a compiler splitting a
        // b128 load, breaks "vmcnt(PER) = the previous stage has landed": keep PER tied to stage() here, and keep the
        // non-finite-locality and random-shape sweep tests of tests/test_corr_gpu.py in the default GPU run -- they are the
        // guard (a wrong count reads rows that have not landed: data-dependent garbage, never a crash).
        constexpr int PER = K::NDMA + CPL;
        static_assert(PER == K::NDMA + CPL && PER <= 15, "vmcnt immediates of compute(): one stage = NDMA DMAs + CPL loads, < 16 in flight");
        stage(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        stage(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(0, 0, std::integral_constant<int, PER>{});
        __builtin_amdgcn_sched_barrier(0);
        stage(2, 0);
        __builtin_amdgcn_sched_barrier(0);
        compute(1, 1, std::integral_constant<int, PER>{});
        __builtin_amdgcn_sched_barrier(0);
        compute(2, 0, std::integral_constant<int, 0>{});

        // the three waves' partial sums meet in LDS (a wave's slot lies over its own, finished, sets)
        {
            f4 *dst = reinterpret_cast<f4 *>(smem + wv * K::WAVE_LDS / 4) + lane;
#pragma unroll
            for (int i = 0; i < CPL; ++i) dst[i * 64] = f4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        }
        __syncthreads();
        const float inv = 1.0f / static_cast<float>(C);
        for (int k = tid; k < CPL * 64; k += K::THREADS) {
            const int i = k >> 6, l = k & 63;
            const f4 *src = reinterpret_cast<const f4 *>(smem) + i * 64 + l;
            f4 sum = src[0];
#pragma unroll
            for (int w = 1; w < K::NWV; ++w) sum += src[w * (K::WAVE_LDS / 16)];
            const int lsx = (l & 15) / K::KI, lg = (l >> 4) * K::KI + (l & 15) % K::KI;
            const int c = cs * K::CSET + lg * CPL + i;
            float *o = gdst + ((static_cast<int64_t>(b) * C + c) * H + y) * W + 4 * lsx;
            if constexpr (K::PART) {
                const f4 r = sum * inv;
                if (4 * lsx < W) store_px2<float>(o, r[0], r[1]);
                if (4 * lsx + 4 <= W) store_px2<float>(o + 2, r[2], r[3]);
            } else {
                if (!K::RAG || 4 * lsx < W) __builtin_nontemporal_store(sum * inv, reinterpret_cast<f4 *>(o));
            }
        }
    };
    if (side == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
#endif
}

template <typename K>
int launch_coarse_bwd(const char *name, const void *in1, const void *in2, const void *goutp, void *g1p, void *g2p,
                      const CorrGeom &g, hipStream_t s) {
    const int ncs = g.C / K::CSET;
    const int64_t nitems = static_cast<int64_t>(g.B) * 2 * g.H * ncs;
    if (nitems * std::max(ncs, g.H) >= (1ll << 32) || nitems + 8 > 0x7fffffff) return CERB_EUNSUPPORTED;   // FastDiv range, grid size
    const unsigned per_xcd = static_cast<unsigned>((nitems + 7) / 8);
    note_kernel(1, name);
    hipLaunchKernelGGL((corr_bwd_d4_coarse_kernel<K>), dim3(8 * per_xcd), dim3(K::THREADS), K::LDS_BYTES, s,
                       static_cast<const float *>(in1), static_cast<const float *>(in2), static_cast<const float *>(goutp),
                       static_cast<float *>(g1p), static_cast<float *>(g2p), g.C, g.H, g.W, per_xcd, static_cast<unsigned>(nitems),
                       FastDiv(static_cast<uint32_t>(ncs)), FastDiv(static_cast<uint32_t>(g.H)));
    return launch_status();
}

}  // namespace

#ifdef CERB_STAMP
extern "C" int cerberus_debug_coarse_stamps(void *dst, int bytes) {
    return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_coarse_stamps), std::min<size_t>(bytes, sizeof(g_coarse_stamps))));
}
#endif

// fp32, vector-aligned tensors, an item below 2 GiB (the caller has checked): W = 16 / 32 / 64 and
// C a multiple of the lane layout's channel count; CERB_EUNSUPPORTED otherwise
template <typename T>
static int coarse_forward_t(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope, int64_t obs,
                            hipStream_t s) {
    // four waves x four-channel batches: 7.2 / 8.3 us alone at the two coarse levels of the 1024 x 512 pyramid, 4 pairs
    // (eight waves x two-channel batches: 7.6 / 9.0 -- the fixed work per wave is then as large as its FMAs).  With the
    // two directions' streams started together the 8-wave layout was 2 % better in the step (86 vs 121 VGPRs); since the
    // second stream forks one launch late (bench.py) the two are level there (6 alternating runs) and the 4-wave one is
    // 0.8 % ahead on one stream
    using K16 = CoarseFwdCfg<4, 4, 2>;
    using K32 = CoarseFwdCfg<8, 4, 4>;
    using K64 = CoarseFwdCfg<16, 4, 4>;
    if (g.W == 64 && g.C % K64::CMULT == 0) return launch_coarse_fwd<K64, T>("corr_fwd_d4_coarse_64", in1, in2, out, g, slope, obs, s);
    if (g.W == 32 && g.C % K32::CMULT == 0) return launch_coarse_fwd<K32, T>("corr_fwd_d4_coarse_32", in1, in2, out, g, slope, obs, s);
    if (g.W == 16 && g.C % K16::CMULT == 0) return launch_coarse_fwd<K16, T>("corr_fwd_d4_coarse_16", in1, in2, out, g, slope, obs, s);
    // round 6: widths between those run on the lanes of the next one (RAG)
    using R16 = CoarseFwdCfg<4, 4, 2, 1>;
    using R32 = CoarseFwdCfg<8, 4, 4, 1>;
    using R64 = CoarseFwdCfg<16, 4, 4, 1>;
    if (g.W % 4 == 0) {
        if (g.W > 32 && g.W < 64 && g.C % R64::CMULT == 0) return launch_coarse_fwd<R64, T>("corr_fwd_d4_coarse_rag64", in1, in2, out, g, slope, obs, s);
        if (g.W > 16 && g.W < 32 && g.C % R32::CMULT == 0) return launch_coarse_fwd<R32, T>("corr_fwd_d4_coarse_rag32", in1, in2, out, g, slope, obs, s);
        if (g.W < 16 && g.C % R16::CMULT == 0) return launch_coarse_fwd<R16, T>("corr_fwd_d4_coarse_rag16", in1, in2, out, g, slope, obs, s);
    }
    // widths that are 2 mod 4: half a strip at the row's end
    using P16 = CoarseFwdCfg<4, 4, 2, 2>;
    using P32 = CoarseFwdCfg<8, 4, 4, 2>;
    using P64 = CoarseFwdCfg<16, 4, 4, 2>;
    if (g.W % 4 == 2) {
        if (g.W > 32 && g.W < 64 && g.C % P64::CMULT == 0) return launch_coarse_fwd<P64, T>("corr_fwd_d4_coarse_half64", in1, in2, out, g, slope, obs, s);
        if (g.W > 16 && g.W < 32 && g.C % P32::CMULT == 0) return launch_coarse_fwd<P32, T>("corr_fwd_d4_coarse_half32", in1, in2, out, g, slope, obs, s);
        if (g.W > 4 && g.W < 16 && g.C % P16::CMULT == 0) return launch_coarse_fwd<P16, T>("corr_fwd_d4_coarse_half16", in1, in2, out, g, slope, obs, s);
    }
    return CERB_EUNSUPPORTED;
}

// vector-aligned tensors (16 bytes fp32, 8 bytes 16-bit storage), an item below 2 GiB (the caller has checked):
// W = 16 / 32 / 64 and C a multiple of the lane layout's channel count; CERB_EUNSUPPORTED otherwise
int corr_coarse_forward(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope,
                        int64_t obs, int dtype, hipStream_t s) {
    switch (dtype) {
        case CERB_F32: return coarse_forward_t<float>(in1, in2, out, g, slope, obs, s);
        case CERB_F16: return coarse_forward_t<__half>(in1, in2, out, g, slope, obs, s);
        case CERB_BF16: return coarse_forward_t<hip_bfloat16>(in1, in2, out, g, slope, obs, s);
        default: return CERB_EUNSUPPORTED;
    }
}

// fp32 backward for the coarse levels, same preconditions as the forward
int corr_coarse_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2, const CorrGeom &g,
                         hipStream_t s) {
    using K16 = CoarseBwdCfg<4, 2>;
    using K32 = CoarseBwdCfg<8, 4>;
    using K64 = CoarseBwdCfg<16, 4>;
    if (g.W == 64 && g.C % K64::CSET == 0) return launch_coarse_bwd<K64>("corr_bwd_d4_coarse_64", in1, in2, gout, gin1, gin2, g, s);
    if (g.W == 32 && g.C % K32::CSET == 0) return launch_coarse_bwd<K32>("corr_bwd_d4_coarse_32", in1, in2, gout, gin1, gin2, g, s);
    if (g.W == 16 && g.C % K16::CSET == 0) return launch_coarse_bwd<K16>("corr_bwd_d4_coarse_16", in1, in2, gout, gin1, gin2, g, s);
    // round 6: widths between those run on the lanes of the next one (RAG)
    using R16 = CoarseBwdCfg<4, 2, 1>;
    using R32 = CoarseBwdCfg<8, 4, 1>;
    using R64 = CoarseBwdCfg<16, 4, 1>;
    if (g.W % 4 == 0) {
        if (g.W > 32 && g.W < 64 && g.C % R64::CSET == 0) return launch_coarse_bwd<R64>("corr_bwd_d4_coarse_rag64", in1, in2, gout, gin1, gin2, g, s);
        if (g.W > 16 && g.W < 32 && g.C % R32::CSET == 0) return launch_coarse_bwd<R32>("corr_bwd_d4_coarse_rag32", in1, in2, gout, gin1, gin2, g, s);
        if (g.W < 16 && g.C % R16::CSET == 0) return launch_coarse_bwd<R16>("corr_bwd_d4_coarse_rag16", in1, in2, gout, gin1, gin2, g, s);
    }
    using P16 = CoarseBwdCfg<4, 2, 2>;
    using P32 = CoarseBwdCfg<8, 4, 2>;
    using P64 = CoarseBwdCfg<16, 4, 2>;
    if (g.W % 4 == 2) {
        if (g.W > 32 && g.W < 64 && g.C % P64::CSET == 0) return launch_coarse_bwd<P64>("corr_bwd_d4_coarse_half64", in1, in2, gout, gin1, gin2, g, s);
        if (g.W > 16 && g.W < 32 && g.C % P32::CSET == 0) return launch_coarse_bwd<P32>("corr_bwd_d4_coarse_half32", in1, in2, gout, gin1, gin2, g, s);
        if (g.W > 4 && g.W < 16 && g.C % P16::CSET == 0) return launch_coarse_bwd<P16>("corr_bwd_d4_coarse_half16", in1, in2, gout, gin1, gin2, g, s);
    }
    return CERB_EUNSUPPORTED;
}

}  // namespace cerb
