// corr_fwd_pipe.hip -- correlation forward (fp32, d = 4) as a PERSISTENT, cross-item pipelined grid (round 5).
//
//   out[(dy+4)*9+(dx+4)][y][x] = 1/C sum_c x1[c][y][x] * x2[c][y+dy][x+dx]      (correlation_cuda_kernel.cu:29-95)
//
// Why: at the 32 x 128 x 256 level (4 pairs) the tile kernel of corr_d4.hip (corr_fwd_d4_dma_kernel: 4 x 64 tile, nine
// displacement-row waves + a loader wave) is ONE round of 512 workgroups: every workgroup runs its channel loop at
// the same time and then every workgroup stores its 81 planes at the same time -- 42.5 MB, 56 % of the launch's bytes,
// in a burst of ~8 us that nothing overlaps, because an output is complete only after the last channel and the
// whole level's accumulators fit the chip at once.  Delaying workgroups does not help (the delay costs
// what it hides); what helps is to give every resident workgroup TWO (or more) sequential work items of half the
// size, so that the stores of item i drain under the channel loop of item i + 1 and the loader wave -- which never
// stops at an item boundary -- has item i + 1's first chunks in LDS before item i's last FMA.
//
// Halving an item without doubling the LDS reads per FMA means halving the lanes' CHANNELS, not their accumulators:
//   * work item = a 4 x 32 pixel tile; a wavefront = one displacement row dy (nine compute waves + one loader, as in
//     the tile kernel); lanes 0-31 sum the first half of the channels for the tile's 32 strips, lanes 32-63 the
//     second half for the same strips (36 accumulators per lane: 9 dx x 4 pixels);
//   * the two halves meet in ONE v_permlane32_swap per PAIR of accumulators (upper half of a <-> lower half of b, then
//     a + b): the lower lanes end up with the finished sums of dx planes 0, 2, 4, 6, 8, the upper lanes with 1, 3, 5, 7
//     -- so the store work is split too: five 16-byte stores per lane instead of nine;
//   * per chunk of 4 + 4 channels the loader streams 8 x2 windows (12 rows x 12 16-byte slots: the 10 the tile needs
//     + 2 so that rows lie 48 floats apart) and the lanes' own x1 strips into a ring of three LDS buffers: 18 + 4
//     LDS-DMA instructions whose 64 lanes each carry their own global offset (a slot's plane, row and column), so the
//     x2 windows of all 8 planes are packed without a gap.  Rows 48 floats apart + the lane order below make every
//     ds_read_b128 of the loop conflict-free (tools/lds_conflicts.py: 4 cycles; the natural 40-float rows: 6);
//   * lane order inside a 32-lane half: the hardware serves a ds_read_b128 in the 16-lane groups {0-3, 12-15, 20-27}
//     and {4-11, 16-19, 28-31}; rows r and r + 2 of the window start 8 slots (mod 16) apart, so the first group owns
//     tile rows 0 and 2, the second rows 1 and 3.
// Summation order differs from the tile kernels' (two channel halves): equal to fp32 rounding, not bit for bit.
#include <atomic>

#include "common.h"

namespace cerb {
namespace {

constexpr int kD = 4;
constexpr int kND = 2 * kD + 1;
constexpr int kP = 4;
[[maybe_unused]] constexpr int kDeadOff = static_cast<int>(0x80000000u);   // buffer offset that is out of range: reads 0

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_void_ptr;

struct PipeCfg {
    static constexpr int S = 2, TSX = 8, TH = 4, TW = TSX * kP;
    static constexpr int HR = TH + 2 * kD;            // window rows
    static constexpr int HW4 = 12;                    // 16-byte slots per window row (10 needed)
    static constexpr int RS = HW4 * 4, PS = HR * RS;  // floats per row / per plane
    static constexpr int CC = 4, NCH = S * CC, NB = 3;
    static constexpr int NI2 = NCH * HR * HW4 / 64;   // DMA instructions for the x2 windows of a chunk
    static constexpr int NINST = NI2 + CC;
    static constexpr int BUF = NCH * PS + CC * 256;   // floats per ring buffer
    static constexpr int THREADS = 64 * (kND + 1);
    static constexpr size_t LDS_BYTES = sizeof(float) * NB * BUF;
    static_assert(NCH * HR * HW4 % 64 == 0, "whole DMA instructions");
    static_assert(NINST * (NB - 2) <= 63, "vmcnt is 6 bits");
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float2v pkfma(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#if defined(__HIP_DEVICE_COMPILE__)
// s, t -> lane l < 32: s[l] + s[l + 32]; lane l >= 32: t[l - 32] + t[l]   (v_permlane32_swap: upper half of the first
// operand <-> lower half of the second; inline asm: hipcc 7.2 folds the builtin's two results into one register)
__device__ __forceinline__ float halves_sum(float s, float t) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s), "+v"(t));
    return s + t;
}
#endif

__global__ __launch_bounds__(PipeCfg::THREADS, 5) void corr_fwd_d4_pipe_kernel(
    const float *__restrict__ x1, const float *__restrict__ x2, float *__restrict__ out, int C, int H, int W, int tiles_x,
    int tiles_y, int nitems, int per_wg, float slope, int64_t out_bstride) {
#if defined(__HIP_DEVICE_COMPILE__)
    using K = PipeCfg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wg = xcd_chunk(blockIdx.x, gridDim.x);
    const int it0 = wg * per_wg, it1 = min(it0 + per_wg, nitems);
    if (it0 >= it1) return;
    const int plane = H * W;
    const int Cg = C / K::S;
    const int nchunks = Cg / K::CC;   // launcher: C % 8 == 0
    // lane -> (channel half, tile row, strip): see the header
    const int cg = lane >> 5, h = lane & 31;
    const int r = h < 4 ? 0 : h < 12 ? 1 : h < 16 ? 0 : h < 20 ? 3 : h < 28 ? 2 : 3;
    const int sx = h < 4 ? h : h < 12 ? h - 4 : h < 16 ? h - 8 : h < 20 ? h - 16 : h < 28 ? h - 20 : h - 24;
    const int tiles = tiles_x * tiles_y;

    if (wave == kND) {
        // ------------------------------ loader wavefront ------------------------------
        // slot s = 64 n + lane of instruction n -> (plane pl = 2 i + g, window row R, 16-byte column c4); fixed for the
        // whole launch: byte offset relative to the tile's origin
        int rel[K::NI2];
#pragma unroll
        for (int n = 0; n < K::NI2; ++n) {
            const int s = 64 * n + lane;
            const int pl = s / (K::HR * K::HW4), rem = s - pl * (K::HR * K::HW4);
            const int R = rem / K::HW4, c4 = rem - R * K::HW4;
            const int i = pl >> 1, g = pl & 1;
            rel[n] = ((g * Cg + i) * plane + (R - kD) * W + (4 * c4 - kD)) * 4;
        }
        const int rel1 = (cg * Cg * plane + r * W + 4 * sx) * 4;
        const int item_bytes = C * plane * 4;
        const int total = (it1 - it0) * nchunks;
        int v1 = kDeadOff, base = 0;
        unsigned dead = 0;     // bit n: the lane's slot of instruction n lies outside the image (border tiles only)
        __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(x1, item_bytes), r2 = uniform_rsrc(x2, item_bytes);
        int iss_it = it0, iss_k = 0, iss_q = 0, iss_slot = 0;
        auto issue_next = [&]() {
            if (iss_q >= total) return;
            if (iss_k == 0) {
                // a new work item: its tile origin, validity of the window's slots, the batch item's resources
                const int tx = __builtin_amdgcn_readfirstlane(iss_it % tiles_x);
                const int ty = __builtin_amdgcn_readfirstlane((iss_it / tiles_x) % tiles_y);
                const int b = __builtin_amdgcn_readfirstlane(iss_it / tiles);
                const int x0 = tx * K::TW, y0 = ty * K::TH;
                base = (y0 * W + x0) * 4;
                const bool interior = y0 >= kD && y0 + K::TH + kD <= H && x0 >= kD && x0 - kD + 4 * K::HW4 <= W;
                dead = 0;
                if (!interior) {
#pragma unroll
                    for (int n = 0; n < K::NI2; ++n) {
                        const int s = 64 * n + lane;
                        const int rem = s % (K::HR * K::HW4);
                        const int gy = y0 - kD + rem / K::HW4, gx = x0 - kD + 4 * (rem % K::HW4);
                        dead |= (gy >= 0 && gy < H && gx >= 0 && gx < W) ? 0u : (1u << n);
                    }
                }
                v1 = (y0 + r < H && x0 + 4 * sx < W) ? rel1 + base : kDeadOff;
                r1 = uniform_rsrc(x1 + static_cast<int64_t>(b) * C * plane, item_bytes);
                r2 = uniform_rsrc(x2 + static_cast<int64_t>(b) * C * plane, item_bytes);
            }
            float *buf = smem + iss_slot * K::BUF;
            const int soff = __builtin_amdgcn_readfirstlane(iss_k * K::CC * plane * 4);
            {
#pragma unroll
            for (int n = 0; n < K::NI2; ++n)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_void_ptr)(buf + n * 256), 16,
                                                         (dead >> n) & 1u ? kDeadOff : rel[n] + base, soff, 0, 0);
#pragma unroll
            for (int i = 0; i < K::CC; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_void_ptr)(buf + K::NCH * K::PS + i * 256), 16, v1,
                                                         soff + i * plane * 4, 0, 0);
            }
            ++iss_q;
            iss_slot = iss_slot + 1 == K::NB ? 0 : iss_slot + 1;
            if (++iss_k == nchunks) { iss_k = 0; ++iss_it; }
        };
#pragma unroll
        for (int k = 0; k < K::NB - 1; ++k) issue_next();
        for (int q = 0; q < total; ++q) {
            // this wave's DMAs for chunk q have landed once at most the younger chunk remains in flight
            if (q + 1 < total) wait_vmcnt<K::NINST>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // chunk q visible to the compute waves; chunk q - 1 consumed
                issue_next();                   // chunk q + 2, into the buffer chunk q - 1 used
        }
        return;
    }

    // -------------------------------- compute wavefronts --------------------------------
    // Loop-invariant state is kept small (80 VGPRs = 6 waves per SIMD, see the launcher): ONE LDS offset and ONE
    // output offset per lane; everything that changes per item or chunk is scalar.
    const float inv_nelems = 1.0f / static_cast<float>(C);
    const int64_t obs = out_bstride ? out_bstride : static_cast<int64_t>(kND * kND) * plane;
    const int x2_lane = cg * K::PS + (r + wave) * K::RS + 4 * sx;       // floats, inside a ring buffer
    const int out_lane = ((wave * kND + cg) * plane + r * W + 4 * sx) * 4;   // bytes: dx plane `cg` of the wave's row of planes
    const int out_bytes = kND * kND * plane * 4;
    int slot = 0;
#pragma unroll 1
    for (int it = it0; it < it1; ++it) {
        float2v accp[kP][4];
        float accs[kP];
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            accs[p] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) accp[p][j] = float2v{0.f, 0.f};
        }
#pragma unroll 1
        for (int k = 0; k < nchunks; ++k) {
            __builtin_amdgcn_s_barrier();
            const float *buf = smem + slot * K::BUF;
            slot = slot + 1 == K::NB ? 0 : slot + 1;
            const float *X2 = buf + x2_lane;
            const float *X1 = buf + K::NCH * K::PS + 4 * lane;
#pragma unroll
            for (int i = 0; i < K::CC; ++i) {
                const float4 a = ld4(X1 + i * 256);
                const float *bp = X2 + i * K::S * K::PS;
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

        // ---- the item's epilogue: unpack, meet the other channel half, scale, LeakyReLU, store.  The stores drain
        // under the next item's channel loop (its first chunks are already in LDS: the loader ran ahead).
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
        const int tx = __builtin_amdgcn_readfirstlane(it % tiles_x);
        const int ty = __builtin_amdgcn_readfirstlane((it / tiles_x) % tiles_y);
        const int b = __builtin_amdgcn_readfirstlane(it / tiles);
        const int x0 = tx * K::TW, y0 = ty * K::TH;
        const __amdgpu_buffer_rsrc_t r_out = uniform_rsrc(out + b * obs, out_bytes);
        const int item_off = (y0 * W + x0) * 4;                               // scalar
        const int voff = (y0 + r < H && x0 + 4 * sx < W) ? out_lane : kDeadOff;   // a lane outside the image stores nothing
        // dx planes 2j (lower lanes) and 2j + 1 (upper lanes); j = 4: plane 8 in both halves, the lower one stores.
        // All swaps first, then the stores back to back: the swaps are inline asm the compiler's hazard recogniser
        // cannot see into, and one that rewrote a register of the 16-byte store just issued before it corrupted that
        // store's data now and then (lanes 12-15 / 28-31 of one or two planes per ~500 tiles).
        f4v v[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int p = 0; p < kP; ++p) {
                const float q = halves_sum(acc[2 * j][p], acc[j < 4 ? 2 * j + 1 : 8][p]) * inv_nelems;
                v[j][p] = q > 0.f ? q : q * slope;
            }
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int j = 0; j < 5; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v[j]), r_out, (j == 4 && cg) ? kDeadOff : voff,
                                                   item_off + 2 * j * plane * 4, 2 /* nt */);
    }
#endif
}

}  // namespace

// fp32, pad = d = 4, W % 4 == 0, 16-byte aligned tensors (checked by the caller); CERB_EUNSUPPORTED for channel counts
// the two 4-channel half chunks do not divide
int corr_fwd_pipe(const void *in1, const void *in2, void *outp, const CorrGeom &g, float slope, int64_t obs,
                  hipStream_t s) {
    using K = PipeCfg;
    if (g.C % (K::S * K::CC) != 0) return CERB_EUNSUPPORTED;
    const int tiles_x = (g.W + K::TW - 1) / K::TW, tiles_y = (g.H + K::TH - 1) / K::TH;
    const int64_t nitems = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    if (nitems > 0x7fffffff) return CERB_ETOOLARGE;
    // two workgroups per CU stay resident (LDS, 5 waves per SIMD); every one walks `per_wg` consecutive tiles
    int wgs = option(OPT_CORR_BWD_CSLICE) > 0 ? option(OPT_CORR_BWD_CSLICE) : 512;   // (experiments: the slice knob sets the grid)
    const int per_wg = static_cast<int>((nitems + wgs - 1) / wgs);
    wgs = static_cast<int>((nitems + per_wg - 1) / per_wg);
    static std::atomic<uint64_t> lds_done{0};
    int rc;
    if ((rc = ensure_lds(corr_fwd_d4_pipe_kernel, K::LDS_BYTES, &lds_done))) return rc;
    note_kernel(0, "corr_fwd_d4_pipe_4x32_s2");
    hipLaunchKernelGGL(corr_fwd_d4_pipe_kernel, dim3(static_cast<unsigned>(wgs)), dim3(K::THREADS), K::LDS_BYTES, s,
                       static_cast<const float *>(in1), static_cast<const float *>(in2), static_cast<float *>(outp), g.C,
                       g.H, g.W, tiles_x, tiles_y, static_cast<int>(nitems), per_wg, slope, obs);
    return launch_status();
}

}  // namespace cerb
