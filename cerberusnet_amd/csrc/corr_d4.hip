// corr_d4.hip -- tuned CDNA4 (gfx950) kernels for the one correlation
// configuration every CerberusNet model uses: pad = max_displacement = 4,
// kernel 1, stride1 = stride2 = 1 (pwcnet_sfd.py:131-133), fp32.
// FORWARD kernels and their dispatch (the backward tile kernels live in corr_d4_bwd.hip; shared helpers in corr_d4_common.h).
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
#include "corr_d4_common.h"

namespace cerb {
namespace {

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


template <typename T>
int fwd_dispatch(const void *x1, const void *x2, void *o, const CorrGeom &g, float slope,
                 int64_t obs, bool vec, bool half, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        // 16-bit storage, 16 < C <= 128: the matrix-core kernel (corr_mfma.hip; 14 forces it); variants 1-8 keep the VALU kernels
        const int v = option(OPT_CORR_FWD_VARIANT);
        if (vec && dma_ok(g) && g.C <= 128 && option(OPT_CORR_NO_MFMA) == 0 && (v == 14 || v == 20 || v == 26 || (v == 0 && g.C > 16)))   // <= 16 channels fill half an MFMA: no gain
            return corr_mfma_forward(x1, x2, o, g, slope, obs,
                                     std::is_same<T, __half>::value ? CERB_F16 : CERB_BF16, s);
    }
    {
        // coarse levels (W <= 64): independent waves, no loader, one barrier (corr_coarse.hip); 15 forces it, 16 keeps it off
        // (4 pairs: 7.0 vs 11.6 us at 256 x 16 x 32, 8.1 vs 10.8 us at 128 x 32 x 64; the tile kernels catch up
        // once a launch has more than ~2500 (row, displacement row) workgroups).  16-bit storage: the same kernel
        // with the loads widened (the matrix-core kernel above keeps 16 < C <= 128)
        const int v = option(OPT_CORR_FWD_VARIANT);
        const bool coarse_auto = g.W <= 64 && static_cast<int64_t>(g.B) * g.H * kND <= 2560;   // 8 pairs of 128 x 32 x 64: 12.5 vs 13.0 us
        if ((vec || half) && dma_ok(g) && (v == 15 || (v == 0 && coarse_auto))) {
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
#ifdef CERB_EXPERIMENTS
        case 17:   // the persistent, cross-item pipelined forward (corr_fwd_pipe.hip): built, measured slower (profiles/r05_fwd_pipe_experiment.txt), test builds only
            if constexpr (sizeof(T) == 4) {
                if (vec && dma_ok(g)) {
                    const int rc = corr_fwd_pipe(x1, x2, o, g, slope, obs, s);
                    if (rc != CERB_EUNSUPPORTED) return rc;
                }
            }
            break;
#endif
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
    if constexpr (sizeof(T) == 4) {
        if (dma) {
            // Round 6: the LDS-DMA variants by a cost estimate instead of "the largest tile with >= 256 workgroups".  That
            // rule assumed widths that are multiples of the tile: at 64 x 44 x 152 it picked 2 x 64 tiles (264 workgroups of
            // which the last eight run alone: 20.2 us) over 4 x 64 (132 workgroups, 15.6 us), at 128 x 28 x 56 the 1 x 16
            // tiles (17.9 us) over 1 x 32 (10.5).  A workgroup takes ~7 us of latency + 0.52 us per 1024 pixel-channels of
            // its tile, the workgroups of a CU run one after the other: est = ceil(workgroups / 256) x (7 + 0.52 x tile
            // pixels x C / 1024) ranks all 45 (shape, variant) timings of profiles/r06_ragged_forward_variants.txt in their
            // measured order and leaves the benched pyramid's choices as they were.
            auto est = [&](int64_t wgs, int tile_px) {
                return static_cast<double>((wgs + 255) / 256) * (7.0 + 0.52 * tile_px * g.C / 1024.0);
            };
            int best = 0;
            double cost = est(fwd_tiles<FwdDma4>(g), 256);
            auto consider = [&](int id, bool ok, int64_t wgs, int px) {
                if (!ok) return;
                const double c = est(wgs, px);
                if (c < cost) { cost = c; best = id; }
            };
            consider(1, g.C % 2 == 0, fwd_tiles<FwdDmaB>(g), 128);
            consider(2, g.C % 4 == 0, fwd_tiles<FwdDmaC>(g), 64);
            consider(3, g.C % 8 == 0, fwd_tiles<FwdDmaD>(g), 32);
            consider(4, g.C % 16 == 0, fwd_tiles<FwdDmaE>(g), 16);
            switch (best) {
                case 1: return launch_fwd_dma<FwdDmaB>("corr_fwd_d4_dma_2x64_s2", x1, x2, o, g, slope, obs, s);
                case 2: return launch_fwd_dma<FwdDmaC>("corr_fwd_d4_dma_1x64_s4", x1, x2, o, g, slope, obs, s);
                case 3: return launch_fwd_dma<FwdDmaD>("corr_fwd_d4_dma_1x32_s8", x1, x2, o, g, slope, obs, s);
                case 4: return launch_fwd_dma<FwdDmaE>("corr_fwd_d4_dma_1x16_s16", x1, x2, o, g, slope, obs, s);
                default: return launch_fwd_dma<FwdDma4>("corr_fwd_d4_dma_4x64", x1, x2, o, g, slope, obs, s);
            }
        }
    }
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


}  // namespace

int corr_d4_forward(const void *in1, const void *in2, void *out, const CorrGeom &g, float slope,
                    int64_t obs, int dtype, hipStream_t s) {
    if (!fast_config(g, dtype)) return CERB_EUNSUPPORTED;
    const bool vec = g.W % 4 == 0 && aligned_group(in1, dtype) && aligned_group(in2, dtype) &&
                     aligned_group(out, dtype) && (obs % 4 == 0);
    // W % 4 == 2: the coarse-level kernels take the row's last two pixels as half a strip (corr_coarse.hip, round 6)
    const bool half = g.W % 4 == 2 && aligned_group(in1, dtype) && aligned_group(in2, dtype) && aligned_group(out, dtype) && (obs % 2 == 0);
    switch (dtype) {
        case CERB_F32: return fwd_dispatch<float>(in1, in2, out, g, slope, obs, vec, half, s);
        case CERB_F16: return fwd_dispatch<__half>(in1, in2, out, g, slope, obs, vec, half, s);
        case CERB_BF16: return fwd_dispatch<hip_bfloat16>(in1, in2, out, g, slope, obs, vec, half, s);
        default: return CERB_EUNSUPPORTED;
    }
}


}  // namespace cerb
