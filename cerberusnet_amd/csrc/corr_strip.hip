// corr_strip.hip -- correlation backward for image rows that fit one wavefront (fp32, d = 4).
//
//   gI1[c][y][x] = 1/C sum_{dy,dx} gO[dy,dx][y][x]          * x2[c][y+dy][x+dx]
//   gI2[c][y][x] = 1/C sum_{ey,ex} gO[-ey,-ex][y+ey][x+ex]  * x1[c][y+ey][x+ex]
//   (reference: correlation_cuda_kernel.cu:97-172 and :174-242; same sums, other order)
//
// The kernels of corr_d4.hip cut the map into 4x64 / 8x64 tiles and read the x window of a
// lane's strip from LDS: three ds_read_b128 per 36 FMAs keep the LDS two thirds busy, the 81
// gradOutput values of a pixel either sit in 162 registers (2 waves/SIMD) or are re-read per
// channel slice, and one round of workgroups runs its phases in lock-step.
// Here a lane owns a 4-pixel STRIP and a wavefront owns whole image rows (W = 256: one row per
// wave, 64 strips; W = 128 / 64: 2 / 4 rows), so the horizontal neighbours of a strip are the
// neighbouring LANES:
//   * x rows go global -> registers, one 16-byte load per lane, channel and row; the left and
//     right neighbour strips of the 12-float window come from v_mov_b32_dpp (wave_shr:1 /
//     wave_shl:1 or row_shr:1 / row_shl:1, bound_ctrl): a row's ends ARE the image's ends, and
//     the zero a DPP shift fills in is the reference's zero padding -- no halo, no LDS for x.
//     (DPP on the FMA itself runs at half rate on gfx950 -- tools/ubench/dpp_probe.hip -- so the
//     shift is 8 moves per channel row, shared by the 72 FMAs of two output rows.)
//   * a lane accumulates NR = 2 vertically adjacent output rows x CW = 4 channels (32
//     accumulators): the x row of step s serves dy = s-4 of the upper and dy = s-5 of the lower
//     row, so a wave reads 10 rows for 2;
//   * a workgroup walks its ten x rows CYCLICALLY, starting where its neighbours are (step
//     (t - y0) mod 10 at time t): the five workgroups whose windows hold an image row fetch it in
//     the same time step -- one L2 miss, four hits.  In natural order they fetch it two steps
//     apart and the 2 x 3.2 MB an XCD moves in two steps do not fit its 4 MiB L2: FETCH_SIZE
//     175 MB instead of 94 MB per launch (algorithmic reads: 76 MB);
//   * gradOutput is streamed, never held: a step needs 2 x 9 planes of one image row (18 KB).
//     They arrive by LDS-DMA three steps ahead in a ring of four slots, shared by the 8 waves of
//     the workgroup (8 x 4 = 32 channels), and a lane fetches the 4 values of (row, dx) just in
//     time -- one ds_read_b128 per 16 FMAs;
//   * side 2 (gI2) is the same arithmetic on the flipped plane index with the gradOutput row
//     read SHIFTED by ex on the global side of the DMA (a dword-aligned 16-byte read per lane);
//     the |ex| taps that fall outside the image row are zeroed in LDS once the row has landed,
//     so they read exact zeros, as the reference skips them.  (An unaligned ds_read_b128 would
//     do the shift on the LDS side, and works, but runs ~15x slower than an aligned one.)
//   * a wave requests the x row of step t+2 and the planes of step t+3 AFTER its FMAs of step t:
//     the ~130 cycles a vector-memory instruction takes to issue when a whole CU issues at once
//     are spread over the waves' staggered arrivals at the step's one barrier;
//   * 120 VGPRs -> 4 waves/SIMD (2 workgroups of 8 waves per CU at the 32 x 128 x 256 level).
// The FMAs, LDS reads and their waits are inline asm in program order (hipcc hoists every LDS
// read of an unrolled step to its top and spills otherwise); global loads, DMAs
// and DPP moves are builtins the compiler counts and pads itself.  Every wave issues the same
// number of vector-memory instructions per step unconditionally, so that count is exact.
// Where the time goes (4 pairs of 32 x 128 x 256, tools/stamp_strip.py): the ten steps are
// 84 % VALU-busy at the 2.35 cycles a v_fmac_f32 takes with 4 waves per SIMD.
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
typedef int i4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_void_ptr;

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// SPR  4-pixel strips per image row (W = 4 * SPR); a wave holds RW = 64 / SPR lane-rows
// NWV  waves per workgroup = channel groups of CW = 4 that share the gradOutput stream
// RAG  1: ragged shapes -- the image is narrower than the lanes' 4 * SPR columns (any W % 4 == 0 up to that), H need not
//      be a multiple of the workgroup's rows, C not of its channels: the true width is a run-time value, lanes past a
//      row's end load nothing (their zeros ARE the zero padding their left neighbour's DPP shift picks up), stores are masked
// NRP  row pairs per workgroup (round 6 experiment): NRP = 2 -> 16 waves = 2 row pairs x 8 channel groups.  The two
//      pairs walk the cyclic schedule two steps apart, so they request the SAME x rows in the same time step (the second request
//      is an L1 hit: 12 x rows from L2 for 4 output rows instead of 10 for 2); the gradOutput ring holds both pairs' planes
//      (2 x 18 per step: 147 KB, one workgroup per CU at the same 4 waves per SIMD)
template <int SPR_, int NWV_, int FLAGS_ = 0, int CW_ = 4, int NIT_ = 1, int RAG_ = 0, int NRP_ = 1>
struct StripCfg {
    static constexpr int NIT = NIT_;
    static constexpr bool RAG = RAG_ != 0;
    static constexpr int NRP = NRP_;   // work items a workgroup walks one after the other (round 5 experiment: 2 -> half the grid)
    // timing-experiment flags (-DCERB_ABLATE builds): 1 gradOutput DMA non-temporal, 2 x loads non-temporal, 4 no FMAs,
    // 8 no loads / DMAs, 16 no LDS reads, 32 natural step order, 64 no step barrier, 256 no stores (all but 1, 2, 32, 512: wrong results),
    // 512 experiment: a step's memory instructions inside its FMA stream instead of behind it (slower: see step())
    static constexpr int FLAGS = FLAGS_;
    static constexpr int SPR = SPR_, W = 4 * SPR_, RW = 64 / SPR_;
    static constexpr int NR = 2, CW = CW_, NWV = NWV_, CWG = CW * NWV_;
    static constexpr int ROWS = NR * RW;                 // image rows per row pair (a wave's rows)
    static constexpr int ROWS_WG = NRP_ * ROWS;          // image rows per workgroup
    static constexpr int NWAVES = NWV_ * NRP_;
    static constexpr int THREADS = 64 * NWAVES;
    static constexpr int NSLOT = 4;
    // waves per SIMD the register allocator leaves room for: two 8-wave workgroups per CU on the 256-wide level
    // (one round of 512 workgroups at 4 pairs); the narrower levels have fewer workgroups than that anyway
    static constexpr int WPS = NIT_ > 1 ? 2 : (SPR_ == 64 || CW_ < 4) ? 4 : 2;
    static constexpr int PITCH = W * 4;                  // bytes per staged gradOutput row
    static constexpr int ENTRY = RW * PITCH;             // one (j, dx) plane: RW rows
    static constexpr int PAIRB = NR * kND * ENTRY;       // one row pair's planes of one step
    static constexpr int SLOT = NRP_ * PAIRB;            // one step
    static constexpr int NPL = NRP_ * NR * kND;          // planes per step
    static constexpr int NDMA = (NPL + NWAVES - 1) / NWAVES;   // DMA instructions per wave and step
    static constexpr size_t LDS_BYTES = static_cast<size_t>(NSLOT) * SLOT + ENTRY;   // + a scratch entry for idle DMA slots
    static_assert(64 % SPR_ == 0, "whole rows per wave");
    static_assert(NRP_ == 1 || NIT_ == 1, "row pairs and multi-item walks are separate experiments");
    static_assert((NR * kND - 1) * ENTRY < 65536, "ds_read immediate offsets");
};

#ifdef CERB_STAMP
// diagnostic build only (-DCERB_STAMP): s_memtime of every wave of the first 64 workgroups at the phase
// boundaries of each step, fetched with cerberus_debug_strip_stamps(); never compiled into the product
__device__ unsigned long long g_strip_stamps[64][8][64];
__device__ unsigned long long g_strip_life[512][8][4];   // s_memrealtime: wave start, loop end, stores acknowledged
#define STRIP_STAMP(k)                                                                                   \
    do {                                                                                                 \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) g_strip_stamps[blockIdx.x][threadIdx.x >> 6][k] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define STRIP_STAMP(k) do {} while (0)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float dpp_mov(float v, std::integral_constant<int, 0>) {   // from lane-1 (wave)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_mov(float v, std::integral_constant<int, 1>) {   // from lane+1 (wave)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_mov(float v, std::integral_constant<int, 2>) {   // from lane-1 (16-lane row)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_mov(float v, std::integral_constant<int, 3>) {   // from lane+1 (16-lane row)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

template <int OFF>
__device__ __forceinline__ void lds_read16(f4 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait(f4 &v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "i"(N));
}
__device__ __forceinline__ void fmac(float &acc, float a, float b) {
    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}

template <typename K, int SIDE>
struct StripBwd {
    __amdgpu_buffer_rsrc_t rsrc_x, rsrc_g;
    float *smem;
    int H, plane, wave, lane, y0;
    int rp;                        // the wave's row pair (0 unless K::NRP > 1); y0 is THIS pair's first row
    int Wr;                        // the image's width (RAG: <= K::W, the lanes' capacity; otherwise K::W itself)
    bool lane_live;                // the lane's strip lies inside the image row
    __device__ __forceinline__ int width() const { if constexpr (K::RAG) return Wr; else return K::W; }
    unsigned lds_base, lds_lane;   // byte address of the ring / of the lane's cell inside an entry
    int row0, voff0;               // image row of the lane's x row at step 0, its byte offset in a plane
    f4 xs[2][K::CW];
    float acc[K::NR][K::CW][4];

    // x rows of step S (image row row0 + S), CW channels; `live` false: nothing is fetched (zeros)
    __device__ __forceinline__ void load_x(int S, bool live, f4 (&dst)[K::CW]) {
        const int row = row0 + S;
        int vo = voff0 + S * (width() * 4);
        vo = (live && (!K::RAG || lane_live) && static_cast<unsigned>(row) < static_cast<unsigned>(H)) ? vo : kDead;
#pragma unroll
        for (int c = 0; c < K::CW; ++c)
            if constexpr (!(K::FLAGS & 8))
                dst[c] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, vo, c * plane * 4, (K::FLAGS & 2) ? 2 : 0));
    }

    // one channel of load_x (the interleaved schedule of step() issues them one at a time)
    __device__ __forceinline__ void load_x1(int S, bool live, f4 (&dst)[K::CW], int c) {
        const int row = row0 + S;
        int vo = voff0 + S * (width() * 4);
        vo = (live && (!K::RAG || lane_live) && static_cast<unsigned>(row) < static_cast<unsigned>(H)) ? vo : kDead;
        if constexpr (!(K::FLAGS & 8))
            dst[c] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, vo, c * plane * 4, (K::FLAGS & 2) ? 2 : 0));
    }

    // gradOutput planes of step S -> ring slot `slot`.  Plane k = j * 9 + dx is requested by wave
    // k % NWV.  Every wave issues the SAME number of DMA instructions per step, unconditionally
    // (a wave without a plane to fetch sends an out-of-range request to a scratch row): the
    // compiler's vmcnt bookkeeping for the x loads then knows exactly how many younger
    // operations exist, and never waits for a request that has just been issued.
    __device__ __forceinline__ void issue_g(int S, bool live, int slot) {
#pragma unroll
        for (int q = 0; q < K::NDMA; ++q) issue_g1(S, live, slot, q);
    }
    __device__ __forceinline__ void issue_g1(int S, bool live, int slot, int q) {
        const int lr = lane / K::SPR, sx = lane % K::SPR;
        {
            const int k = wave + K::NWAVES * q;                 // wave-uniform: plane k of the step's NPL
            // (NRP > 1: plane k belongs to row pair kp, whose rows start (kp - rp) * ROWS below this wave's and whose cyclic
            // schedule is that many steps behind)
            const int kp = K::NRP > 1 ? k / (K::NR * kND) : 0, kk = k - kp * (K::NR * kND);
            const int y0k = y0 + (kp - rp) * K::ROWS;
            int Sk = S;
            if constexpr (K::NRP > 1) { Sk = (S - (kp - rp) * K::ROWS) % 10; Sk = Sk < 0 ? Sk + 10 : Sk; }
            const int j = kk >= kND ? 1 : 0, dx = kk - j * kND;
            const int dyi = Sk - j;                             // vertical displacement index of the block
            const bool act = live && k < K::NPL && dyi >= 0 && dyi < kND;
            const int pl = SIDE == 0 ? dyi * kND + dx : (kND - 1 - dyi) * kND + (kND - 1 - dx);
            // one instruction = the 64 cells of an entry: lane-row lr reads its image row (side 0:
            // the lane's own output row j; side 1: the x row of the step)
            const int row = SIDE == 0 ? y0k + lr * K::NR + j : y0k + lr * K::NR + Sk - kD;
            const bool ok = act && (!K::RAG || lane_live) && static_cast<unsigned>(row) < static_cast<unsigned>(H);
            // side 1 reads the row shifted by ex = dx - 4 (a dword-aligned 16-byte read per lane).
            // The request never leaves the batch item's 81 planes (ex < 0 only occurs on planes
            // >= 5, ex > 0 only on planes <= 75), so the cells at a row's ends pick up a few values
            // of the neighbouring row: patch_g() zeroes them once the data has landed.  (The row
            // offset rides in the VGPR offset: the range check looks at that one alone, and the
            // shift by ex < 0 of a row's first lane must not make it negative.)
            const int vo = (pl * H + row) * (width() * 4) + sx * 16 + (SIDE ? (dx - kD) * 4 : 0);
            const int doff = act ? slot * K::SLOT + k * K::ENTRY : K::NSLOT * K::SLOT;   // idle: scratch entry behind the ring
            if constexpr (!(K::FLAGS & 8))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_g, (lds_void_ptr)(smem + doff / 4), 16,
                                                         ok ? vo : kDead, 0, 0, (K::FLAGS & 1) ? 2 : 0);
        }
    }

    // side 1: zero the taps of step S's staged rows that lie outside the image row (|ex| dwords at
    // one end of the row); each wave patches the planes it requested, after they have landed
    __device__ __forceinline__ void patch_g(int S, int slot) {
        if constexpr (SIDE == 1 && !(K::FLAGS & 8)) {
#pragma unroll
            for (int q = 0; q < K::NDMA; ++q) {
                const int k = wave + K::NWAVES * q;
                const int kp = K::NRP > 1 ? k / (K::NR * kND) : 0, kk = k - kp * (K::NR * kND);
                int Sk = S;
                if constexpr (K::NRP > 1) { Sk = (S - (kp - rp) * K::ROWS) % 10; Sk = Sk < 0 ? Sk + 10 : Sk; }
                const int j = kk >= kND ? 1 : 0, dx = kk - j * kND, ex = dx - kD;
                const int dyi = Sk - j;
                const bool act = k < K::NPL && dyi >= 0 && dyi < kND;
                const int n = act ? (ex < 0 ? -ex : ex) : 0;
#pragma unroll
                for (int lr = 0; lr < K::RW; ++lr) {
                    const unsigned row = lds_base + slot * K::SLOT + k * K::ENTRY + lr * K::PITCH;
                    const unsigned a = ex < 0 ? row + lane * 4 : row + width() * 4 - 4 - lane * 4;   // the image row's true end
                    if (lane < n) asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(0.f) : "memory");
                }
            }
        }
    }

    // the nine (j, dx) blocks of output row j: 4 gradOutput values just in time, 16 FMAs each
    template <int J, typename Hook>
    __device__ __forceinline__ void row_blocks(const float (&win)[K::CW][12], unsigned lds_cur, Hook &&hook) {
        constexpr int DEPTH = 2;                              // reads in flight (3: no gain, 7 more VGPRs)
        f4 gq[DEPTH + 1];
        auto issue = [&](auto bc) {
            constexpr int dx = decltype(bc)::value;
            if constexpr (!(K::FLAGS & 16)) lds_read16<(J * kND + dx) * K::ENTRY>(gq[dx % (DEPTH + 1)], lds_cur);
        };
        static_for<0, DEPTH>(issue);
        static_for<0, kND>([&](auto bc) {
            constexpr int dx = decltype(bc)::value;
            if constexpr (dx + DEPTH < kND) issue(std::integral_constant<int, dx + DEPTH>{});
            constexpr int pending = (kND - 1 - dx) < DEPTH ? (kND - 1 - dx) : DEPTH;
            if constexpr (!(K::FLAGS & 16)) lgkm_wait<pending>(gq[dx % (DEPTH + 1)]);
            const f4 g = gq[dx % (DEPTH + 1)];
#pragma unroll
            for (int c = 0; c < K::CW; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(K::FLAGS & 4) || (c == 0 && dx == 0)) fmac(acc[J][c][i], win[c][i + dx], g[i]);
            hook(bc);
        });
    }

    // one step at time t: x row S of the lane (register set XSET), gradOutput in ring slot `cur`.
    // AFTER its FMAs a wave requests the x row of step t+2 (S2, into the register set it has just
    // finished with) and the planes of step t+3 (S3, into the slot last read a step ago): the
    // vector-memory instructions (~130 cycles each when all waves of a CU issue together) are
    // spread over the waves' staggered arrivals at the barrier instead of following it.
    // LATE (round 5, guide "two waves that run the same program with one barrier per block: try a stagger"): the second half of a
    // workgroup's waves issues its step's gradOutput DMAs (LATE & 1) / x rows (LATE & 2) BEFORE its FMAs, the first half behind
    // them: the two waves a workgroup has on a SIMD are then in complementary phases -- one queueing at the texture path,
    // one on the FMA pipe -- instead of both doing the same thing at the same time.  Same instructions, same counts per step.
    template <int XSET, int LATE = 0>
    __device__ __forceinline__ void step(int S, int S1, int S2, int S3, bool live2, bool live3, int cur, int nxt,
                                         int s3, int t) {
        // ---- 12-float windows of the CW channels: [left strip | own | right strip] ----
        using ShrT = std::integral_constant<int, K::SPR == 64 || K::SPR == 32 ? 0 : 2>;
        using ShlT = std::integral_constant<int, K::SPR == 64 || K::SPR == 32 ? 1 : 3>;
        float win[K::CW][12];
#pragma unroll
        for (int c = 0; c < K::CW; ++c) {
            const f4 v = xs[XSET][c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float l = dpp_mov(v[q], ShrT{}), r = dpp_mov(v[q], ShlT{});
                if constexpr (K::SPR != 64 && K::SPR != 16) {
                    // lane-rows narrower than the DPP row: the image's ends fall inside it
                    l = (lane % K::SPR) == 0 ? 0.f : l;
                    r = (lane % K::SPR) == K::SPR - 1 ? 0.f : r;
                }
                win[c][q] = l; win[c][4 + q] = v[q]; win[c][8 + q] = r;
            }
        }
        const unsigned lds_cur = lds_lane + cur * K::SLOT;
#ifdef CERB_STAMP
        asm volatile("" : "+v"(win[0][0]), "+v"(win[1][3]), "+v"(win[3][11]));
        STRIP_STAMP(3 + 4 * t);
#endif
        // Round 5 experiment (FLAGS & 512, -DCERB_ABLATE builds; measured and rejected): the step's vector-memory instructions --
        // CW x rows for step t+2, NDMA gradOutput DMAs for step t+3 -- issued INSIDE the FMA stream, one every few dx blocks,
        // instead of all together behind it.  The idea came from the ablations (profiles/r05_sol_skeleton.txt): without FMAs the
        // launch takes 17.7 us, without loads 24.2, with both 34.3 -- the parts add.  Interleaved: 40.7 vs 34.5 us (8 pairs: 75 vs
        // 64): a wave that meets a busy texture path stalls AT the memory instruction, now in the middle of its FMAs, and the x
        // rows requested early need 16 registers beside the windows still in use (128 VGPRs + 8 spilled).
        auto mem_hook0 = [&](auto bc) {      // output row 0's blocks carry the x rows: channel c behind block floor(9 c / CW)
            constexpr int dx = decltype(bc)::value;
            if constexpr (K::FLAGS & 512)
                static_for<0, K::CW>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr ((c * kND) / K::CW == dx) load_x1(S2, live2, xs[XSET], c);
                });
        };
        auto mem_hook1 = [&](auto bc) {      // output row 1's blocks carry the DMAs: instruction q behind block floor(9 q / NDMA)
            constexpr int dx = decltype(bc)::value;
            if constexpr (K::FLAGS & 512)
                static_for<0, K::NDMA>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr ((q * kND) / K::NDMA == dx) issue_g1(S3, live3, s3, q);
                });
        };
        if constexpr (LATE & 2) load_x(S2, live2, xs[XSET]);
        if constexpr (LATE & 1) issue_g(S3, live3, s3);
        const bool has0 = S <= kND - 1, has1 = S >= 1;
        if (has0) row_blocks<0>(win, lds_cur, mem_hook0);        // dy = S - 4
        else if constexpr (K::FLAGS & 512) load_x(S2, live2, xs[XSET]);
        if (has1) row_blocks<1>(win, lds_cur, mem_hook1);        // dy = S - 5
        else if constexpr (K::FLAGS & 512) issue_g(S3, live3, s3);
        STRIP_STAMP(4 + 4 * t);
        if constexpr (!(K::FLAGS & 512)) {   // the schedule that measures best: everything behind the FMAs
            if constexpr (!(LATE & 2)) load_x(S2, live2, xs[XSET]);
            if constexpr (!(LATE & 1)) issue_g(S3, live3, s3);
        }
        // the gradOutput planes of the next step (requested two steps ago) have landed once only
        // the requests of the previous and of this step may still be in flight; nobody reads
        // slot `cur` after the barrier
        constexpr int younger = (K::FLAGS & 8) ? 0 : 2 * (K::CW + K::NDMA);
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(younger) : "memory");
        patch_g(S1, nxt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STRIP_STAMP(5 + 4 * t);
        if constexpr (!(K::FLAGS & 64)) __builtin_amdgcn_s_barrier();
        STRIP_STAMP(6 + 4 * t);
    }
};
#endif

template <typename K>
__global__ __launch_bounds__(K::THREADS, K::WPS)
void corr_bwd_d4_strip_kernel(const float *__restrict__ x1, const float *__restrict__ x2,
                              const float *__restrict__ gout, float *__restrict__ gin1,
                              float *__restrict__ gin2, int C, int H, int Wimg, int nyb, int ncb) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane / K::SPR, sx = lane % K::SPR;

    // (row block, channel block, side) with the side fastest: the two workgroups that stream
    // the same gradOutput rows are neighbours in the XCD-contiguous order
    for (int item = 0; item < K::NIT; ++item) {
    if (K::NIT > 1 && item > 0) __builtin_amdgcn_s_barrier();   // the previous item's last slot has been read by every wave
    int bid = xcd_chunk(blockIdx.x, gridDim.x) * K::NIT + item;
    const int side = __builtin_amdgcn_readfirstlane(bid & 1); bid >>= 1;
    const int cb = __builtin_amdgcn_readfirstlane(bid % ncb); bid /= ncb;
    const int yb = __builtin_amdgcn_readfirstlane(bid % nyb);
    const int b = __builtin_amdgcn_readfirstlane(bid / nyb);
    const int Wr = K::RAG ? Wimg : K::W;
    const int plane = H * Wr;
    const int rp = K::NRP > 1 ? wave / K::NWV : 0;            // the wave's row pair and channel group
    const int c0 = cb * K::CWG + (wave - rp * K::NWV) * K::CW;
    const int y0 = yb * K::ROWS_WG + rp * K::ROWS;

    const float *src = (side == 0 ? x2 : x1) + (static_cast<int64_t>(b) * C + c0) * plane;
    float *dst = (side == 0 ? gin1 : gin2) + (static_cast<int64_t>(b) * C + c0) * plane;
    const float *gob = gout + static_cast<int64_t>(b) * (kND * kND) * plane;

    // (two whole copies of the body per side rather than a branch around the step loop: state shared across such a branch
    // -- 32 accumulators, two x-row sets -- made the register allocator spill 106 registers)
    auto run = [&](auto sidec, auto latec) {
        constexpr int SIDE = decltype(sidec)::value;
        constexpr int LATE = decltype(latec)::value;
        StripBwd<K, SIDE> st;
        // (RAG: a wave whose channels lie past C gets an empty range -- every load returns zeros -- and stores nothing)
        st.rsrc_x = uniform_rsrc(src, (K::RAG ? max(0, min(K::CW, C - c0)) : K::CW) * plane * 4);
        st.Wr = Wr;
        st.lane_live = sx * 4 < Wr;
        st.rsrc_g = uniform_rsrc(gob, kND * kND * plane * 4);
        st.smem = smem;
        st.H = H; st.plane = plane; st.wave = wave; st.lane = lane; st.y0 = y0; st.rp = rp;
        st.lds_base = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) float *)smem));
        st.lds_lane = st.lds_base + rp * K::PAIRB + lr * K::PITCH + sx * 16;
        st.row0 = y0 + lr * K::NR - kD;
        st.voff0 = (st.row0 * Wr + sx * 4) * 4;
#pragma unroll
        for (int j = 0; j < K::NR; ++j)
#pragma unroll
            for (int c = 0; c < K::CW; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) st.acc[j][c][i] = 0.f;

        // A workgroup walks its ten x rows CYCLICALLY, starting where its neighbours are: at time
        // t it is at step S = (t - y0) mod 10, i.e. at image row y0 - 4 + S = t - 4 (mod 10) --
        // the five workgroups whose windows hold an image row all fetch it in the same time step,
        // one L2 miss and four hits.  (In natural order they fetch it two steps apart, and the
        // 2 x 3.2 MB an XCD moves in two steps do not fit its 4 MiB L2: 2.1x the bytes at the fabric.)
        const auto next = [](int S) { return S + 1 == 10 ? 0 : S + 1; };
        int S = (K::FLAGS & 32) ? 0 : (10 - y0 % 10) % 10;
        int S1 = next(S), S2 = next(S1), S3 = next(S2);
        st.issue_g(S, true, 0);
        st.load_x(S, true, st.xs[0]);
        st.issue_g(S1, true, 1);
        st.load_x(S1, true, st.xs[1]);
        st.issue_g(S2, true, 2);
        STRIP_STAMP(0);
#ifdef CERB_STAMP
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) g_strip_stamps[blockIdx.x][threadIdx.x >> 6][44] = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) g_strip_life[blockIdx.x][threadIdx.x >> 6][0] = __builtin_amdgcn_s_memrealtime();
#endif
        // the first step needs its own planes and x rows only: everything requested after them stays in flight
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"((K::FLAGS & 8) ? 0 : 2 * K::NDMA + K::CW) : "memory");
        STRIP_STAMP(1);
        st.patch_g(S, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        STRIP_STAMP(2);
        int cur = 0, nxt = 1, n2 = 2, n3 = 3;
#pragma unroll 1
        for (int t = 0; t < 10; t += 2) {
            st.template step<0, LATE>(S, S1, S2, S3, t + 2 < 10, t + 3 < 10, cur, nxt, n3, t);
            S = S1; S1 = S2; S2 = S3; S3 = next(S3);
            { const int q = cur; cur = nxt; nxt = n2; n2 = n3; n3 = q; }
            st.template step<1, LATE>(S, S1, S2, S3, t + 3 < 10, t + 4 < 10, cur, nxt, n3, t + 1);
            S = S1; S1 = S2; S2 = S3; S3 = next(S3);
            { const int q = cur; cur = nxt; nxt = n2; n2 = n3; n3 = q; }
        }

        STRIP_STAMP(43);
#ifdef CERB_STAMP
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) g_strip_stamps[blockIdx.x][threadIdx.x >> 6][45] = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) g_strip_life[blockIdx.x][threadIdx.x >> 6][1] = __builtin_amdgcn_s_memrealtime();
#endif
        const float inv = 1.0f / static_cast<float>(C);
#pragma unroll
        for (int j = 0; j < K::NR; ++j)
#pragma unroll
            for (int c = 0; c < K::CW; ++c) {
                float *p = dst + static_cast<int64_t>(c) * plane + (y0 + lr * K::NR + j) * Wr + sx * 4;
                const f4 r = f4{st.acc[j][c][0] * inv, st.acc[j][c][1] * inv, st.acc[j][c][2] * inv, st.acc[j][c][3] * inv};
                if constexpr (K::FLAGS & 256) { if (r[0] == 1.2345e-30f) *reinterpret_cast<f4 *>(p) = r; }   // experiment: no stores
                else if constexpr (K::RAG) {
                    if (st.lane_live && y0 + lr * K::NR + j < H && c0 + c < C) __builtin_nontemporal_store(r, reinterpret_cast<f4 *>(p));
                } else __builtin_nontemporal_store(r, reinterpret_cast<f4 *>(p));
            }
#ifdef CERB_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) g_strip_life[blockIdx.x][threadIdx.x >> 6][2] = __builtin_amdgcn_s_memrealtime();
#endif
    };
    constexpr int kLate = (K::FLAGS & 1024) ? 1 : (K::FLAGS & 2048) ? 3 : 0;
    using Early = std::integral_constant<int, 0>;
    using Late = std::integral_constant<int, kLate>;
    if (kLate != 0 && wave >= K::NWV / 2) {
        if (side == 0) run(std::integral_constant<int, 0>{}, Late{});
        else run(std::integral_constant<int, 1>{}, Late{});
    } else {
        if (side == 0) run(std::integral_constant<int, 0>{}, Early{});
        else run(std::integral_constant<int, 1>{}, Early{});
    }
    }
#endif
}

template <typename K>
int launch_strip(const char *name, const void *in1, const void *in2, const void *goutp, void *g1p,
                 void *g2p, const CorrGeom &g, hipStream_t s) {
    const int nyb = (g.H + K::ROWS_WG - 1) / K::ROWS_WG, ncb = (g.C + K::CWG - 1) / K::CWG;
    int64_t blocks = static_cast<int64_t>(g.B) * nyb * ncb * 2;
    if (blocks > 0x7fffffff) return CERB_ETOOLARGE;
    if (blocks % K::NIT) return CERB_EUNSUPPORTED;
    blocks /= K::NIT;
    static std::atomic<uint64_t> lds_done{0};
    int rc;
    if ((rc = ensure_lds(corr_bwd_d4_strip_kernel<K>, K::LDS_BYTES, &lds_done))) return rc;
    note_kernel(1, name);
    hipLaunchKernelGGL((corr_bwd_d4_strip_kernel<K>), dim3(static_cast<unsigned>(blocks)),
                       dim3(K::THREADS), K::LDS_BYTES, s, static_cast<const float *>(in1),
                       static_cast<const float *>(in2), static_cast<const float *>(goutp),
                       static_cast<float *>(g1p), static_cast<float *>(g2p), g.C, g.H, g.W, nyb, ncb);
    return launch_status();
}

}  // namespace

#ifdef CERB_STAMP
extern "C" int cerberus_debug_strip_occupancy(void) {
    using K = StripCfg<64, 8>;
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, corr_bwd_d4_strip_kernel<K>, K::THREADS, K::LDS_BYTES);
    return e == hipSuccess ? n : -static_cast<int>(e);
}
extern "C" int cerberus_debug_strip_life(void *dst, int bytes) {
    return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_strip_life), std::min<size_t>(bytes, sizeof(g_strip_life))));
}
extern "C" int cerberus_debug_strip_stamps(void *dst, int bytes) {
    return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_strip_stamps),
                                                std::min<size_t>(bytes, sizeof(g_strip_stamps))));
}
#endif

// fp32, pad = d = 4 (checked by the caller); returns CERB_EUNSUPPORTED for shapes it does not cover (W > 256 or
// W % 4 != 0).  W in {256, 128, 64} with H a multiple of the rows a workgroup owns and C a multiple of its channels
// take the exact configurations; every other W % 4 == 0 up to 256 the ragged ones (StripCfg RAG).
int corr_strip_backward(const void *in1, const void *in2, const void *gout, void *gin1, void *gin2,
                        const CorrGeom &g, hipStream_t s) {
    if (static_cast<int64_t>(kND * kND) * g.H * g.W >= (1ll << 29) ||
        static_cast<int64_t>(g.C) * g.H * g.W >= (1ll << 29))
        return CERB_EUNSUPPORTED;
#define CERB_STRIP(SPR, NWV, FL, NAME)                                                              \
    if (g.H % StripCfg<SPR, NWV, FL>::ROWS == 0 && g.C % StripCfg<SPR, NWV, FL>::CWG == 0)           \
        return launch_strip<StripCfg<SPR, NWV, FL>>(NAME, in1, in2, gout, gin1, gin2, g, s)
    if (g.W == 256) {
#ifdef CERB_ABLATE
        switch (option(OPT_CORR_BWD_CSLICE)) {   // timing experiments (tools/strip_exp.py): flags of StripCfg
            case 1: CERB_STRIP(64, 8, 1, "corr_bwd_d4_strip_w256_f1"); break;
            case 4: CERB_STRIP(64, 8, 4, "corr_bwd_d4_strip_w256_f4"); break;
            case 8: CERB_STRIP(64, 8, 8, "corr_bwd_d4_strip_w256_f8"); break;
            case 16: CERB_STRIP(64, 8, 16, "corr_bwd_d4_strip_w256_f16"); break;
            case 24: CERB_STRIP(64, 8, 24, "corr_bwd_d4_strip_w256_f24"); break;
            case 28: CERB_STRIP(64, 8, 28, "corr_bwd_d4_strip_w256_f28"); break;
            case 32: CERB_STRIP(64, 8, 32, "corr_bwd_d4_strip_w256_f32"); break;
            case 64: CERB_STRIP(64, 8, 64, "corr_bwd_d4_strip_w256_f64"); break;
            case 256: CERB_STRIP(64, 8, 256, "corr_bwd_d4_strip_w256_f256"); break;
            case 512: CERB_STRIP(64, 8, 512, "corr_bwd_d4_strip_w256_f512"); break;   // experiment: memory instructions inside the FMA stream
            case 516: CERB_STRIP(64, 8, 516, "corr_bwd_d4_strip_w256_f516"); break;
            case 1024: CERB_STRIP(64, 8, 1024, "corr_bwd_d4_strip_w256_f1024"); break;   // waves 4-7: DMAs before the FMAs
            case 2048: CERB_STRIP(64, 8, 2048, "corr_bwd_d4_strip_w256_f2048"); break;   // waves 4-7: DMAs and x rows before the FMAs
            default: break;
        }
#endif
#ifdef CERB_EXPERIMENTS
        // round 5, measured and rejected (profiles/r05_sol_skeleton.txt): every workgroup walks TWO items (both gradients of its
        // rows and channels) on half the grid -- item 1's stores drain under item 2 -- at the 2 waves per SIMD that leaves:
        // 41.9 vs 33.6 us at 4 pairs, 70.4 vs 66.3 at 8.  Test builds only.
        if (option(OPT_CORR_BWD_CSLICE) == 2 && g.H % 2 == 0 && g.C % 32 == 0)
            return launch_strip<StripCfg<64, 8, 0, 4, 2>>("corr_bwd_d4_strip_w256_2items", in1, in2, gout, gin1, gin2, g, s);
#endif
        // round 6, VERDICT r5 #4 (one bounded structural attempt): 16 waves = 2 row pairs x 8 channel groups, one workgroup per CU
        if (option(OPT_CORR_BWD_CSLICE) == 16 && g.H % 4 == 0 && g.C % 32 == 0)
            return launch_strip<StripCfg<64, 8, 0, 4, 1, 0, 2>>("corr_bwd_d4_strip_w256_2pairs", in1, in2, gout, gin1, gin2, g, s);
        CERB_STRIP(64, 8, 0, "corr_bwd_d4_strip_w256");
    } else if (g.W == 128) {
#ifdef CERB_ABLATE
        // occupancy experiments at the 128-wide level (256 eight-wave workgroups of 32 channels = 2 waves per SIMD at 4 pairs)
        if (option(OPT_CORR_BWD_CSLICE) == 2002 && g.H % 4 == 0 && g.C % 16 == 0)
            return launch_strip<StripCfg<32, 8, 0, 2>>("corr_bwd_d4_strip_w128_cw2", in1, in2, gout, gin1, gin2, g, s);
        if (option(OPT_CORR_BWD_CSLICE) == 2004 && g.H % 4 == 0 && g.C % 16 == 0)
            return launch_strip<StripCfg<32, 4, 0, 4>>("corr_bwd_d4_strip_w128_nwv4", in1, in2, gout, gin1, gin2, g, s);
#endif
        CERB_STRIP(32, 8, 0, "corr_bwd_d4_strip_w128");
    } else if (g.W == 64) {
        CERB_STRIP(16, 8, 0, "corr_bwd_d4_strip_w64");
        CERB_STRIP(16, 4, 0, "corr_bwd_d4_strip_w64_c16");
    }
#undef CERB_STRIP
    // ragged shapes (round 6): any W % 4 == 0 up to 256 on the lanes of the next power-of-two width, any H, any C
    // (whole groups of 4 channels per wave; a workgroup's unused channel groups idle)
    if (g.W % 4 == 0 && g.W <= 256) {
        if (g.W > 128) return launch_strip<StripCfg<64, 8, 0, 4, 1, 1>>("corr_bwd_d4_strip_rag256", in1, in2, gout, gin1, gin2, g, s);
        if (g.W > 64) return launch_strip<StripCfg<32, 8, 0, 4, 1, 1>>("corr_bwd_d4_strip_rag128", in1, in2, gout, gin1, gin2, g, s);
        if (g.C % 32 != 0 && g.C % 32 <= 16)
            return launch_strip<StripCfg<16, 4, 0, 4, 1, 1>>("corr_bwd_d4_strip_rag64_c16", in1, in2, gout, gin1, gin2, g, s);
        return launch_strip<StripCfg<16, 8, 0, 4, 1, 1>>("corr_bwd_d4_strip_rag64", in1, in2, gout, gin1, gin2, g, s);
    }
    return CERB_EUNSUPPORTED;
}

}  // namespace cerb
