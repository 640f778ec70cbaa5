// corr_grad_prep.hip -- gradOutput as the correlation backward kernels read it, from what the flow head's autograd
// hands over (cerberus_correlation_backward_ex).
//
// The head writes LeakyReLU(correlation) straight into the estimator's concatenation buffer
// (cerberus_correlation_forward_ex; reference: pwcnet_sfd.py:181-187 builds it in three passes).  On the way back
// autograd delivers the gradient of that whole buffer, (B, 81 + others, H, W): the cost volume's 81 planes of a batch
// item are contiguous, batch items are not, and the LeakyReLU derivative -- 1 where the stored volume is positive,
// the slope elsewhere -- still has to be applied.  In stock ops that is `g * slope`, `torch.where(out > 0, g, ...)`
// and a `.contiguous()`: 255 MB of traffic at the 32 x 128 x 256 level for a 42.5 MB gradient, more than the
// correlation backward itself moves.  Here ONE pass reads the strided gradient and the stored volume and writes the
// dense, masked gradient into the caller's workspace (127 MB), with plain stores so that the backward launched
// right behind it finds it in L2 / the Infinity Cache.  (Applying the mask INSIDE the backward kernels was priced
// and not built: both gradients' workgroups stream gradOutput by LDS-DMA, which has no ALU on the way; masking in
// LDS needs the stored volume in LDS too -- twice the ring, one workgroup per CU -- and reading it once per side
// moves 170 MB where this pass plus the two dense reads move 212 MB: the same order, for ~10 % more VALU work in a
// loop that is VALU-bound)
#include "common.h"

namespace cerb {
namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<__half> { typedef uint2 type; };
template <> struct Vec4<hip_bfloat16> { typedef uint2 type; };
template <> struct Vec4<double> { typedef double4 type; };

template <typename T>
__device__ __forceinline__ T masked(T g, T v, float slope) {
    using A = typename Acc<T>::type;
    // torch.where(out > 0, g, g * slope): a NaN in the stored volume takes the slope branch
    const A gv = ld(&g), vv = ld(&v);
    T r;
    st(&r, vv > A(0) ? gv : gv * static_cast<A>(slope));
    return r;
}

// item n: gout + n * g_stride, fwd + n * f_stride (elements), `count` contiguous elements each
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void corr_grad_prep_kernel(const T *__restrict__ gout, int64_t g_stride,
                                                             const T *__restrict__ fwd, int64_t f_stride,
                                                             T *__restrict__ dst, int64_t count, float slope) {
    const int b = blockIdx.y;
    const T *g = gout + b * g_stride;
    const T *f = fwd ? fwd + b * f_stride : nullptr;
    T *d = dst + b * count;
    if constexpr (VEC) {
        typedef typename Vec4<T>::type V;
        const int64_t n4 = count / 4;
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += static_cast<int64_t>(gridDim.x) * 256) {
            V gv = reinterpret_cast<const V *>(g)[i];
            if (f) {
                const V fv = reinterpret_cast<const V *>(f)[i];
                T ge[4], fe[4];
                __builtin_memcpy(ge, &gv, sizeof(V));
                __builtin_memcpy(fe, &fv, sizeof(V));
#pragma unroll
                for (int k = 0; k < 4; ++k) ge[k] = masked<T>(ge[k], fe[k], slope);
                __builtin_memcpy(&gv, ge, sizeof(V));
            }
            reinterpret_cast<V *>(d)[i] = gv;
        }
    } else {
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < count; i += static_cast<int64_t>(gridDim.x) * 256)
            d[i] = f ? masked<T>(g[i], f[i], slope) : g[i];
    }
}

template <typename T>
int launch_prep(const void *gout, int64_t g_stride, const void *fwd, int64_t f_stride, void *dst, int B, int64_t count,
                float slope, hipStream_t s) {
    const uintptr_t align = 4 * sizeof(T) - 1;
    const bool vec = count % 4 == 0 && g_stride % 4 == 0 && (!fwd || f_stride % 4 == 0) &&
                     !(reinterpret_cast<uintptr_t>(gout) & align) && !(reinterpret_cast<uintptr_t>(fwd) & align) &&
                     !(reinterpret_cast<uintptr_t>(dst) & align);
    const int64_t work = vec ? count / 4 : count;
    // ~8 vectors per thread: 1296 workgroups per item at the 81 x 128 x 256 level
    const unsigned bx = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((work + 2047) / 2048, 65535)));
    const dim3 grid(bx, static_cast<unsigned>(B));
    if (vec)
        hipLaunchKernelGGL((corr_grad_prep_kernel<T, true>), grid, dim3(256), 0, s, static_cast<const T *>(gout), g_stride,
                           static_cast<const T *>(fwd), f_stride, static_cast<T *>(dst), count, slope);
    else
        hipLaunchKernelGGL((corr_grad_prep_kernel<T, false>), grid, dim3(256), 0, s, static_cast<const T *>(gout), g_stride,
                           static_cast<const T *>(fwd), f_stride, static_cast<T *>(dst), count, slope);
    return launch_status();
}

}  // namespace

int corr_grad_prep(const void *gout, int64_t g_stride, const void *fwd, int64_t f_stride, void *dst, int B, int64_t count,
                   float slope, int dtype, hipStream_t s) {
    if (B > 65535) return CERB_ETOOLARGE;
    switch (dtype) {
        case CERB_F32: return launch_prep<float>(gout, g_stride, fwd, f_stride, dst, B, count, slope, s);
        case CERB_F16: return launch_prep<__half>(gout, g_stride, fwd, f_stride, dst, B, count, slope, s);
        case CERB_BF16: return launch_prep<hip_bfloat16>(gout, g_stride, fwd, f_stride, dst, B, count, slope, s);
        case CERB_F64: return launch_prep<double>(gout, g_stride, fwd, f_stride, dst, B, count, slope, s);
        default: return CERB_EDTYPE;
    }
}

}  // namespace cerb
