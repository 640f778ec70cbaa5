// api.hip -- the extern "C" surface declared in include/cerberus_hip.h.
// Argument validation + dispatch only; kernels live in corr_d4.hip / corr_d4_bwd.hip / corr_strip.hip / corr_coarse.hip /
// corr_mfma.hip / corr_generic.hip, corr_grad_prep.hip, warp.hip, warp16.hip, warp_corr.hip and upsample.hip.
#include <atomic>
#include <cstring>

#include "common.h"

namespace cerb {
namespace {
// indexed by OptId (common.h)
const char *const g_option_names[OPT_COUNT] = {
    "corr_force_generic",   // 1: always use the generic kernels
    "corr_fwd_variant",     // 0: auto, 1..8: force a register-staged variant, 9..13: LDS-DMA, 14: matrix cores (16-bit), 15: coarse-level kernel, 16: auto without it, 17: persistent pipelined forward (-DCERB_EXPERIMENTS builds only), 20 / 26: the matrix-core forward register-staged / without the walk
    "corr_bwd_variant",     // 0: auto, 1: all-81 per lane, 2/3: 3 dy groups, 4/5: LDS-DMA, 6-9: dy-streaming, 10: column walk, 11: matrix cores, row per wave (16-bit; auto: segment per wave), 12/13: strip, 14/15: coarse-level kernel forced / off
    "corr_bwd_cslice",      // 0: auto, else channels per backward workgroup (matrix-core backward: tiles per column walk)
    "corr_no_mfma",         // 1: 16-bit storage never takes the matrix-core kernels (vector kernels, auto-selected)
    "warp_pair_taps",       // 0: default, 1: pairs everywhere, 2: none
    "warp_tile_ranges",     // 0: auto, else channel ranges per warp-backward tile
    "warp_tile_h",          // 0: auto, 8 / 16: rows per warp-backward tile
    "warp_force_scatter",   // 1: warp backward by global atomics (ATen's method) even with a context
    "warp_staged",          // 0: auto (warp gathers through an LDS window), 2: never, >= 4: that many channels per forward workgroup
    "warp_stagger",         // warp backward phase shift: 0 auto, -1 off, else delays (x 1024 cycles) of the 2nd / 3rd / 4th 256 workgroups, a byte each
    "warp_fewc",            // 0: auto (<= 4 channels without context / grad_image take the lane-per-pixel kernels), -1: off
    "warp_pair16",          // 0: auto (16-bit images take the two-elements-per-lane kernels of warp16.hip), -1: off
#ifdef CERB_ABLATE
    "corr_debug_ablate",    // timing ablation mask (WRONG results when != 0); ablation builds only
#endif
};
std::atomic<int> g_option_values[OPT_COUNT];
// process-wide (diagnostics): autograd runs the backward on its own thread, and the caller that asks
// "which kernel ran" sits on another one; names are string literals, so a relaxed pointer store is enough
std::atomic<const char *> g_last_kernel[2] = {{"none"}, {"none"}};

int find_option(const char *key) {
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!std::strcmp(g_option_names[i], key)) return i;
    return -1;
}

bool dtype_ok(int dtype) { return dtype >= CERB_F32 && dtype <= CERB_F64; }
}  // namespace

int option(OptId id) { return g_option_values[id].load(std::memory_order_relaxed); }
void note_kernel(int which, const char *name) { g_last_kernel[which & 1].store(name, std::memory_order_relaxed); }
}  // namespace cerb

using namespace cerb;

extern "C" {

int cerberus_abi_version(void) { return CERBERUS_HIP_ABI_VERSION; }

const char *cerberus_error_string(int code) {
    switch (code) {
        case CERB_OK: return "success";
        case CERB_EINVAL: return "invalid argument (null pointer, non-positive size or empty output)";
        case CERB_EDTYPE: return "unknown dtype";
        case CERB_ESTRIDE1: return "correlation backward requires stride1 == 1";
        case CERB_EMODE: return "unknown padding or interpolation mode";
        case CERB_EUNSUPPORTED: return "valid request that is not implemented";
        case CERB_ETOOLARGE: return "dimension exceeds launch/index limits";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown cerberus_hip error";
}

int cerberus_correlation_out_shape(int H, int W, int pad_size, int kernel_size,
                                   int max_displacement, int stride1, int stride2,
                                   int *out_channels, int *out_height, int *out_width) {
    if (!out_channels || !out_height || !out_width) return CERB_EINVAL;
    CorrGeom g;
    const int rc = corr_geom_init(g, 1, 1, H, W, pad_size, kernel_size, max_displacement, stride1,
                                  stride2);
    if (rc) return rc;
    *out_channels = g.oC; *out_height = g.oH; *out_width = g.oW;
    return CERB_OK;
}

int cerberus_correlation_forward_ex(const void *input1, const void *input2, void *output, int B,
                                    int C, int H, int W, int pad_size, int kernel_size,
                                    int max_displacement, int stride1, int stride2,
                                    float negative_slope, int64_t out_batch_stride, int dtype,
                                    void *stream) {
    if (!dtype_ok(dtype)) return CERB_EDTYPE;
    CorrGeom g;
    int rc = corr_geom_init(g, B, C, H, W, pad_size, kernel_size, max_displacement, stride1,
                            stride2);
    if (rc) return rc;
    if (B == 0) return CERB_OK;
    if (!input1 || !input2 || !output) return CERB_EINVAL;
    if (out_batch_stride != 0 &&
        out_batch_stride < static_cast<int64_t>(g.oC) * g.oH * g.oW)
        return CERB_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!option(OPT_CORR_FORCE_GENERIC)) {
        rc = corr_d4_forward(input1, input2, output, g, negative_slope, out_batch_stride, dtype, s);
        if (rc != CERB_EUNSUPPORTED) return rc;
    }
    note_kernel(0, "corr_fwd_generic");
    return corr_generic_forward(input1, input2, output, g, negative_slope, out_batch_stride, dtype,
                                s);
}

int cerberus_correlation_forward(const void *input1, const void *input2, void *output, int B,
                                 int C, int H, int W, int pad_size, int kernel_size,
                                 int max_displacement, int stride1, int stride2,
                                 int corr_type_multiply, int dtype, void *stream) {
    (void)corr_type_multiply;  // accepted and ignored, exactly like the reference
    return cerberus_correlation_forward_ex(input1, input2, output, B, C, H, W, pad_size,
                                           kernel_size, max_displacement, stride1, stride2, 1.0f,
                                           0, dtype, stream);
}

int cerberus_correlation_backward(const void *input1, const void *input2, const void *grad_output,
                                  void *grad_input1, void *grad_input2, int B, int C, int H,
                                  int W, int pad_size, int kernel_size, int max_displacement,
                                  int stride1, int stride2, int corr_type_multiply, int dtype,
                                  void *stream) {
    (void)corr_type_multiply;
    if (!dtype_ok(dtype)) return CERB_EDTYPE;
    CorrGeom g;
    int rc = corr_geom_init(g, B, C, H, W, pad_size, kernel_size, max_displacement, stride1,
                            stride2);
    if (rc) return rc;
    if (stride1 != 1) return CERB_ESTRIDE1;
    if (B == 0) return CERB_OK;
    if (!input1 || !input2 || !grad_output || !grad_input1 || !grad_input2) return CERB_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!option(OPT_CORR_FORCE_GENERIC)) {
        rc = corr_d4_backward(input1, input2, grad_output, grad_input1, grad_input2, g, dtype, s);
        if (rc != CERB_EUNSUPPORTED) return rc;
    }
    note_kernel(1, "corr_bwd_generic");
    return corr_generic_backward(input1, input2, grad_output, grad_input1, grad_input2, g, dtype,
                                 s);
}

int64_t cerberus_correlation_backward_ex_workspace_bytes(int B, int H, int W, int pad_size, int kernel_size,
                                                         int max_displacement, int stride1, int stride2, int dtype) {
    if (!dtype_ok(dtype)) return 0;
    CorrGeom g;
    if (corr_geom_init(g, B, 1, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)) return 0;
    const int64_t esz = dtype == CERB_F32 ? 4 : dtype == CERB_F64 ? 8 : 2;
    return static_cast<int64_t>(B) * g.oC * g.oH * g.oW * esz;
}

int cerberus_correlation_backward_ex(const void *input1, const void *input2, const void *grad_output,
                                     int64_t grad_out_batch_stride, const void *fwd_output,
                                     int64_t fwd_out_batch_stride, float negative_slope, void *workspace,
                                     int64_t workspace_bytes, void *grad_input1, void *grad_input2, int B, int C,
                                     int H, int W, int pad_size, int kernel_size, int max_displacement, int stride1,
                                     int stride2, int dtype, void *stream) {
    if (!dtype_ok(dtype)) return CERB_EDTYPE;
    CorrGeom g;
    int rc = corr_geom_init(g, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2);
    if (rc) return rc;
    if (stride1 != 1) return CERB_ESTRIDE1;
    if (B == 0) return CERB_OK;
    const int64_t item = static_cast<int64_t>(g.oC) * g.oH * g.oW;
    if ((grad_out_batch_stride != 0 && grad_out_batch_stride < item) ||
        (fwd_output && fwd_out_batch_stride != 0 && fwd_out_batch_stride < item))
        return CERB_EINVAL;
    const bool dense = B == 1 || grad_out_batch_stride == 0 || grad_out_batch_stride == item;   // one item: no stride to honour
    if (dense && !fwd_output)
        return cerberus_correlation_backward(input1, input2, grad_output, grad_input1, grad_input2, B, C, H, W, pad_size,
                                             kernel_size, max_displacement, stride1, stride2, 1, dtype, stream);
    if (!input1 || !input2 || !grad_output || !grad_input1 || !grad_input2) return CERB_EINVAL;
    const int64_t need = cerberus_correlation_backward_ex_workspace_bytes(B, H, W, pad_size, kernel_size, max_displacement,
                                                                          stride1, stride2, dtype);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return CERB_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = corr_grad_prep(grad_output, dense ? item : grad_out_batch_stride, fwd_output,
                        fwd_out_batch_stride ? fwd_out_batch_stride : item, workspace, B, item, negative_slope, dtype, s);
    if (rc) return rc;
    return cerberus_correlation_backward(input1, input2, workspace, grad_input1, grad_input2, B, C, H, W, pad_size,
                                         kernel_size, max_displacement, stride1, stride2, 1, dtype, stream);
}

static int warp_args_ok(int B, int C, int H, int W, int pad_mode, int interp_mode, int dtype) {
    if (!dtype_ok(dtype)) return CERB_EDTYPE;
    if (B < 0 || C <= 0 || H <= 0 || W <= 0) return CERB_EINVAL;
    if (pad_mode < CERB_PAD_ZEROS || pad_mode > CERB_PAD_REFLECTION) return CERB_EMODE;
    if (interp_mode < CERB_INTERP_BILINEAR || interp_mode > CERB_INTERP_NEAREST) return CERB_EMODE;
    return CERB_OK;
}

int cerberus_flow_warp_forward(const void *image, const void *flow, void *out, int B, int C, int H,
                               int W, int pad_mode, int interp_mode, int dtype, void *stream) {
    return cerberus_flow_warp_forward_ctx(image, flow, out, nullptr, 0, B, C, H, W, pad_mode,
                                          interp_mode, dtype, dtype, stream);
}

int64_t cerberus_flow_warp_context_bytes(int B, int H, int W) {
    if (B < 0 || H < 0 || W < 0) return 0;
    return warp_context_bytes(B, H, W);
}

int cerberus_flow_warp_forward_ctx(const void *image, const void *flow, void *out, void *context,
                                   int64_t context_bytes, int B, int C, int H, int W,
                                   int pad_mode, int interp_mode, int dtype, int flow_dtype,
                                   void *stream) {
    const int rc = warp_args_ok(B, C, H, W, pad_mode, interp_mode, dtype);
    if (rc) return rc;
    if (!dtype_ok(flow_dtype)) return CERB_EDTYPE;
    if (B == 0) return CERB_OK;
    if (!image || !flow || !out) return CERB_EINVAL;
    return warp_forward(image, flow, out, context, context_bytes, B, C, H, W, pad_mode,
                        interp_mode, dtype, flow_dtype, static_cast<hipStream_t>(stream));
}

int64_t cerberus_warp_correlation_workspace_bytes(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return warp_corr_workspace_bytes(B, C, H, W);
}

int cerberus_warp_correlation_forward(const void *input1, const void *input2, const void *flow, void *output, void *workspace,
                                      int64_t workspace_bytes, int B, int C, int H, int W, int pad_mode, float negative_slope,
                                      int64_t out_batch_stride, int dtype, int flow_dtype, void *stream) {
    const int rc = warp_args_ok(B, C, H, W, pad_mode, CERB_INTERP_BILINEAR, dtype);
    if (rc) return rc;
    if (!dtype_ok(flow_dtype)) return CERB_EDTYPE;
    if (pad_mode == CERB_PAD_REFLECTION) return CERB_EUNSUPPORTED;   // (no reference caller; the training path could not differentiate it)
    if (B == 0) return CERB_OK;
    if (!input1 || !input2 || !flow || !output) return CERB_EINVAL;
    if (out_batch_stride != 0 && out_batch_stride < static_cast<int64_t>(81) * H * W) return CERB_EINVAL;
    return warp_corr_forward(input1, input2, flow, output, workspace, workspace_bytes, B, C, H, W, pad_mode, negative_slope,
                             out_batch_stride, dtype, flow_dtype, static_cast<hipStream_t>(stream));
}

int64_t cerberus_flow_warp_backward_workspace_bytes(int B, int C, int H, int W) {
    if (B < 0 || C < 0 || H < 0 || W < 0) return 0;
    return warp_backward_workspace_bytes(B, C, H, W);
}

int cerberus_flow_warp_backward(const void *image, const void *flow, const void *grad_out,
                                void *grad_image, void *grad_flow, const void *context,
                                int64_t context_bytes, void *workspace, int64_t workspace_bytes,
                                int B, int C, int H, int W, int pad_mode, int interp_mode,
                                int dtype, int flow_dtype, void *stream) {
    const int rc = warp_args_ok(B, C, H, W, pad_mode, interp_mode, dtype);
    if (rc) return rc;
    if (!dtype_ok(flow_dtype)) return CERB_EDTYPE;
    if (B == 0) return CERB_OK;
    if (!image || !flow || !grad_out) return CERB_EINVAL;
    if (!grad_image && !grad_flow) return CERB_OK;
    return warp_backward(image, flow, grad_out, grad_image, grad_flow, context, context_bytes,
                         workspace, workspace_bytes, B, C, H, W, pad_mode, interp_mode, dtype,
                         flow_dtype, static_cast<hipStream_t>(stream));
}

static int upsample_entry(bool fwd, const void *src, void *dst, int64_t planes, int H, int W, int factor,
                          int dtype, void *stream) {
    if (!dtype_ok(dtype)) return CERB_EDTYPE;
    if (planes < 0 || H <= 0 || W <= 0 || factor < 1) return CERB_EINVAL;
    if (planes == 0) return CERB_OK;
    if (!src || !dst) return CERB_EINVAL;
    if (static_cast<int64_t>(H) * factor > 0x7fffffff || static_cast<int64_t>(W) * factor > 0x7fffffff)
        return CERB_ETOOLARGE;
    return flow_upsample(fwd, src, dst, planes, H, W, factor, dtype, static_cast<hipStream_t>(stream));
}

int cerberus_flow_upsample_forward(const void *src, void *dst, int64_t planes, int H, int W, int factor,
                                   int dtype, void *stream) {
    return upsample_entry(true, src, dst, planes, H, W, factor, dtype, stream);
}

int cerberus_flow_upsample_backward(const void *grad_out, void *grad_in, int64_t planes, int H, int W,
                                    int factor, int dtype, void *stream) {
    return upsample_entry(false, grad_out, grad_in, planes, H, W, factor, dtype, stream);
}

int cerberus_area_resize(const void *src, void *dst, int64_t planes, int H, int W, int out_h, int out_w,
                         int dtype, void *stream) {
    if (!dtype_ok(dtype)) return CERB_EDTYPE;
    if (planes < 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return CERB_EINVAL;
    if (planes == 0) return CERB_OK;
    if (!src || !dst) return CERB_EINVAL;
    return area_resize(src, dst, planes, H, W, out_h, out_w, dtype, static_cast<hipStream_t>(stream));
}

int cerberus_area_pyramid(const void *src, void *const *dsts, const int *out_h, const int *out_w, int n_scales,
                          int64_t planes, int H, int W, int dtype, void *stream) {
    if (!dtype_ok(dtype)) return CERB_EDTYPE;
    if (planes < 0 || H <= 0 || W <= 0 || n_scales < 0) return CERB_EINVAL;
    if (n_scales == 0 || planes == 0) return CERB_OK;
    if (!src || !dsts || !out_h || !out_w) return CERB_EINVAL;
    for (int i = 0; i < n_scales; ++i)
        if (!dsts[i] || out_h[i] <= 0 || out_w[i] <= 0) return CERB_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rc = area_pyramid(src, dsts, out_h, out_w, n_scales, planes, H, W, dtype, s);
    if (rc != CERB_EUNSUPPORTED) return rc;
    // general ratios (or more scales than one launch holds): scale by scale
    for (int i = 0; i < n_scales; ++i) {
        const int r = area_resize(src, dsts[i], planes, H, W, out_h[i], out_w[i], dtype, s);
        if (r) return r;
    }
    return CERB_OK;
}

int cerberus_set_option(const char *key, int value) {
    if (!key) return CERB_EINVAL;
    const int i = find_option(key);
    if (i < 0) return CERB_EINVAL;
    g_option_values[i].store(value, std::memory_order_relaxed);
    return CERB_OK;
}

int cerberus_get_option(const char *key, int *value) {
    if (!key || !value) return CERB_EINVAL;
    // read-only: 1 when the library was built with -DCERB_EXPERIMENTS (the measured-and-rejected
    // kernel variants the dispatcher never picks exist only in such test builds)
    if (!strcmp(key, "experiments_build")) {
#ifdef CERB_EXPERIMENTS
        *value = 1;
#else
        *value = 0;
#endif
        return CERB_OK;
    }
    const int i = find_option(key);
    if (i < 0) return CERB_EINVAL;
    *value = g_option_values[i].load(std::memory_order_relaxed);
    return CERB_OK;
}

const char *cerberus_last_kernel(int which) { return g_last_kernel[which & 1].load(std::memory_order_relaxed); }

}  // extern "C"
