/*
 * cerberus_hip.h -- C ABI of libcerberus_hip.so, the MI355X (gfx950) drop-in for
 * CerberusNet's cost-volume correlation + flow-warp hot path.
 *
 * Plain pointers and sizes only: no torch types, no C++ in the signatures.  All
 * tensors are dense, contiguous NCHW device buffers owned by the caller
 * (outputs included: the caller allocates, the library fully overwrites them).
 * Every entry point enqueues asynchronously on `stream` (a hipStream_t passed
 * as void*, NULL = default stream), never synchronises, allocates nothing
 * (scratch, where needed, is a caller-provided workspace), keeps no per-call state
 * (re-entrant: autograd may call backward from its own engine thread; the only process-wide
 * state is atomic: the tuning knobs below and a per-device "large LDS opted in" mask) and is
 * therefore safe to capture into a hipGraph.
 *
 * Return value: 0 on success; a negative CERB_E* code for rejected arguments;
 * a positive value is the hipError_t reported by the launch.
 * cerberus_error_string() turns either into text.
 *
 * Reference interfaces these replace (paths under /root/reference/):
 *   nnet_training/correlation_package/correlation_cuda.cpp:3-26   correlation_forward_cuda
 *   nnet_training/correlation_package/correlation_cuda.cpp:28-43  correlation_backward_cuda
 *   nnet_training/correlation_package/correlation_cuda_kernel.cuh:5-14 (kernel launchers)
 *   nnet_training/loss_functions/UnFlowLoss.py:83-94              flow_warp (-> ATen grid_sampler_2d fwd/bwd)
 * The Python binding that registers torch.ops.cerberus.{correlation,
 * correlation_backward} on top of this ABI is cerberusnet_amd/ops.py; the
 * reference-side stub a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef CERBERUS_HIP_H
#define CERBERUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CERBERUS_HIP_ABI_VERSION 7   /* 7: cerberus_warp_correlation_forward (f2), option warp_pair16 (round 6); 6: cerberus_correlation_backward_ex, cerberus_area_pyramid, option warp_fewc (round 5) */

/* element types (AT_DISPATCH_FLOATING_TYPES_AND_HALF in the reference,
 * correlation_cuda_kernel.cu:269,303; bf16 is an extension) */
enum cerb_dtype {
    CERB_F32  = 0,
    CERB_F16  = 1,  /* fp16 storage, fp32 accumulation (reference accumulates in fp16: Q6) */
    CERB_BF16 = 2,  /* bf16 storage, fp32 accumulation (not in the reference)             */
    CERB_F64  = 3
};

/* grid_sample modes used by flow_warp(image, flow12, pad='border', mode='bilinear') */
enum cerb_pad_mode    { CERB_PAD_ZEROS = 0, CERB_PAD_BORDER = 1, CERB_PAD_REFLECTION = 2 };
enum cerb_interp_mode { CERB_INTERP_BILINEAR = 0, CERB_INTERP_NEAREST = 1 };

/* argument-rejection codes (negative) */
#define CERB_OK            0
#define CERB_EINVAL       -1  /* null pointer, non-positive size, empty output          */
#define CERB_EDTYPE       -2  /* unknown dtype                                          */
#define CERB_ESTRIDE1     -3  /* correlation backward with stride1 != 1 (reference is   */
                              /* memory-unsafe there, correlation_cuda_kernel.cu:106)   */
#define CERB_EMODE        -4  /* unknown padding / interpolation mode                   */
#define CERB_EUNSUPPORTED -5  /* valid but not implemented (reflection-pad backward)    */
#define CERB_ETOOLARGE    -6  /* a dimension exceeds 32-bit launch / index limits       */

int cerberus_abi_version(void);

/* Human-readable text for a return code of any function below. */
const char *cerberus_error_string(int code);

/* Output geometry, exactly correlation_cuda.cpp:6-14:
 *   oC = ((d/s2)*2+1)^2, oH = ceil((H+2p-2(kr+d))/s1), oW likewise, kr=(k-1)/2. */
int cerberus_correlation_out_shape(int H, int W, int pad_size, int kernel_size,
                                   int max_displacement, int stride1,
                                   int stride2, int *out_channels,
                                   int *out_height, int *out_width);

/* correlation forward.  Replaces correlation_forward_cuda
 * (correlation_cuda.cpp:3-26 -> correlation_cuda_kernel.cu:244-324).
 *   input1,input2 : (B,C,H,W)      output : (B,oC,oH,oW), fully overwritten
 *   out[n][(tj+dr)*D+(ti+dr)][y][x] = 1/(k*k*C) * sum_{j,i in kernel} sum_c
 *        pad(in1)[n][c][y*s1+d+j][x*s1+d+i] * pad(in2)[n][c][y*s1+d+tj*s2+j][x*s1+d+ti*s2+i]
 * corr_type_multiply is accepted and ignored, as in the reference. */
int cerberus_correlation_forward(const void *input1, const void *input2,
                                 void *output, int B, int C, int H, int W,
                                 int pad_size, int kernel_size,
                                 int max_displacement, int stride1, int stride2,
                                 int corr_type_multiply, int dtype,
                                 void *stream);

/* Same, with the consumer's epilogue fused (the caller
 * pwcnet_sfd.py:181-187 applies leaky_relu(0.1) in place and concatenates):
 *   out = v > 0 ? v : v * negative_slope   (negative_slope = 1.0f -> identity)
 *   written at output + n*out_batch_stride elements (out_batch_stride = 0 means
 *   dense oC*oH*oW), so the 81 channels can land inside a wider concat buffer. */
int cerberus_correlation_forward_ex(const void *input1, const void *input2,
                                    void *output, int B, int C, int H, int W,
                                    int pad_size, int kernel_size,
                                    int max_displacement, int stride1,
                                    int stride2, float negative_slope,
                                    int64_t out_batch_stride, int dtype,
                                    void *stream);

/* correlation backward.  Replaces correlation_backward_cuda
 * (correlation_cuda.cpp:28-43 -> correlation_cuda_kernel.cu:326-429).
 *   grad_output : (B,oC,oH,oW)   grad_input1, grad_input2 : (B,C,H,W), fully
 *   overwritten (zeros where the reference's kernels early-return).
 *   Requires stride1 == 1 (CERB_ESTRIDE1 otherwise). */
int cerberus_correlation_backward(const void *input1, const void *input2,
                                  const void *grad_output, void *grad_input1,
                                  void *grad_input2, int B, int C, int H, int W,
                                  int pad_size, int kernel_size,
                                  int max_displacement, int stride1,
                                  int stride2, int corr_type_multiply,
                                  int dtype, void *stream);

/* Same, for the gradient of the buffer cerberus_correlation_forward_ex wrote into (round 5; the caller
 * pwcnet_sfd.py:181-187, seen from autograd: the cost volume is channels [0, oC) of the estimator's
 * concatenation buffer, with leaky_relu already applied):
 *   grad_output item n starts at grad_output + n * grad_out_batch_stride elements (0 = dense oC*oH*oW);
 *   its oC planes are contiguous.
 *   fwd_output (may be NULL): the volume the forward stored, item n at fwd_output + n *
 *   fwd_out_batch_stride (0 = dense).  The gradient the kernels see is
 *       g_eff = fwd_output > 0 ? g : g * negative_slope
 *   -- the derivative of the forward's fused LeakyReLU, taken from the stored value's sign (a positive
 *   slope keeps the sign; a NaN stored value takes the slope branch, as torch.where(out > 0, g, g * slope)).
 *   workspace: cerberus_correlation_backward_ex_workspace_bytes() bytes, 16-byte aligned, caller-owned,
 *   needed whenever the gradient is strided or masked (one pass writes the dense g_eff there; the backward
 *   kernels stream gradOutput by LDS-DMA and cannot apply a mask on the way).  With a dense gradient and
 *   fwd_output == NULL this is cerberus_correlation_backward and the workspace is not touched. */
int64_t cerberus_correlation_backward_ex_workspace_bytes(int B, int H, int W, int pad_size,
                                                         int kernel_size, int max_displacement,
                                                         int stride1, int stride2, int dtype);
int cerberus_correlation_backward_ex(const void *input1, const void *input2,
                                     const void *grad_output, int64_t grad_out_batch_stride,
                                     const void *fwd_output, int64_t fwd_out_batch_stride,
                                     float negative_slope, void *workspace, int64_t workspace_bytes,
                                     void *grad_input1, void *grad_input2, int B, int C, int H, int W,
                                     int pad_size, int kernel_size, int max_displacement, int stride1,
                                     int stride2, int dtype, void *stream);

/* flow_warp forward.  Replaces the body of flow_warp (UnFlowLoss.py:83-94):
 * mesh_grid + flow -> norm_grid by (W-1),(H-1) -> grid_sample(align_corners=False),
 * fused: no grid tensor is materialised.
 *   image : (B,C,H,W)  flow : (B,2,H,W) (ch0 = x, ch1 = y, pixels)  out : (B,C,H,W) */
int cerberus_flow_warp_forward(const void *image, const void *flow, void *out,
                               int B, int C, int H, int W, int pad_mode,
                               int interp_mode, int dtype, void *stream);

/* The same forward, additionally saving what the backward needs again (what autograd's
 * save_for_backward is to the reference's grid_sample): every pixel's sample position and,
 * per 64-pixel strip, the signed range of tap displacements.
 *   context    : caller-owned device buffer of cerberus_flow_warp_context_bytes(B,H,W) bytes
 *                (16-byte aligned, contents irrelevant, fully written); NULL = plain forward.
 *                The layout is private to the library (an opaque blob between the two calls).
 *   flow_dtype : element type of `flow`: equal to `dtype`, or CERB_F32 with a 16-bit image
 *                (under autocast the reference's grid_sample runs in fp32 on whatever
 *                precision the flow arrives in: an fp32 flow is not rounded to 16 bits). */
int64_t cerberus_flow_warp_context_bytes(int B, int H, int W);
int cerberus_flow_warp_forward_ctx(const void *image, const void *flow, void *out,
                                   void *context, int64_t context_bytes, int B, int C,
                                   int H, int W, int pad_mode, int interp_mode,
                                   int dtype, int flow_dtype, void *stream);

/* f2 of SURVEY.md section 8(f): flow_warp FUSED into the correlation forward -- what the reference's head computes at
 * nnet_models/pwcnet_sfd.py:178 -> :181-182 with pad_size = max_displacement = 4, kernel_size = stride1 = stride2 = 1:
 *   output[b][(dy+4)*9 + (dx+4)][y][x] = leaky( mean_c input1[b][c][y][x] * warped[b][c][y+dy][x+dx] ),
 *   warped = flow_warp(input2, flow, pad_mode, bilinear), zero outside the image,
 * without ever writing `warped` (no round trip, nothing to save for the backward: the training path recomputes the warp).
 * Same arithmetic as the two stand-alone calls (the warp's coordinate rounding order; a 16-bit warped value is rounded through
 * the storage type as the stand-alone warp's output is); the channel sum runs in another order (fp32 rounding).
 *   input1, input2 : (B,C,H,W), dtype (CERB_F32 / F16 / BF16);  flow : (B,2,H,W), flow_dtype (= dtype, or CERB_F32)
 *   output         : (B,81,H,W), dtype; batch stride out_batch_stride elements (0 = dense) as in cerberus_correlation_forward_ex
 *   negative_slope : LeakyReLU slope applied to the result (1.0f = none)
 *   workspace      : caller-owned device scratch of cerberus_warp_correlation_workspace_bytes(B,C,H,W) bytes (4-byte aligned,
 *                    contents irrelevant), or NULL.  Only small maps need it (fewer than 256 tiles of 8 x 32 pixels: the channel
 *                    sum is then split over several workgroups per tile, summed with float atomics in an fp32 volume and
 *                    finished by a second launch: results agree to fp32 rounding run to run, not bit for bit); without it
 *                    such a map takes the one-launch form (bit-reproducible, few workgroups).
 * Built in round 6 so that the row exists as code; measured SLOWER than the two tuned launches (bench.py extra.f2_fused): opt-in. */
int64_t cerberus_warp_correlation_workspace_bytes(int B, int C, int H, int W);
int cerberus_warp_correlation_forward(const void *input1, const void *input2, const void *flow, void *output, void *workspace,
                                      int64_t workspace_bytes, int B, int C, int H, int W, int pad_mode, float negative_slope,
                                      int64_t out_batch_stride, int dtype, int flow_dtype, void *stream);

/* flow_warp backward (autograd of the above w.r.t. image and flow).
 *   grad_image : (B,C,H,W), dtype -- fully overwritten.  With a context or a workspace
 *                (fp32 / fp16 / bf16): built tile by tile in LDS in 64-bit fixed point, no
 *                global atomics, bit-reproducible; NaN / Inf in grad_out reach the same
 *                elements as in ATen's scatter.  With neither (or fp64): global float atomics
 *                as ATen's (summation order not fixed).
 *   grad_flow  : (B,2,H,W), flow_dtype -- fully overwritten, deterministic.  With a context or a workspace
 *                it comes from the LDS-window role of the tile launch -- round 6: also when grad_image is NULL
 *                and C > 4 (the bits are the both-gradients call's); with neither, from per-pixel gathers
 *                (another channel-group order: equal within rounding)
 *   context    : the buffer a cerberus_flow_warp_forward_ctx call with the SAME flow,
 *                shape and pad_mode filled, or NULL (the backward then derives it from the
 *                flow with one extra launch, into the workspace).
 *   workspace  : caller-owned device scratch of at least
 *                cerberus_flow_warp_backward_workspace_bytes(B,C,H,W) bytes (16-byte
 *                aligned, contents irrelevant), private to this call until it completes;
 *                only needed when context is NULL.
 * Either grad pointer may be NULL to skip that gradient. */
int64_t cerberus_flow_warp_backward_workspace_bytes(int B, int C, int H, int W);
int cerberus_flow_warp_backward(const void *image, const void *flow,
                                const void *grad_out, void *grad_image,
                                void *grad_flow, const void *context,
                                int64_t context_bytes, void *workspace,
                                int64_t workspace_bytes, int B, int C, int H, int W,
                                int pad_mode, int interp_mode, int dtype,
                                int flow_dtype, void *stream);

/* The flow pyramid's upsampling step, fused.  Replaces
 *   F.interpolate(flow * factor, scale_factor=factor, mode='bilinear', align_corners=True)
 * (nnet_training/nnet_models/pwcnet_sfd.py:176 with factor 2, :199-201 with factor 4): one launch
 * instead of a multiply + ATen upsample_bilinear2d; the backward is a deterministic gather
 * instead of ATen's float-atomic scatter.
 *   forward : src (planes, H, W) -> dst (planes, H*factor, W*factor)
 *   backward: src = grad of the upsampled tensor (planes, H*factor, W*factor) -> dst (planes, H, W)
 * planes = B * channels of the contiguous NCHW tensor; factor >= 1; fp32 / fp16 / bf16. */
int cerberus_flow_upsample_forward(const void *src, void *dst, int64_t planes, int H, int W,
                                   int factor, int dtype, void *stream);
int cerberus_flow_upsample_backward(const void *grad_out, void *grad_in, int64_t planes, int H,
                                    int W, int factor, int dtype, void *stream);

/* The photometric loss's image pyramid.  Replaces
 *   F.interpolate(image, (out_h, out_w), mode='area')
 * (nnet_training/loss_functions/UnFlowLoss.py:279-280: the target images resized to every flow
 * scale) = ATen adaptive_avg_pool2d: output (oy, ox) is the mean over rows
 * [floor(oy*H/out_h), ceil((oy+1)*H/out_h)) x the same in x, summed in fp32 in row-major order
 * and divided by the window height, then width (bit-identical to torch's CPU kernel in fp32).
 *   src (planes, H, W) -> dst (planes, out_h, out_w); any sizes >= 1; fp32 / fp16 / bf16.
 * Forward only: the reference applies it to target images, which carry no gradient. */
int cerberus_area_resize(const void *src, void *dst, int64_t planes, int H, int W, int out_h,
                         int out_w, int dtype, void *stream);

/* All scales of that pyramid in one call (round 5): dsts[i] (planes, out_h[i], out_w[i]) for i < n_scales
 * (dsts, out_h, out_w are HOST arrays; dsts[i] device pointers).  When every scale is an integer ratio r in
 * {2, 4, 8, 16, 32, 64} of the source (and n_scales <= 4) ONE launch reads the source once and writes every
 * scale (unFlowLoss resizes each 25 MB target image to every flow scale: UnFlowLoss.py:279-280); otherwise
 * the scales are resized one by one.  Results are bit-identical to cerberus_area_resize either way. */
int cerberus_area_pyramid(const void *src, void *const *dsts, const int *out_h, const int *out_w,
                          int n_scales, int64_t planes, int H, int W, int dtype, void *stream);

/* Diagnostics / tuning knobs (process-wide, read at launch time, default 0):
 *   "corr_force_generic" : 1 = always use the generic kernels (testing)
 *   "corr_fwd_variant"   : 0 = auto, 1..8 = force one register-staged forward variant,
 *                          9..13 = the LDS-DMA variants (fp32, W % 4 == 0) with 1, 2, 4, 8,
 *                          16 channel groups, 14 = the matrix-core kernel (fp16 / bf16 storage,
 *                          C <= 128; auto uses it for 16 < C <= 128), 15 = the coarse-level kernel (fp32, and
 *                          fp16 / bf16 storage with C > 128; any even W up to 64 -- round 6 -- and a channel count its lane
 *                          layout divides: corr_coarse.hip; auto uses it there up to 2560 (row, displacement
 *                          row) workgroups), 16 = auto without it, 17 = the persistent, cross-item pipelined forward
 *                          (fp32, C % 8 == 0: corr_fwd_pipe.hip; built and measured in round 5, slower than the tile
 *                          kernels: -DCERB_EXPERIMENTS builds only since round 6; with it, "corr_bwd_cslice" > 0 sets
 *                          its number of workgroups); round 6, 16-bit storage with W % 8 == 0: 16 < C <= 64 take the
 *                          column-walk form of the matrix-core forward (LDS-DMA tiles, ds_read_b64_tr_b16 operands;
 *                          corr_mfma.hip) -- 20 = the register-staged form of rounds 4-5 instead (what other widths and
 *                          65 .. 128 channels use), 26 = the walk's stand-still form (4 x 64 tiles, C <= 32); all three
 *                          give identical bits; with them "corr_bwd_cslice" > 0 sets the tiles per walk
 *   "corr_bwd_variant"   : 0 = auto, 1 = all 81 displacements per lane (register-staged),
 *                          3 = three displacement groups, 4 / 5 = LDS-DMA with the 8x64 /
 *                          16x32 tile (fp32, W % 4 == 0), 8 = displacement-row streaming,
 *                          11 = the matrix-core kernel in its row-per-wave form of rounds 2-4 (fp16 / bf16 storage; auto
 *                          uses the segment-per-wave form of round 5: same bits, 8-12 % faster),
 *                          12 = whole image rows per wavefront (fp32; round 6: any W % 4 == 0 up to 256, any
 *                          H and C -- widths between 64 / 128 / 256 run on the lanes of the next one; auto uses
 *                          it on maps wider than 64 with enough workgroups for the chip and on exact 64-wide
 *                          ones; 13 = auto, but not on 64-wide maps), 14 = the coarse-level kernel (fp32, round 6:
 *                          any W % 4 == 0 up to 64, corr_coarse.hip; auto uses it there up to 4096 workgroups), 15 = auto without
 *                          it; 2, 6, 7, 9, 10 (and forward
 *                          1, 2, 8) are measured-and-rejected variants that exist only in
 *                          -DCERB_EXPERIMENTS test builds (otherwise: auto)
 *                          NOTE on batch invariance: auto picks kernels from WORKGROUP COUNTS (the 2560 /
 *                          4096 / 192-workgroup thresholds above), which include the batch size, and the
 *                          kernels differ in summation order: a batch item's correlation values (and
 *                          forward of a stacked batch vs two separate calls) agree to fp32 rounding
 *                          (<= 1e-6 relative), not bit for bit, across batch sizes that change the kernel.
 *                          The warp ops choose from one image's shape only, except the grad_image tile
 *                          height (8 rows up to 32768 pixels per call, else 16), which moves the
 *                          fixed-point scale by <= 2^-29 of a tile's largest gradient.
 *   "experiments_build"  : read-only (cerberus_get_option): 1 in a -DCERB_EXPERIMENTS build
 *   "corr_bwd_cslice"    : 0 = auto, else channels per backward workgroup (the matrix-core
 *                          backward reads it as the number of tiles a workgroup walks down)
 *   "corr_no_mfma"       : 1 = fp16 / bf16 storage never takes the matrix-core kernels: the vector
 *                          kernels, selected as for fp32 (the fp32 path uses no MFMA either way)
 *   "warp_pair_taps"     : warp gather variant (0 default, 1 paired everywhere, 2 unpaired)
 *   "warp_tile_ranges"   : channel ranges per warp-backward tile (0 auto)
 *   "warp_tile_h"        : rows per warp-backward tile (0 auto, 8, 16)
 *   "warp_force_scatter" : 1 = warp backward by global atomics even when a context exists
 *   "warp_staged"        : 0 = auto (the forward gather goes through an LDS copy of the source
 *                          window on large maps with 16-byte aligned rows), 2 = always gather
 *                          from global memory, >= 4 = always staged where possible, with that
 *                          many channels per workgroup
 *   "warp_stagger"       : phase shift of the warp backward's co-resident workgroups (round 4): 0 = auto (when
 *                          the whole launch is resident at once: the second 16-row tile workgroup of every CU
 *                          starts 6 k cycles late, or, with one tile workgroup per CU, the two halves of the
 *                          grad_flow workgroups 6 k and 2 k cycles late), -1 = off, else the delays of the 2nd / 3rd / 4th
 *                          256 workgroups in units of 1024 cycles, one byte each.  Speed only: same results.
 *   "warp_fewc"          : 0 = auto (a warp of <= 4 channels takes the lane-per-pixel kernels when no context /
 *                          no grad_image is asked for: the photometric loss's RGB warps), -1 = off.  Same bits.
 *   "warp_pair16"        : (round 6) 0 = auto: fp16 / bf16 images with W % 8 == 0 take the kernels of warp16.hip --
 *                          forward and the backward's grad_flow role with two pixels per lane and a raw 16-bit LDS
 *                          window filled by LDS-DMA, the backward's tile role with its sources as pairs (W % 2 == 0);
 *                          -1 = off (the general kernels); 1 = additionally the fp32 forward through the same
 *                          window kernel (measured slower than the staged one: for A/B only).  Same bits in every case.
 * Returns CERB_EINVAL for an unknown key. */
int cerberus_set_option(const char *key, int value);
int cerberus_get_option(const char *key, int *value);

/* Name of the kernel variant the most recent correlation forward / backward of this PROCESS
 * dispatched to, whichever thread issued it (autograd runs backward on its own thread); for
 * tests and the bench's roofline report. */
const char *cerberus_last_kernel(int which /*0 = forward, 1 = backward*/);

#ifdef __cplusplus
}
#endif
#endif /* CERBERUS_HIP_H */
