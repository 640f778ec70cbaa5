"""torch.ops.cerberus.* -- the reference's operator names and schemas, bound to
the HIP kernels through the C ABI (include/cerberus_hip.h).

Reference registration being replaced
(/root/reference/nnet_training/correlation_package/correlation_cuda.cpp:45-48):

    TORCH_LIBRARY(cerberus, m) {
        m.def("correlation", correlation_forward_cuda);            // :3-26
        m.def("correlation_backward", correlation_backward_cuda);  // :28-43
    }

Same names, same argument order (the ONNX symbolic in utilities/onnx_export.py:18-23
depends on it), same ownership (the op allocates and returns its outputs), same
threading contract (enqueue on the current stream, never synchronise).  Extra
ops ``cerberus::flow_warp`` / ``flow_warp_backward`` / ``correlation_leaky``
carry the fused warp (UnFlowLoss.py:83-94) and the fused LeakyReLU epilogue
(pwcnet_sfd.py:181-182).

CPU tensors are rejected with RuntimeError: the product has no CPU path.
PyTorch is plumbing here (device memory, streams, autograd graph) -- the
arithmetic is all in libcerberus_hip.so.
"""
import ctypes
from typing import List

import torch
from torch.library import Library

from . import _lib

_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}
PAD_MODES = {"zeros": 0, "border": 1, "reflection": 2}
INTERP_MODES = {"bilinear": 0, "nearest": 1}

_CORR_ARGS = ("int pad_size, int kernel_size, int max_displacement, "
              "int stride1, int stride2, int corr_type_multiply")

_def = Library("cerberus", "DEF")
_def.define("correlation(Tensor input1, Tensor input2, %s) -> Tensor" % _CORR_ARGS)
_def.define("correlation_backward(Tensor input1, Tensor input2, Tensor gradOutput, %s) "
            "-> Tensor[]" % _CORR_ARGS)
_def.define("correlation_leaky(Tensor input1, Tensor input2, %s, float negative_slope) "
            "-> Tensor" % _CORR_ARGS)
_def.define("correlation_leaky_into(Tensor(a!) buffer, Tensor input1, Tensor input2, int channel_offset, "
            "%s, float negative_slope) -> ()" % _CORR_ARGS)
_def.define("correlation_backward_leaky(Tensor input1, Tensor input2, Tensor grad_buffer, Tensor fwd_buffer, "
            "int channel_offset, %s, float negative_slope) -> Tensor[]" % _CORR_ARGS)
_def.define("warp_correlation_leaky(Tensor input1, Tensor input2, Tensor flow, int pad_mode, float negative_slope) -> Tensor")
_def.define("area_pyramid(Tensor image, int[] sizes) -> Tensor[]")
_def.define("flow_upsample(Tensor flow, int factor) -> Tensor")
_def.define("flow_upsample_backward(Tensor grad_out, int factor) -> Tensor")
_def.define("area_resize(Tensor image, int out_h, int out_w) -> Tensor")
_def.define("flow_warp(Tensor image, Tensor flow, int pad_mode, int interp_mode) -> Tensor")
_def.define("flow_warp_ctx(Tensor image, Tensor flow, int pad_mode, int interp_mode) -> "
            "(Tensor, Tensor)")
_def.define("flow_warp_backward_ctx(Tensor image, Tensor flow, Tensor context, Tensor grad_out, "
            "int pad_mode, int interp_mode, bool need_image, bool need_flow) -> Tensor[]")
_def.define("flow_warp_backward(Tensor image, Tensor flow, Tensor grad_out, int pad_mode, "
            "int interp_mode, bool need_image, bool need_flow) -> Tensor[]")


def _stream_ptr(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dtype_code(t: torch.Tensor, what: str) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError("%s: unsupported dtype %s (float32/float16/bfloat16/float64)"
                           % (what, t.dtype)) from None


def _check_pair(a: torch.Tensor, b: torch.Tensor, what: str):
    if a.dim() != 4 or b.dim() != 4:
        raise RuntimeError("%s: expected 4-D NCHW tensors, got %s and %s"
                           % (what, tuple(a.shape), tuple(b.shape)))
    if a.shape != b.shape:
        raise RuntimeError("%s: input shapes differ: %s vs %s"
                           % (what, tuple(a.shape), tuple(b.shape)))
    if a.dtype != b.dtype:
        raise RuntimeError("%s: input dtypes differ: %s vs %s" % (what, a.dtype, b.dtype))
    if a.device != b.device:
        raise RuntimeError("%s: inputs on different devices: %s vs %s"
                           % (what, a.device, b.device))


def _corr_out_shape(H, W, pad, k, d, s1, s2):
    lib = _lib.get()
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.cerberus_correlation_out_shape(H, W, pad, k, d, s1, s2, ctypes.byref(oc),
                                                  ctypes.byref(oh), ctypes.byref(ow)),
               "cerberus::correlation (output shape)")
    return oc.value, oh.value, ow.value


def _meta_out_shape(H, W, pad, k, d, s1, s2):
    # shape maths of correlation_cuda.cpp:6-14 without touching the library
    import math
    kr = (k - 1) // 2
    border = kr + d
    oc = ((d // s2) * 2 + 1) ** 2
    oh = math.ceil((H + 2 * pad - 2 * border) / s1)
    ow = math.ceil((W + 2 * pad - 2 * border) / s1)
    return oc, oh, ow


# ----------------------------------------------------------------------------
# correlation forward
# ----------------------------------------------------------------------------
def _correlation_impl(input1, input2, pad, k, d, s1, s2, mult, slope, what):
    _check_pair(input1, input2, what)
    code = _dtype_code(input1, what)
    x1 = input1.contiguous()   # the reference honours strides via accessors (.cu:271-272)
    x2 = input2.contiguous()
    B, C, H, W = x1.shape
    oc, oh, ow = _corr_out_shape(H, W, pad, k, d, s1, s2)
    out = torch.empty((B, oc, oh, ow), dtype=x1.dtype, device=x1.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device(x1.device):
        rc = _lib.get().cerberus_correlation_forward_ex(
            x1.data_ptr(), x2.data_ptr(), out.data_ptr(), B, C, H, W, pad, k, d, s1, s2,
            ctypes.c_float(slope), 0, code, _stream_ptr(x1))
    _lib.check(rc, what)
    return out


def _correlation_cuda(input1, input2, pad_size, kernel_size, max_displacement, stride1,
                      stride2, corr_type_multiply):
    return _correlation_impl(input1, input2, pad_size, kernel_size, max_displacement, stride1,
                             stride2, corr_type_multiply, 1.0, "cerberus::correlation")


def _correlation_leaky_cuda(input1, input2, pad_size, kernel_size, max_displacement, stride1,
                            stride2, corr_type_multiply, negative_slope):
    return _correlation_impl(input1, input2, pad_size, kernel_size, max_displacement, stride1,
                             stride2, corr_type_multiply, float(negative_slope),
                             "cerberus::correlation_leaky")


def _correlation_leaky_into_cuda(buffer, input1, input2, channel_offset, pad_size, kernel_size,
                                 max_displacement, stride1, stride2, corr_type_multiply,
                                 negative_slope):
    """The cost volume (with the fused LeakyReLU) written straight into channels
    [channel_offset, channel_offset + oC) of a wider, caller-owned NCHW buffer: the
    ``torch.cat([out_corr, im1_1by1, flow])`` of pwcnet_sfd.py:186-187 without the pass that
    re-reads and re-writes the 81-channel volume."""
    what = "cerberus::correlation_leaky_into"
    _check_pair(input1, input2, what)
    code = _dtype_code(input1, what)
    x1, x2 = input1.contiguous(), input2.contiguous()
    B, C, H, W = x1.shape
    oc, oh, ow = _corr_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    if (buffer.dim() != 4 or not buffer.is_contiguous() or buffer.dtype != x1.dtype or
            buffer.device != x1.device or buffer.shape[0] != B or
            tuple(buffer.shape[2:]) != (oh, ow) or channel_offset < 0 or
            channel_offset + oc > buffer.shape[1]):
        raise RuntimeError("%s: buffer %s (%s, contiguous=%s) cannot hold channels [%d, %d) of a "
                           "(%d, %d, %d, %d) %s cost volume"
                           % (what, tuple(buffer.shape), buffer.dtype, buffer.is_contiguous(),
                              channel_offset, channel_offset + oc, B, oc, oh, ow, x1.dtype))
    if B * oc * oh * ow == 0:
        return
    first = buffer[:, channel_offset]          # address of (0, channel_offset, 0, 0)
    with torch.cuda.device(x1.device):
        rc = _lib.get().cerberus_correlation_forward_ex(
            x1.data_ptr(), x2.data_ptr(), first.data_ptr(), B, C, H, W, pad_size, kernel_size,
            max_displacement, stride1, stride2, ctypes.c_float(float(negative_slope)),
            buffer.stride(0), code, _stream_ptr(x1))
    _lib.check(rc, what)


def _correlation_meta(input1, input2, pad_size, kernel_size, max_displacement, stride1,
                      stride2, corr_type_multiply, *_):
    B, _, H, W = input1.shape
    oc, oh, ow = _meta_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    return input1.new_empty((B, oc, oh, ow))


# ----------------------------------------------------------------------------
# correlation backward
# ----------------------------------------------------------------------------
def _correlation_backward_cuda(input1, input2, gradOutput, pad_size, kernel_size,
                               max_displacement, stride1, stride2,
                               corr_type_multiply) -> List[torch.Tensor]:
    what = "cerberus::correlation_backward"
    _check_pair(input1, input2, what)
    code = _dtype_code(input1, what)
    if stride1 != 1:
        raise RuntimeError(what + ": stride1 must be 1 (the reference kernel writes out of "
                           "bounds for stride1 > 1, correlation_cuda_kernel.cu:106-107,169)")
    x1 = input1.contiguous()
    x2 = input2.contiguous()
    B, C, H, W = x1.shape
    oc, oh, ow = _corr_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    if tuple(gradOutput.shape) != (B, oc, oh, ow):
        raise RuntimeError("%s: gradOutput shape %s, expected %s"
                           % (what, tuple(gradOutput.shape), (B, oc, oh, ow)))
    go = gradOutput.to(dtype=x1.dtype).contiguous()
    g1 = torch.empty_like(x1)
    g2 = torch.empty_like(x2)
    if x1.numel() == 0:
        return [g1, g2]
    with torch.cuda.device(x1.device):
        rc = _lib.get().cerberus_correlation_backward(
            x1.data_ptr(), x2.data_ptr(), go.data_ptr(), g1.data_ptr(), g2.data_ptr(), B, C, H,
            W, pad_size, kernel_size, max_displacement, stride1, stride2, corr_type_multiply,
            code, _stream_ptr(x1))
    _lib.check(rc, what)
    return [g1, g2]


def _correlation_backward_meta(input1, input2, gradOutput, *_):
    return [torch.empty_like(input1), torch.empty_like(input2)]


def _correlation_backward_leaky_cuda(input1, input2, grad_buffer, fwd_buffer, channel_offset, pad_size,
                                     kernel_size, max_displacement, stride1, stride2, corr_type_multiply,
                                     negative_slope) -> List[torch.Tensor]:
    """Backward of ``correlation_leaky_into``: ``grad_buffer`` is the gradient of the whole concatenation
    buffer, ``fwd_buffer`` the buffer the forward wrote; the cost volume is channels [channel_offset,
    channel_offset + oC) of both.  The LeakyReLU derivative (from the stored volume's sign) and the
    gather of the batch-strided slice happen in ONE pass of the library (cerberus_correlation_backward_ex)
    instead of ``g * slope``, ``torch.where`` and ``.contiguous()`` (pwcnet_sfd.py:181-187 seen from autograd)."""
    what = "cerberus::correlation_backward_leaky"
    _check_pair(input1, input2, what)
    code = _dtype_code(input1, what)
    if stride1 != 1:
        raise RuntimeError(what + ": stride1 must be 1 (the reference kernel writes out of "
                           "bounds for stride1 > 1, correlation_cuda_kernel.cu:106-107,169)")
    x1, x2 = input1.contiguous(), input2.contiguous()
    B, C, H, W = x1.shape
    oc, oh, ow = _corr_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    for name, t in (("grad_buffer", grad_buffer), ("fwd_buffer", fwd_buffer)):
        if (t.dim() != 4 or t.shape[0] != B or tuple(t.shape[2:]) != (oh, ow) or t.device != x1.device or
                channel_offset < 0 or channel_offset + oc > t.shape[1]):
            raise RuntimeError("%s: %s %s cannot hold channels [%d, %d) of a (%d, %d, %d, %d) cost volume"
                               % (what, name, tuple(t.shape), channel_offset, channel_offset + oc, B, oc, oh, ow))
    # plane-dense AND one item per batch stride: an expanded gradient (``out.sum(0)`` hands over ``grad.expand(...)``
    # with batch stride 0) has dense planes but B items in ONE item's storage -- the library would read past it
    # (ADVICE r5), so it is materialised like any other non-dense layout
    def usable(t):
        return (t.stride(3) == 1 and t.stride(2) == ow and t.stride(1) == oh * ow and
                (B == 1 or t.stride(0) >= t.shape[1] * oh * ow))
    go = grad_buffer.to(dtype=x1.dtype)
    if not usable(go):
        go = go.contiguous()
    fo = fwd_buffer if (fwd_buffer.dtype == x1.dtype and usable(fwd_buffer)) else fwd_buffer.to(x1.dtype).contiguous()
    # the real batch stride always (the C ABI reads 0 as "dense": never hand it a Python-side 0)
    go_bs = go.stride(0) if B > 1 else go.shape[1] * oh * ow
    fo_bs = fo.stride(0) if B > 1 else fo.shape[1] * oh * ow
    g1, g2 = torch.empty_like(x1), torch.empty_like(x2)
    if x1.numel() == 0:
        return [g1, g2]
    lib = _lib.get()
    ws_bytes = lib.cerberus_correlation_backward_ex_workspace_bytes(B, H, W, pad_size, kernel_size, max_displacement,
                                                                    stride1, stride2, code)
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.int64, device=x1.device)   # caching allocator: 512-byte aligned
    with torch.cuda.device(x1.device):
        rc = lib.cerberus_correlation_backward_ex(
            x1.data_ptr(), x2.data_ptr(), go[:, channel_offset].data_ptr(), go_bs,
            fo[:, channel_offset].data_ptr(), fo_bs, ctypes.c_float(float(negative_slope)),
            ws.data_ptr(), ws_bytes, g1.data_ptr(), g2.data_ptr(), B, C, H, W, pad_size, kernel_size,
            max_displacement, stride1, stride2, code, _stream_ptr(x1))
    _lib.check(rc, what)
    return [g1, g2]


# ----------------------------------------------------------------------------
# flow warp
# ----------------------------------------------------------------------------
def _warp_check(image, flow, what):
    if image.dim() != 4 or flow.dim() != 4:
        raise RuntimeError("%s: expected image (B,C,H,W) and flow (B,2,H,W)" % what)
    B, _, H, W = image.shape
    if tuple(flow.shape) != (B, 2, H, W):
        raise RuntimeError("%s: flow shape %s, expected %s"
                           % (what, tuple(flow.shape), (B, 2, H, W)))
    if image.device != flow.device:
        raise RuntimeError("%s: image and flow on different devices" % what)


def _flow_for(img, flow):
    """The flow as the kernels take it: in the image's dtype, except that an fp32 flow stays
    fp32 beside a 16-bit image.  (Reference, UnFlowLoss.py:89-93: ``base_grid.type_as(image) +
    flow12`` promotes to fp32 and grid_sample is on autocast's fp32 list, so a half image with
    an fp32 flow is sampled at full flow precision -- rounding the flow to fp16/bf16 would
    cost 0.03-0.25 px at |flow| of 32-512 px.)  The output keeps the image's dtype: the
    reference's caller casts it back at once (pwcnet_sfd.py:178 ``.type(im1.dtype)``)."""
    if flow.dtype == torch.float32 and img.dtype in (torch.float16, torch.bfloat16):
        return flow.contiguous()
    return flow.to(dtype=img.dtype).contiguous()


def _check_aligned16(t, what, name):
    """The warp kernels read the context / workspace header as int4 (ABI v4: CERB_EINVAL otherwise).
    torch's caching allocator hands out 512-byte aligned blocks; a sliced or re-viewed tensor may not be."""
    if t is not None and t.numel() and t.data_ptr() % 16:
        raise RuntimeError("%s: the warp %s must be 16-byte aligned (got an address ending in 0x%x): "
                           "pass the tensor the forward returned, not a slice of it" % (what, name, t.data_ptr() % 16))


def _flow_warp_run(image, flow, pad_mode, interp_mode, want_ctx, what):
    _warp_check(image, flow, what)
    code = _dtype_code(image, what)
    img = image.contiguous()
    flo = _flow_for(img, flow)
    out = torch.empty_like(img)
    B, C, H, W = img.shape
    lib = _lib.get()
    ctx_bytes = lib.cerberus_flow_warp_context_bytes(B, H, W) if want_ctx else 0
    # int64 elements; the kernels need 16-byte alignment (checked below: the caching allocator gives 512)
    ctx = torch.empty((ctx_bytes + 7) // 8, dtype=torch.int64, device=img.device) if want_ctx else None
    if want_ctx:
        _check_aligned16(ctx, what, "context")
    if out.numel() == 0:
        return out, ctx
    with torch.cuda.device(img.device):
        rc = lib.cerberus_flow_warp_forward_ctx(
            img.data_ptr(), flo.data_ptr(), out.data_ptr(),
            ctx.data_ptr() if want_ctx else None, ctx_bytes, B, C, H, W, pad_mode, interp_mode,
            code, _DTYPES[flo.dtype], _stream_ptr(img))
    _lib.check(rc, what)
    return out, ctx


def _flow_warp_cuda(image, flow, pad_mode, interp_mode):
    return _flow_warp_run(image, flow, pad_mode, interp_mode, False, "cerberus::flow_warp")[0]


def _flow_warp_ctx_cuda(image, flow, pad_mode, interp_mode):
    """Forward that also returns the backward context (sample positions + tap extents):
    the analogue of what autograd saves for grid_sample in the reference."""
    return _flow_warp_run(image, flow, pad_mode, interp_mode, True, "cerberus::flow_warp_ctx")


def _warp_context_bytes(B, H, W):
    """Size of the warp backward's context in pure Python (what cerberus_flow_warp_context_bytes returns; a test holds
    the two equal): one int4 tap range per 2 x 32 pixel strip + two fp32 sample-position planes per image.  Meta /
    fake-tensor tracing must not need the shared library."""
    if B <= 0 or H <= 0 or W <= 0:
        return 0
    strips = ((W + 31) // 32) * ((H + 1) // 2)
    return B * strips * 16 + B * 2 * H * W * 4


def _flow_warp_ctx_meta(image, flow, pad_mode, interp_mode):
    B, _, H, W = image.shape
    n = (_warp_context_bytes(B, H, W) + 7) // 8
    return torch.empty_like(image), image.new_empty((n,), dtype=torch.int64)


def _flow_warp_backward_run(image, flow, context, grad_out, pad_mode, interp_mode, need_image,
                            need_flow, what) -> List[torch.Tensor]:
    _warp_check(image, flow, what)
    code = _dtype_code(image, what)
    img = image.contiguous()
    flo = _flow_for(img, flow)
    go = grad_out.to(dtype=img.dtype).contiguous()
    if go.shape != img.shape:
        raise RuntimeError("%s: grad_out shape %s, expected %s"
                           % (what, tuple(go.shape), tuple(img.shape)))
    gi = torch.empty_like(img) if need_image else img.new_empty(0)
    gf = torch.empty_like(flo) if need_flow else flo.new_empty(0)
    if img.numel() == 0 or not (need_image or need_flow):
        return [gi, gf]
    B, C, H, W = img.shape
    lib = _lib.get()
    ctx_ptr, ctx_bytes = None, 0
    if context is not None:
        ctx_bytes = lib.cerberus_flow_warp_context_bytes(B, H, W)
        if (context.device != img.device or not context.is_contiguous()
                or context.numel() * context.element_size() < ctx_bytes):
            raise RuntimeError("%s: context does not belong to this image/flow shape" % what)
        _check_aligned16(context, what, "context")
        ctx_ptr = context.data_ptr()
    # device scratch in which the tiled grad_image builds its context when the forward saved
    # none (caching allocator: no sync, graph-capturable; stream-ordered reuse keeps it
    # private to this call)
    ws, ws_bytes = None, 0
    if (need_image or (need_flow and C > 4)) and context is None:   # (round 6: grad_flow alone takes the tile launch's flow role too)
        ws_bytes = lib.cerberus_flow_warp_backward_workspace_bytes(B, C, H, W)
        ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.int64, device=img.device)
        _check_aligned16(ws, what, "workspace")
    with torch.cuda.device(img.device):
        rc = lib.cerberus_flow_warp_backward(
            img.data_ptr(), flo.data_ptr(), go.data_ptr(),
            gi.data_ptr() if need_image else None, gf.data_ptr() if need_flow else None,
            ctx_ptr, ctx_bytes,
            ws.data_ptr() if ws is not None else None, ws_bytes,
            B, C, H, W, pad_mode, interp_mode, code, _DTYPES[flo.dtype], _stream_ptr(img))
    _lib.check(rc, what)
    return [gi, gf]


def _flow_warp_backward_cuda(image, flow, grad_out, pad_mode, interp_mode, need_image,
                             need_flow) -> List[torch.Tensor]:
    return _flow_warp_backward_run(image, flow, None, grad_out, pad_mode, interp_mode,
                                   need_image, need_flow, "cerberus::flow_warp_backward")


def _flow_warp_backward_ctx_cuda(image, flow, context, grad_out, pad_mode, interp_mode,
                                 need_image, need_flow) -> List[torch.Tensor]:
    return _flow_warp_backward_run(image, flow, context, grad_out, pad_mode, interp_mode,
                                   need_image, need_flow, "cerberus::flow_warp_backward_ctx")


# ----------------------------------------------------------------------------
# f2: the warp fused into the correlation forward (pwcnet_sfd.py:178 -> :181-182, d = 4)
# ----------------------------------------------------------------------------
def _warp_correlation_leaky_cuda(input1, input2, flow, pad_mode, negative_slope):
    """LeakyReLU(correlation(input1, flow_warp(input2, flow))) with pad = d = 4, k = s1 = s2 = 1, without writing the
    warped features (cerberus_warp_correlation_forward).  Opt-in: measured slower than the two tuned launches."""
    what = "cerberus::warp_correlation_leaky"
    _check_pair(input1, input2, what)
    _warp_check(input2, flow, what)
    code = _dtype_code(input1, what)
    x1, x2 = input1.contiguous(), input2.contiguous()
    flo = _flow_for(x2, flow)
    B, C, H, W = x1.shape
    out = x1.new_empty((B, 81, H, W))
    if out.numel() == 0:
        return out
    lib = _lib.get()
    ws_bytes = lib.cerberus_warp_correlation_workspace_bytes(B, C, H, W)    # small maps: fp32 volume for the split channel sum
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.int64, device=x1.device) if ws_bytes else None
    with torch.cuda.device(x1.device):
        rc = lib.cerberus_warp_correlation_forward(
            x1.data_ptr(), x2.data_ptr(), flo.data_ptr(), out.data_ptr(), ws.data_ptr() if ws is not None else None, ws_bytes,
            B, C, H, W, pad_mode, ctypes.c_float(float(negative_slope)), 0, code, _DTYPES[flo.dtype], _stream_ptr(x1))
    _lib.check(rc, what)
    return out


# ----------------------------------------------------------------------------
# flow upsample (pwcnet_sfd.py:176, :199-201)
# ----------------------------------------------------------------------------
def _flow_upsample_run(t, factor, forward, what):
    if t.dim() != 4:
        raise RuntimeError("%s: expected a 4-D NCHW tensor, got %s" % (what, tuple(t.shape)))
    if factor < 1:
        raise RuntimeError("%s: factor must be >= 1" % what)
    code = _dtype_code(t, what)
    if code == 3:
        raise RuntimeError("%s: float64 is not supported" % what)
    x = t.contiguous()
    B, C, H, W = x.shape
    if forward:
        h, w = H, W
        out = x.new_empty((B, C, H * factor, W * factor))
    else:
        if H % factor or W % factor:
            raise RuntimeError("%s: grad shape %s is not a multiple of the factor %d"
                               % (what, tuple(x.shape), factor))
        h, w = H // factor, W // factor
        out = x.new_empty((B, C, h, w))
    if out.numel() == 0 or x.numel() == 0:
        return out
    fn = _lib.get().cerberus_flow_upsample_forward if forward else _lib.get().cerberus_flow_upsample_backward
    with torch.cuda.device(x.device):
        rc = fn(x.data_ptr(), out.data_ptr(), B * C, h, w, factor, code, _stream_ptr(x))
    _lib.check(rc, what)
    return out


def _flow_upsample_cuda(flow, factor):
    return _flow_upsample_run(flow, factor, True, "cerberus::flow_upsample")


def _flow_upsample_backward_cuda(grad_out, factor):
    return _flow_upsample_run(grad_out, factor, False, "cerberus::flow_upsample_backward")


def _area_resize_cuda(image, out_h, out_w):
    """F.interpolate(image, (out_h, out_w), mode='area') (UnFlowLoss.py:279-280)."""
    what = "cerberus::area_resize"
    if image.dim() != 4:
        raise RuntimeError("%s: expected a 4-D NCHW tensor, got %s" % (what, tuple(image.shape)))
    if out_h < 1 or out_w < 1:
        raise RuntimeError("%s: output size must be positive, got (%d, %d)" % (what, out_h, out_w))
    code = _dtype_code(image, what)
    if code == 3:
        raise RuntimeError("%s: float64 is not supported" % what)
    x = image.contiguous()
    B, C, H, W = x.shape
    out = x.new_empty((B, C, out_h, out_w))
    if out.numel() == 0:
        return out
    if H == 0 or W == 0:
        raise RuntimeError("%s: input has no pixels" % what)
    with torch.cuda.device(x.device):
        rc = _lib.get().cerberus_area_resize(x.data_ptr(), out.data_ptr(), B * C, H, W, out_h, out_w,
                                             code, _stream_ptr(x))
    _lib.check(rc, what)
    return out


def _area_pyramid_cuda(image, sizes) -> List[torch.Tensor]:
    """``[F.interpolate(image, (h, w), mode='area') for (h, w) in sizes]`` (``sizes`` flat: h0, w0, h1, w1, ...):
    how unFlowLoss brings a target image to every flow scale (UnFlowLoss.py:279-280).  A scale of the image's
    own size is a copy of the image; the others come from one pass over the source when they
    are integer ratios of it (cerberus_area_pyramid), bit-identical to ``area_resize`` scale by scale."""
    what = "cerberus::area_pyramid"
    if image.dim() != 4:
        raise RuntimeError("%s: expected a 4-D NCHW tensor, got %s" % (what, tuple(image.shape)))
    if len(sizes) % 2:
        raise RuntimeError("%s: sizes must be a flat list of (height, width) pairs" % what)
    code = _dtype_code(image, what)
    if code == 3:
        raise RuntimeError("%s: float64 is not supported" % what)
    x = image.contiguous()
    B, C, H, W = x.shape
    pairs = [(int(sizes[i]), int(sizes[i + 1])) for i in range(0, len(sizes), 2)]
    if any(h < 1 or w < 1 for h, w in pairs):
        raise RuntimeError("%s: output sizes must be positive, got %s" % (what, pairs))
    # a scale of the image's own size is a COPY (F.interpolate returns one; a custom op must not return an alias of
    # its input: ADVICE r5) -- unFlowLoss asks for it at most once per step and only on pyramids that start at 1/1
    outs = [x.clone() if (h, w) == (H, W) else x.new_empty((B, C, h, w)) for h, w in pairs]
    todo = [(o, hw) for o, hw in zip(outs, pairs) if hw != (H, W) and o.numel()]
    if not todo:
        return outs
    if H == 0 or W == 0:
        raise RuntimeError("%s: input has no pixels" % what)
    lib = _lib.get()
    with torch.cuda.device(x.device):
        for i in range(0, len(todo), 4):        # one launch holds up to four scales
            chunk = todo[i:i + 4]
            n = len(chunk)
            dsts = (ctypes.c_void_p * n)(*[o.data_ptr() for o, _ in chunk])
            hs = (ctypes.c_int * n)(*[hw[0] for _, hw in chunk])
            ws = (ctypes.c_int * n)(*[hw[1] for _, hw in chunk])
            _lib.check(lib.cerberus_area_pyramid(x.data_ptr(), dsts, hs, ws, n, B * C, H, W, code, _stream_ptr(x)), what)
    return outs


def _area_pyramid_meta(image, sizes):
    B, C, _, _ = image.shape
    return [image.new_empty((B, C, int(sizes[i]), int(sizes[i + 1]))) for i in range(0, len(sizes), 2)]


def _no_cpu(name):
    def _raise(*_a, **_k):
        raise RuntimeError("cerberus::%s has no CPU implementation: this build is the "
                           "MI355X HIP path only (move the tensors to the GPU)" % name)
    return _raise


_def.impl("correlation", _correlation_cuda, "CUDA")
_def.impl("correlation", _correlation_meta, "Meta")
_def.impl("correlation", _no_cpu("correlation"), "CPU")
_def.impl("correlation_leaky", _correlation_leaky_cuda, "CUDA")
_def.impl("correlation_leaky", _correlation_meta, "Meta")
_def.impl("correlation_leaky", _no_cpu("correlation_leaky"), "CPU")
_def.impl("warp_correlation_leaky", _warp_correlation_leaky_cuda, "CUDA")
_def.impl("warp_correlation_leaky", lambda a, b, f, p, sl: a.new_empty((a.shape[0], 81, a.shape[2], a.shape[3])), "Meta")
_def.impl("warp_correlation_leaky", _no_cpu("warp_correlation_leaky"), "CPU")
_def.impl("correlation_leaky_into", _correlation_leaky_into_cuda, "CUDA")
_def.impl("correlation_leaky_into", lambda *a: None, "Meta")
_def.impl("correlation_leaky_into", _no_cpu("correlation_leaky_into"), "CPU")
_def.impl("correlation_backward", _correlation_backward_cuda, "CUDA")
_def.impl("correlation_backward", _correlation_backward_meta, "Meta")
_def.impl("correlation_backward", _no_cpu("correlation_backward"), "CPU")
_def.impl("correlation_backward_leaky", _correlation_backward_leaky_cuda, "CUDA")
_def.impl("correlation_backward_leaky", _correlation_backward_meta, "Meta")
_def.impl("correlation_backward_leaky", _no_cpu("correlation_backward_leaky"), "CPU")
_def.impl("area_pyramid", _area_pyramid_cuda, "CUDA")
_def.impl("area_pyramid", _area_pyramid_meta, "Meta")
_def.impl("area_pyramid", _no_cpu("area_pyramid"), "CPU")
_def.impl("flow_upsample", _flow_upsample_cuda, "CUDA")
_def.impl("flow_upsample", lambda f, k: f.new_empty((f.shape[0], f.shape[1], f.shape[2] * k, f.shape[3] * k)), "Meta")
_def.impl("flow_upsample", _no_cpu("flow_upsample"), "CPU")
_def.impl("flow_upsample_backward", _flow_upsample_backward_cuda, "CUDA")
_def.impl("flow_upsample_backward",
          lambda g, k: g.new_empty((g.shape[0], g.shape[1], g.shape[2] // k, g.shape[3] // k)), "Meta")
_def.impl("flow_upsample_backward", _no_cpu("flow_upsample_backward"), "CPU")
_def.impl("area_resize", _area_resize_cuda, "CUDA")
_def.impl("area_resize", lambda x, h, w: x.new_empty((x.shape[0], x.shape[1], h, w)), "Meta")
_def.impl("area_resize", _no_cpu("area_resize"), "CPU")
_def.impl("flow_warp", _flow_warp_cuda, "CUDA")
_def.impl("flow_warp", lambda image, flow, p, m: torch.empty_like(image), "Meta")
_def.impl("flow_warp", _no_cpu("flow_warp"), "CPU")
_def.impl("flow_warp_backward", _flow_warp_backward_cuda, "CUDA")
_def.impl("flow_warp_backward",
          lambda image, flow, go, p, m, ni, nf: [torch.empty_like(image), torch.empty_like(flow)],
          "Meta")
_def.impl("flow_warp_backward", _no_cpu("flow_warp_backward"), "CPU")
_def.impl("flow_warp_ctx", _flow_warp_ctx_cuda, "CUDA")
_def.impl("flow_warp_ctx", _flow_warp_ctx_meta, "Meta")
_def.impl("flow_warp_ctx", _no_cpu("flow_warp_ctx"), "CPU")
_def.impl("flow_warp_backward_ctx", _flow_warp_backward_ctx_cuda, "CUDA")
_def.impl("flow_warp_backward_ctx",
          lambda image, flow, ctx, go, p, m, ni, nf: [torch.empty_like(image),
                                                      torch.empty_like(flow)], "Meta")
_def.impl("flow_warp_backward_ctx", _no_cpu("flow_warp_backward_ctx"), "CPU")


# ----------------------------------------------------------------------------
# autograd formulas for the raw ops.  The reference registers none (its raw op
# is only differentiable through CorrelationFunction); having them makes the
# eval-mode path of Correlation.forward (correlation.py:78-80) differentiable
# too and costs nothing.
# ----------------------------------------------------------------------------
def _corr_setup(ctx, inputs, output):
    input1, input2, *params = inputs
    ctx.save_for_backward(input1, input2)
    ctx.params = tuple(params[:6])


def _corr_backward(ctx, grad):
    input1, input2 = ctx.saved_tensors
    g1, g2 = torch.ops.cerberus.correlation_backward(input1, input2, grad, *ctx.params)
    return (g1, g2) + (None,) * 6


def _corr_leaky_setup(ctx, inputs, output):
    input1, input2, *params = inputs
    ctx.save_for_backward(input1, input2, output)
    ctx.params = tuple(params[:6])
    ctx.slope = params[6]


def _corr_leaky_backward(ctx, grad):
    input1, input2, output = ctx.saved_tensors
    # d leaky(v)/dv from the sign of the output (slope > 0 keeps the sign), applied by the library in one pass
    g1, g2 = torch.ops.cerberus.correlation_backward_leaky(input1, input2, grad, output, 0, *ctx.params, ctx.slope)
    return (g1, g2) + (None,) * 7


def _warp_setup(ctx, inputs, output):
    image, flow, pad_mode, interp_mode = inputs
    ctx.save_for_backward(image, flow)
    ctx.modes = (pad_mode, interp_mode)


def _warp_backward(ctx, grad):
    image, flow = ctx.saved_tensors
    need_image, need_flow = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    gi, gf = torch.ops.cerberus.flow_warp_backward(image, flow, grad, ctx.modes[0],
                                                   ctx.modes[1], need_image, need_flow)
    return (gi if need_image else None, gf.to(flow.dtype) if need_flow else None, None, None)


def _warp_corr_setup(ctx, inputs, output):
    input1, input2, flow, pad_mode, slope = inputs
    # f2's point in training: neither the warped features nor the warp's context are kept -- only the three inputs and the
    # (81-channel) result, whose sign carries the LeakyReLU derivative
    ctx.save_for_backward(input1, input2, flow, output)
    ctx.pad_mode, ctx.slope = pad_mode, slope


def _warp_corr_backward(ctx, grad):
    input1, input2, flow, output = ctx.saved_tensors
    bil = INTERP_MODES["bilinear"]
    # recompute the warp (the tuned kernels, with its context), then the tuned correlation and warp backward
    warped, context = torch.ops.cerberus.flow_warp_ctx(input2, flow, ctx.pad_mode, bil)
    g1, gw = torch.ops.cerberus.correlation_backward_leaky(input1, warped, grad, output, 0, 4, 1, 4, 1, 1, 1, ctx.slope)
    need_image, need_flow = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
    g2 = gf = None
    if need_image or need_flow:
        gi, gfl = torch.ops.cerberus.flow_warp_backward_ctx(input2, flow, context, gw, ctx.pad_mode, bil, need_image, need_flow)
        g2 = gi if need_image else None
        gf = gfl.to(flow.dtype) if need_flow else None
    return g1, g2, gf, None, None


torch.library.register_autograd("cerberus::warp_correlation_leaky", _warp_corr_backward, setup_context=_warp_corr_setup)
torch.library.register_autograd("cerberus::correlation", _corr_backward,
                                setup_context=_corr_setup)
torch.library.register_autograd("cerberus::correlation_leaky", _corr_leaky_backward,
                                setup_context=_corr_leaky_setup)
def _warp_ctx_setup(ctx, inputs, output):
    image, flow, pad_mode, interp_mode = inputs
    ctx.save_for_backward(image, flow, output[1])
    ctx.modes = (pad_mode, interp_mode)
    ctx.mark_non_differentiable(output[1])


def _warp_ctx_backward(ctx, grad, _grad_context):
    image, flow, context = ctx.saved_tensors
    need_image, need_flow = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    if ctx.modes[1] != INTERP_MODES["bilinear"] or ctx.modes[0] == PAD_MODES["reflection"]:
        context = None   # unsupported combinations raise in the raw op, as before
        gi, gf = torch.ops.cerberus.flow_warp_backward(image, flow, grad, ctx.modes[0],
                                                       ctx.modes[1], need_image, need_flow)
    else:
        gi, gf = torch.ops.cerberus.flow_warp_backward_ctx(image, flow, context, grad,
                                                           ctx.modes[0], ctx.modes[1],
                                                           need_image, need_flow)
    return (gi if need_image else None, gf.to(flow.dtype) if need_flow else None, None, None)


def _upsample_setup(ctx, inputs, output):
    ctx.factor = inputs[1]


def _upsample_backward(ctx, grad):
    return torch.ops.cerberus.flow_upsample_backward(grad, ctx.factor), None


def _upsample_bwd_setup(ctx, inputs, output):
    ctx.factor = inputs[1]


def _upsample_bwd_backward(ctx, grad):
    # the adjoint of the adjoint is the forward (the op is linear)
    return torch.ops.cerberus.flow_upsample(grad, ctx.factor), None


def _area_resize_backward(ctx, grad):
    raise RuntimeError("cerberus::area_resize is not differentiable: the reference resizes the "
                       "TARGET images with it (UnFlowLoss.py:279-280), which carry no gradient")


torch.library.register_autograd("cerberus::area_resize", _area_resize_backward,
                                setup_context=lambda ctx, inputs, output: None)
torch.library.register_autograd("cerberus::area_pyramid", _area_resize_backward,
                                setup_context=lambda ctx, inputs, output: None)
torch.library.register_autograd("cerberus::flow_upsample", _upsample_backward,
                                setup_context=_upsample_setup)
torch.library.register_autograd("cerberus::flow_upsample_backward", _upsample_bwd_backward,
                                setup_context=_upsample_bwd_setup)
torch.library.register_autograd("cerberus::flow_warp", _warp_backward,
                                setup_context=_warp_setup)
torch.library.register_autograd("cerberus::flow_warp_ctx", _warp_ctx_backward,
                                setup_context=_warp_ctx_setup)
