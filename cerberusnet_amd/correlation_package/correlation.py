"""Drop-in for ``nnet_training/correlation_package/correlation.py`` (public surface only).

Three names, with the reference's constructor / call signatures and defaults:

  ``CorrelationTorch(max_displacement=4)``                                  reference :4-21
  ``CorrelationFunction.apply(input1, input2, pad_size=3, kernel_size=3,
        max_displacement=20, stride1=1, stride2=2, corr_multiply=1)``       reference :23-57
  ``Correlation(pad_size=0, kernel_size=0, max_displacement=0, stride1=1,
        stride2=2, corr_multiply=1)``                                       reference :60-80

Where the reference loads a CUDA extension by a CWD-relative, CPython-3.8-specific path
at import time (:2), importing this module registers ``torch.ops.cerberus.correlation``
and ``torch.ops.cerberus.correlation_backward`` through ``cerberusnet_amd.ops`` (HIP
kernels behind a C ABI, located relative to the package).
"""
import torch
import torch.nn.functional as F

from .. import ops as _ops  # noqa: F401  -- side effect: torch.ops.cerberus.* exist

_HYPER = ("pad_size", "kernel_size", "max_displacement", "stride1", "stride2", "corr_multiply")


class CorrelationTorch(torch.nn.Module):
    """Cost volume in stock PyTorch ops (any device): the semantic anchor of the HIP op at
    pad = d, kernel 1, strides 1.  Output channel ``i * (2d+1) + j`` is the channel mean of
    ``x1 * x2`` shifted by ``i - d`` rows and ``j - d`` columns (zero outside)."""

    def __init__(self, max_displacement=4, *args, **kwargs):
        super().__init__()
        self.max_displacement = max_displacement
        self.pad_size = max_displacement
        self.output_dim = 2 * max_displacement + 1

    def forward(self, x1, x2):
        rows, cols = x1.shape[-2:]
        span, border = self.output_dim, self.pad_size
        x2_padded = F.pad(x2, (border, border, border, border))
        out = x1.new_empty((x1.shape[0], span * span, rows, cols))
        for k in range(span * span):
            i, j = divmod(k, span)  # i: vertical (slow), j: horizontal (fast)
            out[:, k] = (x1 * x2_padded[..., i:i + rows, j:j + cols]).mean(dim=1)
        return out


class CorrelationFunction(torch.autograd.Function):
    """autograd glue around the two raw ops.  Only the inputs are saved: callers apply an
    in-place ``leaky_relu`` to the output (pwcnet_sfd.py:182), so it must not be needed
    by backward.  Runs in the caller's dtype under autocast (no input cast)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3, max_displacement=20,
                stride1=1, stride2=2, corr_multiply=1):
        ctx.hyper = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
        ctx.save_for_backward(input1, input2)
        return torch.ops.cerberus.correlation(input1, input2, *ctx.hyper)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        grads = torch.ops.cerberus.correlation_backward(*ctx.saved_tensors, grad_output,
                                                        *ctx.hyper)
        return (grads[0], grads[1]) + (None,) * len(_HYPER)


class Correlation(torch.nn.Module):
    """Stateless module (nothing in ``state_dict``): training mode goes through
    ``CorrelationFunction`` (differentiable), eval mode calls the raw op."""

    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0,
                 stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        for name, value in zip(_HYPER, (pad_size, kernel_size, max_displacement, stride1,
                                        stride2, corr_multiply)):
            setattr(self, name, value)

    def forward(self, input1, input2):
        hyper = tuple(getattr(self, name) for name in _HYPER)
        run = CorrelationFunction.apply if self.training else torch.ops.cerberus.correlation
        return run(input1, input2, *hyper)


class CostVolumeConcat(torch.autograd.Function):
    """``torch.cat([leaky_relu(corr(x1, x2), slope), *others], dim=1)`` with the cost volume
    written by the correlation kernel straight into the concatenation buffer (the reference
    builds it in three passes: the op, an in-place ``leaky_relu``, ``torch.cat``:
    pwcnet_sfd.py:181-187).  Not on the reference's surface: an opt-in beside the drop-in one
    (SURVEY.md section 8(f)-1).

    The other tensors are copied into their channel slices (what ``cat`` does for them too);
    backward hands their gradients out as views of the incoming gradient; the LeakyReLU
    derivative (from the sign of the stored volume) and the gather of the volume's batch-strided
    gradient slice are one pass inside ``cerberus::correlation_backward_leaky``, followed by the
    one-launch correlation backward.  The buffer is saved for backward like any output: writing to it in
    place before backward raises autograd's version error."""

    @staticmethod
    def forward(ctx, input1, input2, slope, hyper, *others):
        oc, oh, ow = _ops._corr_out_shape(input1.shape[2], input1.shape[3], hyper[0], hyper[1],
                                          hyper[2], hyper[3], hyper[4])
        widths = [t.shape[1] for t in others]
        buf = input1.new_empty((input1.shape[0], oc + sum(widths), oh, ow))
        torch.ops.cerberus.correlation_leaky_into(buf, input1, input2, 0, *hyper, slope)
        at = oc
        for t, wdt in zip(others, widths):
            buf[:, at:at + wdt].copy_(t)
            at += wdt
        # the buffer itself is saved (an output: autograd keeps it without a reference cycle), so an
        # in-place write to it between forward and backward -- which would flip LeakyReLU derivative
        # signs silently -- trips autograd's version check instead
        ctx.save_for_backward(input1, input2, buf)
        ctx.hyper, ctx.slope, ctx.widths, ctx.oc = tuple(hyper), slope, widths, oc
        return buf

    @staticmethod
    def backward(ctx, grad):
        input1, input2, buf = ctx.saved_tensors
        g1 = g2 = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # the cost volume's slice of the incoming gradient is batch-strided and still needs the LeakyReLU
            # derivative: one pass of the library (cerberus_correlation_backward_ex) instead of g * slope,
            # torch.where and .contiguous() -- 127 MB instead of 255 MB of traffic at the 32 x 128 x 256 level
            g1, g2 = torch.ops.cerberus.correlation_backward_leaky(input1, input2, grad, buf, 0, *ctx.hyper, ctx.slope)
        outs, at = [], ctx.oc
        for wdt in ctx.widths:
            outs.append(grad[:, at:at + wdt])
            at += wdt
        return (g1, g2, None, None) + tuple(outs)


def cost_volume_concat(input1, input2, others, hyper, negative_slope=0.1):
    """See ``CostVolumeConcat``; ``hyper`` = (pad_size, kernel_size, max_displacement, stride1,
    stride2, corr_multiply)."""
    return CostVolumeConcat.apply(input1, input2, float(negative_slope), tuple(hyper), *others)
