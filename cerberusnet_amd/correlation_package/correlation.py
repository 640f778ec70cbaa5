"""Drop-in for ``nnet_training/correlation_package/correlation.py``.

Same three public classes, same constructor / forward signatures and defaults:

  * ``CorrelationTorch(max_displacement=4)``          reference :4-21
  * ``CorrelationFunction.apply(input1, input2, pad_size=3, kernel_size=3,
        max_displacement=20, stride1=1, stride2=2, corr_multiply=1)``  :23-57
  * ``Correlation(pad_size=0, kernel_size=0, max_displacement=0, stride1=1,
        stride2=2, corr_multiply=1)``                 :60-80

The import-time ``torch.ops.load_library(<cwd-relative cpython-38 path>)`` of
the reference (:2) is replaced by importing ``cerberusnet_amd.ops``, which
registers ``torch.ops.cerberus.correlation{,_backward}`` on top of
libcerberus_hip.so (resolved relative to the package, any CWD, any CPython).
"""
import torch

from .. import ops as _ops  # noqa: F401  (registers torch.ops.cerberus.*)


class CorrelationTorch(torch.nn.Module):
    """Pure-PyTorch cost volume (device-agnostic), as the reference ships it."""

    def __init__(self, max_displacement=4, *args, **kwargs):
        super().__init__()
        self.max_displacement = max_displacement
        self.output_dim = 2 * self.max_displacement + 1
        self.pad_size = self.max_displacement

    def forward(self, x1, x2):
        height, width = x1.shape[2], x1.shape[3]
        padded = torch.nn.functional.pad(x2, [self.pad_size] * 4)
        volume = [
            torch.mean(x1 * padded[:, :, dy:dy + height, dx:dx + width], 1, keepdim=True)
            for dy in range(self.output_dim) for dx in range(self.output_dim)
        ]
        return torch.cat(volume, 1)


class CorrelationFunction(torch.autograd.Function):
    """
    Typical Parameters: pad_size=3, kernel_size=3, max_displacement=20,
    stride1=1, stride2=2, corr_multiply=1
    """

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3,
                max_displacement=20, stride1=1, stride2=2, corr_multiply=1):
        # only the two inputs are saved (the output is NOT: callers apply an
        # in-place leaky_relu to it, pwcnet_sfd.py:182)
        ctx.save_for_backward(input1, input2)
        ctx.pad_size = pad_size
        ctx.kernel_size = kernel_size
        ctx.max_displacement = max_displacement
        ctx.stride1 = stride1
        ctx.stride2 = stride2
        ctx.corr_multiply = corr_multiply
        return torch.ops.cerberus.correlation(
            input1, input2, pad_size, kernel_size,
            max_displacement, stride1, stride2, corr_multiply)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_outputs):
        input1, input2 = ctx.saved_tensors
        grad_input1, grad_input2 = torch.ops.cerberus.correlation_backward(
            input1, input2, grad_outputs, ctx.pad_size, ctx.kernel_size,
            ctx.max_displacement, ctx.stride1, ctx.stride2, ctx.corr_multiply)
        return grad_input1, grad_input2, None, None, None, None, None, None


class Correlation(torch.nn.Module):
    """Parameter-free module; nothing enters ``state_dict``."""

    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0,
                 stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply

    def forward(self, input1, input2):
        if self.training:
            return CorrelationFunction.apply(
                input1, input2, self.pad_size, self.kernel_size,
                self.max_displacement, self.stride1, self.stride2, self.corr_multiply)
        return torch.ops.cerberus.correlation(
            input1, input2, self.pad_size, self.kernel_size,
            self.max_displacement, self.stride1, self.stride2, self.corr_multiply)
