"""Pure-PyTorch restatement of the reference's ``CorrelationTorch``
(TEST INFRASTRUCTURE ONLY; also the ``cpu_baseline`` "port" timed by bench.py).

Follows /root/reference/nnet_training/correlation_package/correlation.py:4-21:
zero-pad the second map by d on all four sides (:14); for every row offset i
and column offset j in 0..2d take the channel mean of x1 * shifted(x2)
(:16-19); concatenate so that output channel = i*(2d+1)+j (:21, quirk Q1:
vertical displacement is the slow index).  Same op sequence as the reference
(one multiply + one mean per displacement, one final cat) so that timing it on
host cores is a faithful stand-in for the reference's CPU path.
"""
import torch
import torch.nn.functional as F


def correlation_torch_ref(x1: torch.Tensor, x2: torch.Tensor,
                          max_displacement: int = 4) -> torch.Tensor:
    d = int(max_displacement)
    span = 2 * d + 1
    _, _, height, width = x1.shape
    x2p = F.pad(x2, (d, d, d, d))
    planes = []
    for i in range(span):          # vertical offset, slow index
        for j in range(span):      # horizontal offset, fast index
            window = x2p[:, :, i:i + height, j:j + width]
            planes.append((x1 * window).mean(dim=1, keepdim=True))
    return torch.cat(planes, dim=1)


class CorrelationTorchRef(torch.nn.Module):
    """Module form, mirroring ``CorrelationTorch(max_displacement)`` (:5-9)."""

    def __init__(self, max_displacement: int = 4, *_, **__):
        super().__init__()
        self.max_displacement = max_displacement
        self.output_dim = 2 * max_displacement + 1
        self.pad_size = max_displacement

    def forward(self, x1, x2):
        return correlation_torch_ref(x1, x2, self.max_displacement)
