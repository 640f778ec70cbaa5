"""Building blocks of the PWC flow head, counterparts of
``nnet_training/nnet_models/pwcnet_modules.py`` (reference :7-102).

Only stock ``torch.nn`` layers (Conv2d + LeakyReLU run on MIOpen); what matters
here is that parameter names and shapes are identical to the reference's, so its
checkpoints load unchanged: every conv block is ``Sequential(Conv2d[, LeakyReLU])``
(keys ``<block>.0.weight`` / ``<block>.0.bias``).
"""
import torch
from torch import nn


def conv_block(c_in, c_out, kernel_size=3, stride=1, dilation=1, activation=True):
    """'same'-padded convolution, optionally followed by LeakyReLU(0.1) (reference pwc_conv :7-20)."""
    layers = [nn.Conv2d(c_in, c_out, kernel_size, stride=stride, dilation=dilation,
                        padding=(kernel_size - 1) * dilation // 2, bias=True)]
    if activation:
        layers.append(nn.LeakyReLU(0.1, inplace=True))
    return nn.Sequential(*layers)


class FlowEstimatorDense(nn.Module):
    """DenseNet-style estimator: every block sees all earlier feature maps (:44-61)."""
    widths = (128, 128, 96, 64, 32)

    def __init__(self, ch_in):
        super().__init__()
        seen = ch_in
        for i, w in enumerate(self.widths, start=1):
            setattr(self, "conv%d" % i, conv_block(seen, w))
            seen += w
        self.feat_dim = seen
        self.conv_last = conv_block(seen, 2, activation=False)

    def forward(self, x):
        for i in range(1, len(self.widths) + 1):
            x = torch.cat([getattr(self, "conv%d" % i)(x), x], dim=1)
        return x, self.conv_last(x)


class FlowEstimatorLite(nn.Module):
    """Light estimator: each block sees the two previous outputs (:64-82)."""

    def __init__(self, ch_in):
        super().__init__()
        self.conv1 = conv_block(ch_in, 128)
        self.conv2 = conv_block(128, 128)
        self.conv3 = conv_block(128 + 128, 96)
        self.conv4 = conv_block(128 + 96, 64)
        self.conv5 = conv_block(96 + 64, 32)
        self.feat_dim = 32
        self.predict_flow = conv_block(64 + 32, 2, activation=False)

    def forward(self, x):
        a = self.conv1(x)
        b = self.conv2(a)
        c = self.conv3(torch.cat([a, b], dim=1))
        d = self.conv4(torch.cat([b, c], dim=1))
        e = self.conv5(torch.cat([c, d], dim=1))
        return e, self.predict_flow(torch.cat([d, e], dim=1))


class ContextNetwork(nn.Module):
    """Dilated refinement stack, dilations 1-2-4-8-16-1-1 (:85-102)."""
    plan = ((128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1))

    def __init__(self, ch_in):
        super().__init__()
        blocks, c = [], ch_in
        for width, dil in self.plan:
            blocks.append(conv_block(c, width, 3, 1, dil))
            c = width
        blocks.append(conv_block(c, 2, activation=False))
        self.convs = nn.Sequential(*blocks)

    def forward(self, x):
        return self.convs(x)
