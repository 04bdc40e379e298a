"""ResNet9 (Myrtle) on 128x128 log-mel images — reference models2d.py:13-87, same
``state_dict`` keys (conv1, conv2, res1.{0,1}, conv3, conv4, res2.{0,1}, linear) and the same
``forward(x, depth=None, pass_part=None)`` signature.  This is the one place on the hot path
where the MFMA units matter (6.05 GMAC forward per sample); the convolutions go through MIOpen.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _block(c_in: int, c_out: int, pool: bool = False) -> nn.Sequential:
    layers = [nn.Conv2d(c_in, c_out, kernel_size=3, padding=1), nn.BatchNorm2d(c_out),
              nn.ReLU(inplace=True)]
    if pool:
        layers.append(nn.MaxPool2d(2))
    return nn.Sequential(*layers)


class ResNet9_myrtle(nn.Module):
    def __init__(self, in_channels: int, num_classes: int, linear: int):
        super().__init__()
        self.conv1 = _block(in_channels, 64)
        self.conv2 = _block(64, 128, pool=True)
        self.res1 = nn.Sequential(_block(128, 128), _block(128, 128))
        self.conv3 = _block(128, 256, pool=True)
        self.conv4 = _block(256, 512, pool=True)
        self.res2 = nn.Sequential(_block(512, 512), _block(512, 512))
        self.pool2d = nn.MaxPool2d(4)
        self.flat = nn.Flatten()
        self.linear = nn.Linear(linear, num_classes)
        # Conv weights are kept channels_last and, on a HIP device, so are the activations: MIOpen's
        # fp32 implicit-GEMM kernels are NHWC and otherwise get wrapped in transposes
        # (117 -> 101 ms per bs=256 forward+backward, profiles/probes/resnet2d_layout_probe.py).
        # Values, state_dict keys and shapes are unchanged (a memory format, not a reshape).
        self.to(memory_format=torch.channels_last)

    def _block(self, seq, h, skip=None):
        if not h.is_cuda:
            return seq(h) if skip is None else seq(h) + skip
        from .models import conv_bn_relu_pool               # bias folded into the BatchNorm
        pool = seq[3].kernel_size if len(seq) > 3 else None
        return conv_bn_relu_pool(h, seq[0].weight, seq[0].bias, seq[0].padding, seq[1],
                                 self.training, pool, skip)

    def _stage1(self, out):
        out = self._block(self.conv2, self._block(self.conv1, out))
        return self._block(self.res1[1], self._block(self.res1[0], out), skip=out)   # res1(out) + out

    def _stage2(self, out):
        out = self._block(self.conv4, self._block(self.conv3, out))
        return self._block(self.res2[1], self._block(self.res2[0], out), skip=out)   # res2(out) + out

    def forward(self, out, depth=None, pass_part=None):
        if pass_part == "first" and depth == 0:
            return out
        if out.is_cuda and out.dim() == 4:
            out = out.contiguous(memory_format=torch.channels_last)
            if out.shape[1] == 1 and out.is_contiguous():
                # A one-channel image is both NCHW- and NHWC-dense, and torch resolves the tie to
                # NCHW: conv1 then runs as an NCHW convolution, its 1 GB output comes back NCHW and
                # is re-laid-out three times on the way to conv2 (1.4 ms of an 85 ms bs=256 step,
                # profiles/r2_resnet2d_step_kernels.csv).  Stating the NHWC strides explicitly
                # (same memory) makes the whole network channels_last from the first layer on.
                B, _, H, W = out.shape
                out = out.as_strided((B, 1, H, W), (H * W, 1, W, 1))
        if pass_part == "first":
            out = self._stage1(out)
            if depth == 1:
                return out
            out = self._stage2(out)
            if depth == 2:
                return out
            out = self.flat(self.pool2d(out))
            if depth == 3:
                return out
            return self.linear(out)
        if pass_part == "second":
            if depth <= 0:
                out = self._stage1(out)
            if depth <= 1:
                out = self._stage2(out)
            if depth <= 2:
                out = self.flat(self.pool2d(out))
            if depth <= 3:
                out = self.linear(out)
            return out
        return self.linear(self.flat(self.pool2d(self._stage2(self._stage1(out)))))


def ResNet9(num_classes: int = 2, linear: int = 8192) -> ResNet9_myrtle:
    """Reference factory, models2d.py:86."""
    return ResNet9_myrtle(in_channels=1, num_classes=num_classes, linear=linear)
