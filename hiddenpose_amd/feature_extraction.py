"""Single-channel 3-D feature extraction stencil of NlosPose.

Drop-in for models/feature_extraction.py `FeatureExtraction` (:122-171) and
`ResConv3D` (:228-256): same constructor, same state_dict keys
(`weights`, `conv1.1.*`, `conv1.{2,3}.tmp.{1,4}.*`), same forward contract
(B,1,T,H,W) -> (B,basedim,T,H,W).

y = ResConv(ResConv(conv(reppad(x)))) + conv_zeropad(x, weights)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from . import hip_ops as ops


class ResConv3D(nn.Module):
    """leaky(x + conv(reppad(leaky(conv(reppad(x)), .2))), .2)  -- parameter holder;
    indices 1 and 4 of `tmp` are the two convolutions as in the reference."""

    def __init__(self, basedim, inplace=False):
        super().__init__()
        self.tmp = nn.Sequential(
            nn.Identity(), nn.Conv3d(basedim, basedim, 3, padding=0, bias=True),
            nn.Identity(), nn.Identity(), nn.Conv3d(basedim, basedim, 3, padding=0, bias=True))

    def forward(self, x):
        # both LeakyReLUs and the residual add ride in the convolutions' epilogues
        h = ops.conv3d_reppad(x, self.tmp[1].weight, self.tmp[1].bias, slope=0.2)
        return ops.conv3d_reppad(h, self.tmp[4].weight, self.tmp[4].bias, residual=x, slope=0.2)


class FeatureExtraction(nn.Module):
    def __init__(self, basedim, in_channels, stride=2, norm=None):
        super().__init__()
        assert in_channels == 1, f"input channels should be 1, not {in_channels}"
        _lib.lib()
        self.stride = stride
        w = np.zeros((1, 1, 3, 3, 3), dtype=np.float32)
        w[:, :, 1:, 1:, 1:] = 1.0
        self.weights = nn.Parameter(torch.from_numpy(w / np.sum(w)))
        self.conv1 = nn.Sequential(
            nn.Identity(), nn.Conv3d(in_channels, basedim, 3, padding=0, stride=stride, bias=True),
            ResConv3D(basedim), ResConv3D(basedim))

    def forward(self, x):
        if self.stride == 1 and self.conv1[1].out_channels == 1:
            return ops.feature_extraction_fused(x, self)
        raise NotImplementedError("FeatureExtraction: only stride 1, basedim 1 (the NlosPose configuration) is built")
