"""3-D U-Net "autoencoder" of NlosPose.

Drop-in for unet/unet3d.py `UNet3d(in_channels, n_channels)` (:74-104):
DoubleConv = (Conv3d 3^3 pad 1 -> GroupNorm(4) -> ReLU) x 2 (:11-28), Down =
MaxPool3d(2) + DoubleConv (:31-39), Up = trilinear x2 align_corners=True, pad,
cat([skip, up]), DoubleConv (:42-62), Out = 1^3 conv (:65-71).  Same state_dict keys.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from . import hip_ops as ops


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, num_groups=4):
        super().__init__()
        self.double_conv = nn.Sequential(
            nn.Conv3d(in_channels, out_channels, 3, 1, 1), nn.GroupNorm(num_groups, out_channels), nn.Identity(),
            nn.Conv3d(out_channels, out_channels, 3, 1, 1), nn.GroupNorm(num_groups, out_channels), nn.Identity())

    def forward(self, x):
        s = self.double_conv
        x = ops.conv3_gn_relu(x, s[0].weight, s[0].bias, s[1].weight, s[1].bias, s[1].num_groups, s[1].eps)
        return ops.conv3_gn_relu(x, s[3].weight, s[3].bias, s[4].weight, s[4].bias, s[4].num_groups, s[4].eps)


class Down(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.encoder = nn.Sequential(nn.Identity(), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.encoder[1](ops.max_pool3d_2(x))


class Up(nn.Module):
    def __init__(self, in_channels, out_channels, trilinear=True):
        super().__init__()
        assert trilinear, "only the trilinear variant used by NlosPose is provided"
        self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        # upsample x1 and concatenate [skip, up] (the reference's F.pad is a no-op for even sizes)
        return self.conv(ops.upsample_cat(x1, x2))


class Out(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        return ops.conv3d(x, self.conv.weight, self.conv.bias)


class UNet3d(nn.Module):
    def __init__(self, in_channels, n_channels):
        super().__init__()
        _lib.lib()
        self.in_channels, self.n_channels = in_channels, n_channels
        n = n_channels
        self.conv = DoubleConv(in_channels, n)
        self.enc1, self.enc2, self.enc3, self.enc4 = Down(n, 2 * n), Down(2 * n, 4 * n), Down(4 * n, 8 * n), Down(8 * n, 8 * n)
        self.dec1, self.dec2, self.dec3, self.dec4 = Up(16 * n, 4 * n), Up(8 * n, 2 * n), Up(4 * n, n), Up(2 * n, n)
        self.out = Out(n, in_channels)

    def forward(self, x):
        x1 = self.conv(x)
        x2 = self.enc1(x1)
        x3 = self.enc2(x2)
        x4 = self.enc3(x3)
        x5 = self.enc4(x4)
        o = self.dec1(x5, x4)
        o = self.dec2(o, x3)
        o = self.dec3(o, x2)
        o = self.dec4(o, x1)
        return self.out(o)


def freeze_layer(model):
    """unet/unet3d.py:107-119"""
    for _, p in model.named_parameters():
        p.requires_grad = False
