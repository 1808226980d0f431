"""Volume-refinement U-Net ("autoencoder") of NlosPose, on HIP kernels.

API and checkpoint layout follow unet/unet3d.py of the reference: `UNet3d(in_channels, n_channels)`
with sub-modules conv / enc1..enc4 / dec1..dec4 / out, whose parameters live under
`<stage>.double_conv.{0,1,3,4}`, `<enc>.encoder.1.*`, `<dec>.conv.*`, `out.conv.*`.
Per stage (reference lines): conv3^3+GroupNorm(4)+ReLU twice (:11-28); max-pool 2 before an encoder stage (:31-39);
trilinear x2 upsampling with align_corners=True, concatenation [skip, up], then the double conv (:42-62);
final 1^3 convolution (:65-71).  Every operation is a kernel of libhiddenpose_hip.so
(csrc/dconv_kernels.hip, csrc/unet_kernels.hip); parameters are held by stock nn.Conv3d / nn.GroupNorm
objects only so that the state_dict keys match.
"""
from __future__ import annotations

from torch import nn

from . import _lib
from . import hip_ops as K

_GROUPS = 4


def _conv_gn(cin: int, cout: int):
    return [nn.Conv3d(cin, cout, kernel_size=3, stride=1, padding=1), nn.GroupNorm(_GROUPS, cout), nn.Identity()]


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, num_groups=_GROUPS):
        super().__init__()
        assert num_groups == _GROUPS
        self.double_conv = nn.Sequential(*_conv_gn(in_channels, out_channels), *_conv_gn(out_channels, out_channels))

    def forward(self, x):
        for conv_idx in (0, 3):
            conv, gn = self.double_conv[conv_idx], self.double_conv[conv_idx + 1]
            x = K.conv3_gn_relu(x, conv.weight, conv.bias, gn.weight, gn.bias, gn.num_groups, gn.eps)
        return x


class Down(nn.Module):
    """2x max-pool, then DoubleConv; parameters under `encoder.1`."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.encoder = nn.Sequential(nn.Identity(), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.encoder[1](K.max_pool3d_2(x))


class Up(nn.Module):
    """Upsample the coarse tensor, stack it behind the skip tensor, DoubleConv; parameters under `conv`."""

    def __init__(self, in_channels, out_channels, trilinear=True):
        super().__init__()
        if not trilinear:
            raise NotImplementedError("NlosPose uses the trilinear variant only")
        self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, coarse, skip):
        return self.conv(K.upsample_cat(coarse, skip))


class Out(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        return K.conv3d(x, self.conv.weight, self.conv.bias)


class UNet3d(nn.Module):
    # (stage name, kind, input multiple of n, output multiple of n); decoder inputs count skip + upsampled channels
    _PLAN = (("enc1", Down, 1, 2), ("enc2", Down, 2, 4), ("enc3", Down, 4, 8), ("enc4", Down, 8, 8),
             ("dec1", Up, 16, 4), ("dec2", Up, 8, 2), ("dec3", Up, 4, 1), ("dec4", Up, 2, 1))

    def __init__(self, in_channels, n_channels):
        super().__init__()
        _lib.lib()
        self.in_channels, self.n_channels = in_channels, n_channels
        self.conv = DoubleConv(in_channels, n_channels)
        for name, kind, mi, mo in self._PLAN:
            setattr(self, name, kind(mi * n_channels, mo * n_channels))
        self.out = Out(n_channels, in_channels)

    def _body(self, x):
        skips = [self.conv(x)]
        for name in ("enc1", "enc2", "enc3", "enc4"):
            skips.append(getattr(self, name)(skips[-1]))
        y = skips.pop()
        for name in ("dec1", "dec2", "dec3", "dec4"):
            y = getattr(self, name)(y, skips.pop())
        return y

    def forward(self, x):
        return self.out(self._body(x))

    def forward_and_sum(self, x):
        """(refined, x + refined): the sum the pose regressor reads (models/NlosPose.py:57) is written by the output
        convolution's own pass."""
        return K.conv3d_sum(self._body(x), self.out.conv.weight, self.out.conv.bias, x)


def freeze_layer(model):
    """Stop training of every parameter of `model` (reference helper of the same name)."""
    for p in model.parameters():
        p.requires_grad_(False)
