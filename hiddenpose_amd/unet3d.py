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

import torch
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

    def forward(self, x, out=None):
        for conv_idx in (0, 3):
            conv, gn = self.double_conv[conv_idx], self.double_conv[conv_idx + 1]
            x = K.conv3_gn_relu(x, conv.weight, conv.bias, gn.weight, gn.bias, gn.num_groups, gn.eps,
                                out if conv_idx == 3 else None)
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

    def forward(self, coarse, skip, buf=None):
        return self.conv(K.upsample_cat(coarse, skip, buf))


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
        # Batch 1 (the 256 x 256 x 1024 run): a level's output is PRODUCED in the first half of its decoder's concatenation
        # buffer (for one sample a channel slice of a planar tensor is contiguous), so the skip is never copied.  Every level
        # concatenates as many upsampled channels as skip channels (_PLAN).
        def cat_buffer(like, c, shape):
            return torch.empty(1, 2 * c, *shape, dtype=like.dtype, device=like.device) if like.shape[0] == 1 else None

        n = self.n_channels
        bufs = [cat_buffer(x, n, x.shape[2:])]
        skips = [self.conv(x, None if bufs[0] is None else bufs[0][:, :n])]
        for li, name in enumerate(("enc1", "enc2", "enc3", "enc4")):
            # a level's output feeds the next level's pool and the decoder's concatenation: one node for both, so that its two
            # gradients are summed inside the pool's backward pass (hip_ops._PoolSkip)
            skip, pooled = K.pool_and_skip(skips[-1])
            skips[-1] = skip
            cout = self._PLAN[li][3] * n
            buf = cat_buffer(x, cout, pooled.shape[2:]) if name != "enc4" else None   # enc4's output is not a skip
            bufs.append(buf)
            skips.append(getattr(self, name).encoder[1](pooled, None if buf is None else buf[:, :cout]))
        y = skips.pop()
        bufs.pop()
        for name in ("dec1", "dec2", "dec3", "dec4"):
            y = getattr(self, name)(y, skips.pop(), bufs.pop())
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
