"""3-D ResNet-50 + deconvolution head: the pose regressor of NlosPose.

Drop-in for models/posenet3d_50.py: `get_pose_net_50()` (:308-318) builds
ResNet(Bottleneck, [3,4,6,3]) with a 7^3 stride-1 single-channel stem, BatchNorm3d,
MaxPool3d(3,2,1), shortcut type 'B', and DeconvHead(2048, 3 x ConvTranspose3d(k4,s2,p1)
-> 256, final 1^3 conv -> 24) (:98-153, :156-270).  Same state_dict keys and init.
(B,1,T,H,W) -> (B,24,T/2,H/2,W/2).
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from . import hip_ops as ops

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, in_planes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv3d(in_planes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # The block input feeds two consumers; their gradients are summed inside the data-gradient GEMM that
        # finishes last (hip_ops.GradLink) instead of by a separate accumulation pass.
        link = ops.GradLink() if (torch.is_grad_enabled() and x.requires_grad) else None
        # the BatchNorm-backward sums of the PREVIOUS block's output unit are taken by this block's conv1 data gradient when that is
        # where the block-input gradient becomes complete: identity blocks with a gradient link (hip_ops.offer_res_bn)
        prev_out = ops.take_res_bn(x)
        if self.downsample is not None or link is None:
            prev_out = None
        res = rl = None
        if self.downsample is not None:
            # The shortcut unit is recorded FIRST so that its backward runs LAST: conv1's dense data gradient then
            # writes d(block input) and the strided shortcut gradient is added to it in place at the voxels it
            # reaches (the other order zero-fills a full-size tensor that conv1 has to read back as its addend).
            # The unit takes the block's raw output gradient plus the output sign mask (hip_ops.ResLink).
            rl = ops.ResLink() if torch.is_grad_enabled() else None
            res = ops.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False, link_in=link, res_link=rl)
        # bn1 and bn2 each feed ONE convolution: that convolution's data gradient takes their backward sums (hip_ops.BnLink)
        b1, b2 = ops.bn_link(), ops.bn_link()
        o = ops.conv_bn_act(x, self.conv1, self.bn1, relu=True, link_in=link, bn_in=prev_out, bn_out=b1)
        o = ops.conv_bn_act(o, self.conv2, self.bn2, relu=True, bn_in=b1, bn_out=b2)
        if self.downsample is not None:
            if link is not None:
                link.arrivals = 2      # conv1 and the shortcut convolution both produce d(block input)
            return ops.conv_bn_act(o, self.conv3, self.bn3, relu=True, residual=res, res_link=rl, bn_in=b2)
        if link is not None:
            link.arrivals = 1          # conv1 adds the identity-shortcut gradient parked by conv3's node
        b3 = ops.bn_link() if (link is not None and ops._BN_FUSE_RES) else None
        y = ops.conv_bn_act(o, self.conv3, self.bn3, relu=True, residual=x, link_out=link, bn_in=b2, bn_out=b3)
        ops.offer_res_bn(y, b3)
        return y


class DeconvHead(nn.Module):
    def __init__(self, in_channels, num_layers, num_filters, kernel_size, conv_kernel_size, num_joints):
        super().__init__()
        assert kernel_size == 4 and conv_kernel_size == 1
        self.features = nn.ModuleList()
        for i in range(num_layers):
            cin = in_channels if i == 0 else num_filters
            self.features.append(nn.ConvTranspose3d(cin, num_filters, 4, stride=2, padding=1, bias=False))
            self.features.append(nn.BatchNorm3d(num_filters))
            self.features.append(nn.Identity())
        self.features.append(nn.Conv3d(num_filters, num_joints, 1, bias=True))
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                nn.init.normal_(m.weight, mean=0, std=0.001)
                if getattr(m, "bias", None) is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        f = self.features
        prev = ops.take_res_bn(x)      # layer4's last block: its output feeds the first deconvolution only
        for i in range(0, len(f) - 1, 3):
            nxt = ops.bn_link()
            x = ops.deconv_bn_relu(x, f[i], f[i + 1], bn_in=prev, bn_out=nxt)
            prev = nxt
        return ops.head_conv_to_ncdhw(x, f[-1], prev)


class ResNet(nn.Module):
    conv_precision = "fp32"  # "bf16": bf16-operand MFMA for every convolution GEMM (hip_ops.set_conv_precision)

    def __init__(self, n_input_channels=1):
        super().__init__()
        _lib.lib()
        self.in_planes = 64
        self.conv1 = nn.Conv3d(n_input_channels, 64, 7, stride=1, padding=3, bias=False)
        self.bn1 = nn.BatchNorm3d(64)
        self.layer1 = self._make_layer(PLANES[0], LAYERS[0], 1)
        self.layer2 = self._make_layer(PLANES[1], LAYERS[1], 2)
        self.layer3 = self._make_layer(PLANES[2], LAYERS[2], 2)
        self.layer4 = self._make_layer(PLANES[3], LAYERS[3], 2)
        self.head = DeconvHead(2048, 3, 256, 4, 1, 24)
        # reference init order (:207-214): the kaiming pass also overrides the head's 1^3 conv
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.in_planes != planes * 4:
            ds = nn.Sequential(nn.Conv3d(self.in_planes, planes * 4, 1, stride=stride, bias=False),
                               nn.BatchNorm3d(planes * 4))
        layers = [Bottleneck(self.in_planes, planes, stride, ds)]
        self.in_planes = planes * 4
        layers += [Bottleneck(self.in_planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        prev = ops.set_conv_precision(self.conv_precision)
        ops.begin_forward()   # per-weight use counts of the side-stream weight gradients belong to THIS forward
        try:
            x = ops.stem_conv_bn_relu_pool(x, self.conv1, self.bn1)
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
            return self.head(x)
        finally:
            ops.set_conv_precision(prev)


def get_pose_net_50(conv_precision: str = "fp32"):
    net = ResNet(n_input_channels=1)
    net.conv_precision = conv_precision
    return net
