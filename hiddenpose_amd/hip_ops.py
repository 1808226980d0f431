"""Operator layer between the reference-shaped nn.Modules and libhiddenpose_hip.so.

Every function here is one fused stage of the hot path.  A stage
either calls a hand-written HIP kernel through the C ABI (wrapped in a
torch.autograd.Function so that optimisers and torch.distributed stay stock), or --
for rows whose kernel has not landed yet -- the stock PyTorch-ROCm device operator.
`HIP_STAGES` / `ATEN_STAGES` say which is which; DESIGN.md tracks the same table.
There is no CPU path: tensors must live on a HIP device.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

HIP_STAGES = {"lct_forward", "lct_backward"}
ATEN_STAGES = {"feature_extraction", "normalize_feature", "unet3d", "posenet3d_50", "softmax_integral",
               "bce_dice"}


def _need_cuda(x: torch.Tensor, what: str) -> None:
    if not x.is_cuda:
        from ._lib import HiddenPoseHipError

        raise HiddenPoseHipError(f"{what}: tensor is on {x.device}; this package has no CPU path")


# ---------------------------------------------------------------- FeatureExtraction (rows A1, A2)
def conv3d_reppad(x, w, b, stride=1):
    return F.conv3d(F.pad(x, (1, 1, 1, 1, 1, 1), mode="replicate"), w, b, stride=stride)


def feature_extraction_fused(x, fe):
    _need_cuda(x, "feature_extraction")
    a = conv3d_reppad(x, fe.conv1[1].weight, fe.conv1[1].bias)
    a = fe.conv1[3](fe.conv1[2](a))
    return a + F.conv3d(x, fe.weights, None, stride=1, padding=1)


# ---------------------------------------------------------------- normalize_feature (row C8)
def normalize_feature(x):
    """(x - min)/(max(x - min) + 1e-15) * 10 per (b, c); NO ReLU (feature_propagation.py:273-286)."""
    _need_cuda(x, "normalize_feature")
    b, c = x.shape[:2]
    f = x.reshape(b, c, -1)
    z = f - f.min(2, keepdim=True)[0]
    return (z / (z.max(2, keepdim=True)[0] + 1e-15) * 10.0).view_as(x)


# ---------------------------------------------------------------- UNet3d (row U1)
def conv3d(x, w, b=None, stride=1, padding=0):
    return F.conv3d(x, w, b, stride=stride, padding=padding)


def conv3_gn_relu(x, w, b, gw, gb, groups, eps):
    return F.relu(F.group_norm(F.conv3d(x, w, b, padding=1), groups, gw, gb, eps))


def max_pool3d_2(x):
    return F.max_pool3d(x, 2, 2)


def upsample_trilinear_2x(x):
    return F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True)


# ---------------------------------------------------------------- posenet3d_50 (rows P1-P3)
def conv_bn_act(x, conv, bn, relu=True, residual=None):
    y = bn(F.conv3d(x, conv.weight, None, stride=conv.stride, padding=conv.padding))
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def deconv_bn_relu(x, deconv, bn):
    return F.relu(bn(F.conv_transpose3d(x, deconv.weight, None, stride=2, padding=1)))


def stem_conv_bn_relu_pool(x, conv, bn):
    _need_cuda(x, "posenet3d_50")
    return F.max_pool3d(F.relu(bn(F.conv3d(x, conv.weight, None, stride=1, padding=3))), 3, 2, 1)


# ---------------------------------------------------------------- decode + losses (rows L1-L3)
def softmax_integral(preds, num_joints, W, H, D):
    """soft-argmax in voxel units, order (x,y,z) = (W,H,D axis) (utils/criterion.py:96-153)."""
    _need_cuda(preds, "softmax_integral")
    B = preds.shape[0]
    p = F.softmax(preds.reshape(B, num_joints, -1), 2).reshape(B, num_joints, D, H, W)
    ar = lambda n: torch.arange(n, dtype=p.dtype, device=p.device)
    ax = (p.sum(dim=(2, 3)) * ar(W)).sum(2, keepdim=True)
    ay = (p.sum(dim=(2, 4)) * ar(H)).sum(2, keepdim=True)
    az = (p.sum(dim=(3, 4)) * ar(D)).sum(2, keepdim=True)
    return torch.cat((ax, ay, az), dim=2).reshape(B, num_joints * 3)


def bce_dice(logits, targets, eps=1e-9):
    _need_cuda(logits, "bce_dice")
    prob = torch.sigmoid(logits)
    dice = (2.0 * (prob * targets).sum() + eps) / (prob.sum() + targets.sum())
    return F.binary_cross_entropy_with_logits(logits, targets) + (1.0 - dice)
