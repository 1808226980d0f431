"""Decode and losses of the NlosPose training/eval callers.

Drop-in for the live symbols of utils/criterion.py: `softmax_integral_tensor`
(:129-153), `L2JointLocationLoss` (:66-87), `weighted_mse_loss` (:156-162),
`DiceLoss`/`BCEDiceLoss` (:348-385).  Joint coordinates are returned in heat-map voxel
units, order (x, y, z) = (W-, H-, D-axis expectation), concatenated to (B, 3J); the
[-0.5, 0.5] normalisation is commented out in the reference and is not applied.
"""
from __future__ import annotations

import torch
from torch import nn

from . import hip_ops as ops


def softmax_integral_tensor(preds, num_joints, output_3d, hm_width, hm_height, hm_depth):
    assert output_3d, "Not Implemented!"
    return ops.softmax_integral(preds, num_joints, hm_width, hm_height, hm_depth)


def weighted_mse_loss(input, target, weights, size_average):
    """sum((input - target)^2 * weights) [/ len(input)]  (utils/criterion.py:156-162) in one HIP kernel."""
    return ops.weighted_mse(input, target, weights, size_average)


class L2JointLocationLoss(nn.Module):
    def __init__(self, output_3d, size_average=True, reduce=True):
        super().__init__()
        self.size_average, self.reduce, self.output_3d = size_average, reduce, output_3d

    def forward(self, preds, *args):
        gt_joints, gt_joints_vis = args[0], args[1]
        assert not gt_joints.requires_grad and not gt_joints_vis.requires_grad
        num_joints = int(gt_joints_vis.shape[1] / 3)
        pred = softmax_integral_tensor(preds, num_joints, self.output_3d, preds.shape[-1], preds.shape[-2],
                                       preds.shape[-3])
        return weighted_mse_loss(pred, gt_joints, gt_joints_vis, self.size_average)


class DiceLoss(nn.Module):
    def __init__(self, eps: float = 1e-9):
        super().__init__()
        self.eps = eps

    def forward(self, logits, targets):
        p = torch.sigmoid(logits)
        return 1.0 - (2.0 * (p * targets).sum() + self.eps) / (p.sum() + targets.sum())


class BCEDiceLoss(nn.Module):
    """BCEWithLogits(mean) + Dice over the whole batch (one fused reduction).

    `global_batch=True` (data parallelism): the Dice sums are taken over every rank's samples, as the reference's
    single process does over its whole batch (utils/criterion.py:358-368); with gradient averaging over ranks the
    result is exactly the loss and gradient of the concatenated batch (hip_ops._BceDiceGlobal)."""

    def __init__(self, global_batch: bool = False, group=None):
        super().__init__()
        self.global_batch, self.group = global_batch, group

    def forward(self, logits, targets):
        assert logits.shape == targets.shape
        return ops.bce_dice(logits, targets, global_batch=self.global_batch, group=self.group)
