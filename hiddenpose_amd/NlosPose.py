"""NlosPose: transient measurement volume -> 24-joint heat-maps.

Drop-in for models/NlosPose.py `NlosPose(cfg)` (:13-59): same submodule names (hence
the same 423-tensor state_dict; nothing under `feature_propagation.`), same
`forward(meas (B,1,T,H,W)) -> (heatmap (B,24,T/2,H/2,W/2), refine_feature (B,1,T,H,W))`.
"""
from __future__ import annotations

from torch import nn

from . import _lib
from . import hip_ops as ops
from .feature_extraction import FeatureExtraction
from .feature_propagation import FeaturePropagation
from .posenet3d_50 import get_pose_net_50
from .unet3d import UNet3d


class NlosPose(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        _lib.lib()  # fail loudly if the HIP extension is missing
        self.time_begin = 0
        self.time_end = cfg.MODEL.TIME_SIZE
        self.feature_extraction = FeatureExtraction(basedim=cfg.MODEL.BASEDIM, in_channels=cfg.MODEL.IN_CHANNELS,
                                                    stride=1)
        self.feature_propagation = FeaturePropagation(
            time_size=cfg.MODEL.TIME_SIZE, image_size=cfg.MODEL.IMAGE_SIZE[0], wall_size=cfg.MODEL.WALL_SIZE,
            bin_len=cfg.MODEL.BIN_LEN, dnum=cfg.MODEL.DNUM, dev=cfg.DEVICE)
        if getattr(cfg.MODEL, "PRETRAIN_AUTOENCODER", False):
            raise NotImplementedError("PRETRAIN_AUTOENCODER loads a pickled module (models/NlosPose.py:34-35); "
                                      "load its state_dict into .autoencoder instead")
        self.autoencoder = UNet3d(in_channels=1, n_channels=4)
        if cfg.MODEL.BACKBONE != "posenet3d_50":
            raise NotImplementedError(f"backbone {cfg.MODEL.BACKBONE!r}: only posenet3d_50 (config_noise.py:35) is built")
        self.pose_net = get_pose_net_50()

    def forward(self, meas):
        b = meas.shape[0]
        meas = self.feature_extraction(meas)
        feature = self.feature_propagation(meas, [self.time_begin] * b, [self.time_end] * b)
        feature = ops.normalize_feature(feature)
        refine_feature = self.autoencoder(feature)
        output = self.pose_net(ops.add(feature, refine_feature))
        return output, refine_feature
