"""Top-level transient-to-pose network on HIP kernels.

Same constructor argument (a cfg node), same attribute / state_dict names and the same call contract as the
reference's models/NlosPose.py: `NlosPose(cfg)(meas)` maps a transient volume (B,1,T,H,W) to
(heat-maps (B,24,T/2,H/2,W/2), refined volume (B,1,T,H,W)).  Stage order (reference :49-59):
feature extraction -> light-cone transform over the full time window -> per-volume normalisation to [0,10] ->
U-Net refinement -> heat-map regression on (feature + refinement).  The LCT owns no parameters, so there is
nothing under `feature_propagation.` in the state_dict (as in the reference).
"""
from __future__ import annotations

from torch import nn

from . import _lib
from . import hip_ops as K
from . import ranges as R
from .feature_extraction import FeatureExtraction
from .feature_propagation import FeaturePropagation
from .posenet3d_50 import get_pose_net_50
from .unet3d import UNet3d


class NlosPose(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        _lib.lib()  # the HIP extension is mandatory: fail here, not in the first forward
        m = cfg.MODEL
        if getattr(m, "PRETRAIN_AUTOENCODER", False):
            raise NotImplementedError("a pickled autoencoder module cannot be adopted; load its state_dict into "
                                      "`.autoencoder` after construction")
        if m.BACKBONE != "posenet3d_50":
            # models/NlosPose.py:41-45 constructs VisibleNet + the 2-D posenet for 'posenet2d', but its forward (:49-59) never
            # calls vis_net and feeds the 5-D volume to a Conv2d: there is no runnable reference behaviour to reproduce
            raise NotImplementedError(f"backbone {m.BACKBONE!r} is not built: the reference's own forward cannot run with it "
                                      "(models/NlosPose.py:49-59); its config selects posenet3d_50")
        self.time_begin, self.time_end = 0, m.TIME_SIZE
        self.feature_extraction = FeatureExtraction(m.BASEDIM, m.IN_CHANNELS, stride=1)
        self.feature_propagation = FeaturePropagation(image_size=m.IMAGE_SIZE[0], time_size=m.TIME_SIZE,
                                                      bin_len=m.BIN_LEN, wall_size=m.WALL_SIZE, dnum=m.DNUM,
                                                      dev=cfg.DEVICE)
        self.autoencoder = UNet3d(1, 4)
        self.pose_net = get_pose_net_50(getattr(m, "CONV_PRECISION", "fp32"))
        # Thin-channel (U-Net / FeatureExtraction) convolutions: exact fp32 unless the config asks for 'bf16' BY NAME.  Round 3
        # let 'auto' follow CONV_PRECISION == 'bf16s'; its own 20-step curves then ended +1.5 .. +20 % above fp32 (mean +11 %)
        # against +0.8 .. +7.3 % with an fp32 U-Net -- 6 of 155 ms were not worth a worse training curve, so 'auto' is 'fp32'.
        dp = getattr(m, "DCONV_PRECISION", "auto")
        self.dconv_precision = "fp32" if dp == "auto" else dp

    def forward(self, meas):
        n = meas.shape[0]
        window = ([self.time_begin] * n, [self.time_end] * n)
        K.begin_forward()   # per-weight use counts of the side-stream weight gradients belong to THIS forward (hip_ops)
        prev = K.set_dconv_precision(self.dconv_precision)  # the backward of every node uses what its forward ran with
        try:
            # stage ranges for the profiler's marker trace (ranges.py; no-ops unless HP_ROCTX=1)
            with R.stage("feature_extraction"):
                feature = R.mark_backward(self.feature_extraction(meas), "feature_extraction")
            with R.stage("feature_propagation"):
                feature = self.feature_propagation(feature, *window)
                feature = R.mark_backward(K.normalize_feature(feature), "feature_propagation")
            with R.stage("autoencoder"):
                if isinstance(self.autoencoder, UNet3d) and self.autoencoder.in_channels == 1:
                    refine_feature, summed = self.autoencoder.forward_and_sum(feature)   # `feature + refine` in the same pass
                else:
                    refine_feature = self.autoencoder(feature)
                    summed = K.add(feature, refine_feature)
                summed = R.mark_backward(summed, "autoencoder")
        finally:
            K.set_dconv_precision(prev)
        with R.stage("pose_net"):
            heat = R.mark_backward(self.pose_net(summed), "pose_net")
        return heat, refine_feature
