"""hiddenpose_amd: MI355X-native NlosPose transient-to-pose path (HIP kernels behind a C ABI).

Module names mirror the reference: NlosPose, feature_extraction, feature_propagation,
unet3d, posenet3d_50, criterion, optimizer, config.
"""
__all__ = ["NlosPose", "config", "criterion", "feature_extraction", "feature_propagation", "posenet3d_50",
           "unet3d", "optimizer", "testing"]
