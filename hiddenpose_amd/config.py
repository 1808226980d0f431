"""Configuration node for the NlosPose path.

The reference keeps a module-level yacs tree (config/config_noise.py:4-112) that
train.py/test.py patch in place (utils/record.py:42-60, train.py:77-86).  yacs is
not a dependency here: `CfgNode` is a small attribute dictionary with the same
`defrost()/freeze()/clone()` surface, and `NlosPose(cfg)` accepts ANY object that
exposes the attributes the reference reads (models/NlosPose.py:18-47), so a real
yacs node from the reference works unchanged.
"""
from __future__ import annotations

import copy


class CfgNode(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        object.__setattr__(self, "_frozen", False)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"attempted to modify frozen CfgNode key {k!r}")
        self[k] = v

    def _set_frozen(self, f: bool) -> None:
        object.__setattr__(self, "_frozen", f)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(f)

    def defrost(self) -> None:
        self._set_frozen(False)

    def freeze(self) -> None:
        self._set_frozen(True)

    def clone(self) -> "CfgNode":
        c = copy.deepcopy(self)
        c._set_frozen(False)
        return c

    def __deepcopy__(self, memo):
        c = CfgNode()
        for k, v in self.items():
            c[k] = copy.deepcopy(v, memo)
        return c


def get_cfg_defaults() -> CfgNode:
    """Keys and values of config/config_noise.py that the hot path and its callers
    read (model shape, optimiser, schedule, loss)."""
    c = CfgNode()
    c.OUTPUT_DIR = ""
    c.DEVICE = 0
    c.NUM_WOKERS = 8
    c.LOG_DIR = "./log"
    c.RESULT_DIR = "./result"
    c.PHASE = "train"
    c.MODEL = CfgNode(
        NAME="nlospose", DNUM=1, BASEDIM=1, BIN_LEN=0.01, WALL_SIZE=2.0, IN_CHANNELS=1, OUT_CHANNELS=1,
        TIME_SIZE=512, IMAGE_SIZE=[256, 256], HEATMAP_SIZE=[64, 64, 64], MODE="lct",
        COORD_REPRESENTATION="3DHeatmap", NUM_JOINTS=24, BACKBONE="posenet3d_50",
        PRETRAIN_AUTOENCODER=False, PRETRAIN_AUTOENCODER_PATH="./lib/nlos_unet.pth", LOCATION="",
        # not a reference key: arithmetic of the regressor's convolution GEMMs, "fp32" (default, exact) or
        # "bf16" (bf16 operands / fp32 accumulation; LCT, U-Net, norms and losses stay fp32)
        CONV_PRECISION="fp32",
        # not a reference key: arithmetic of the thin-channel 3^3 convolutions of the U-Net / FeatureExtraction (layers with
        # more than one channel): "fp32" (exact) or "bf16" (bf16 operands on v_mfma_f32_4x4x4_16b_bf16, fp32 accumulation,
        # fp32 tensors); "auto" = bf16 exactly when CONV_PRECISION is "bf16s" (BASELINE configs[2]: bf16 with fp32 LCT)
        DCONV_PRECISION="auto",
    )
    c.DATASET = CfgNode(NAME="NlosPoseDataset", NUM_JOINTS=24, HEATMAP_SIZE=[64, 64, 64], VOL_SIZE=[256, 256, 256],
                        DAWNSAMPLE_CNT=1, PHASE="train", TRAIN_PATH="", TEST_PATH="")
    c.TRAIN = CfgNode(OPTIMIZER="adam", LR=0.001, LR_FACTOR=0.2, LR_STEP=[2, 4, 13], BATCH_SIZE=2,
                      BEGIN_EPOCH=0, END_EPOCH=15)
    c.TEST = CfgNode(TYPE="pose_v2", BATCH_SIZE=2)
    c.LOSS = CfgNode(TYPE="L2JointLocationLoss")
    c.RESULT = CfgNode(FINAL_OUTPUT_DIR="./checkpoints")
    return c


def update_config_t128_128x128(cfg) -> None:
    """train.py:77-86 / test.py:68-77: the shape every reference run actually uses."""
    cfg.defrost()
    cfg.MODEL.TIME_SIZE = 128
    cfg.MODEL.IMAGE_SIZE = [128, 128]
    cfg.MODEL.BIN_LEN = 0.04
    cfg.freeze()


def make_cfg(T: int, N: int, bin_len: float | None = None, device=0, conv_precision: str = "fp32") -> CfgNode:
    """Config for a (T, N, N) transient; bin_len defaults to the value that keeps the
    time range at 5.12 m as in both reference shapes (128 x 0.04, 512 x 0.01)."""
    c = get_cfg_defaults()
    c.DEVICE = device
    c.MODEL.TIME_SIZE = T
    c.MODEL.IMAGE_SIZE = [N, N]
    c.MODEL.BIN_LEN = 5.12 / T if bin_len is None else bin_len
    c.DATASET.HEATMAP_SIZE = [N // 2, N // 2, T // 2]
    c.MODEL.HEATMAP_SIZE = [N // 2, N // 2, T // 2]
    c.MODEL.CONV_PRECISION = conv_precision
    return c
