"""Dev-container-only helpers that make the read-only reference checkout at
/root/reference importable on a CPU-only box with torch 2.x, so that
make_goldens.py can capture golden vectors from the reference's own code.

Never imported by tests, bench.py or the package: /root/reference does not
exist on the GPU box.  Nothing here restates reference arithmetic; it only
(1) provides empty stand-ins for third-party modules the hot path never calls
    (plots, dataset readers, config library),
(2) maps the two legacy FFT entry points removed in torch>=1.8 onto torch.fft,
(3) redirects the hard-coded 'cuda' device of models/feature_propagation.py:109
    and the CUDA-only arange broadcast of utils/criterion.py:118-120 to the CPU.
"""
from __future__ import annotations

import sys
import types

import torch

REF_ROOT = "/root/reference"


class _AttrDict(dict):
    """Minimal stand-in for yacs.config.CfgNode (attribute access + freeze hooks)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def defrost(self):
        pass

    def freeze(self):
        pass

    def clone(self):
        return self


def _stub(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install() -> None:
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    sys.dont_write_bytecode = True

    # (1) third-party modules that are absent here and unused by the hot path
    yacs = _stub("yacs")
    yacs.config = _stub("yacs.config", CfgNode=_AttrDict)
    _stub("cv2", imshow=lambda *a, **k: None, waitKey=lambda *a, **k: None)
    _stub("torchsummary", summary=lambda *a, **k: None)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.transforms.functional = _stub("torchvision.transforms.functional")
    tv.models = _stub("torchvision.models")
    tv.models.resnet = _stub("torchvision.models.resnet", BasicBlock=object, Bottleneck=object)
    _stub("mat73")
    timm = _stub("timm")
    timm.models = _stub("timm.models")
    timm.models.layers = _stub("timm.models.layers")
    timm.models.layers.weight_init = _stub("timm.models.layers.weight_init", trunc_normal_=torch.nn.init.trunc_normal_)

    # (2) legacy FFT names (torch<1.8) used by models/feature_propagation.py:228,235
    def rfft(x, signal_ndim, onesided=True):
        assert signal_ndim == 3 and not onesided
        return torch.view_as_real(torch.fft.fftn(x, dim=(-3, -2, -1)))

    def ifft(x, signal_ndim):
        assert signal_ndim == 3
        return torch.view_as_real(torch.fft.ifftn(torch.view_as_complex(x.contiguous()), dim=(-3, -2, -1)))

    torch.rfft = rfft
    torch.ifft = ifft

    # (3a) LCT constructor ends with self.todev('cuda', 1)
    import models.feature_propagation as fp

    orig_todev = fp.LCT.todev

    def todev(self, dev, dnum):
        if isinstance(dev, str) and dev.startswith("cuda"):
            dev = "cpu"
        if isinstance(dev, int):
            dev = "cpu"
        return orig_todev(self, dev, dnum)

    fp.LCT.todev = todev

    # (3b) criterion builds its index ramps with torch.cuda.comm.broadcast(arange.cuda())
    class _CpuFloatTensor:
        pass

    def _type(self, t=None, *a, **k):
        if t is _CpuFloatTensor:
            return self.to(torch.float32)
        return _orig_type(self, t, *a, **k)

    _orig_type = torch.Tensor.type
    torch.Tensor.type = _type
    torch.cuda.FloatTensor = _CpuFloatTensor
    import importlib

    importlib.import_module("torch.cuda.comm")  # submodule is not loaded by `import torch`
    torch.cuda.comm.broadcast = lambda t, devices=None: [t]


def make_cfg(T: int, N: int, bin_len: float, device="cpu"):
    """cfg node with the attributes models/NlosPose.py:18-47 reads."""
    c = _AttrDict()
    c.DEVICE = device
    c.MODEL = _AttrDict(
        TIME_SIZE=T, IMAGE_SIZE=[N, N], BIN_LEN=bin_len, WALL_SIZE=2.0, DNUM=1,
        BASEDIM=1, IN_CHANNELS=1, PRETRAIN_AUTOENCODER=False, PRETRAIN_AUTOENCODER_PATH="",
        BACKBONE="posenet3d_50", NUM_JOINTS=24,
    )
    c.DATASET = _AttrDict(NUM_JOINTS=24, HEATMAP_SIZE=[N // 2, N // 2, T // 2])
    c.TRAIN = _AttrDict(OPTIMIZER="adam", LR=1e-3, LR_STEP=[2, 4, 13], LR_FACTOR=0.2, BATCH_SIZE=2)
    return c
