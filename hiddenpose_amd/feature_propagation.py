"""Physics layer of NlosPose: the light-cone transform (LCT).

Drop-in for the reference's models/feature_propagation.py:
`FeaturePropagation(image_size, time_size, bin_len, wall_size, mode, material, dnum, dev)`
with `.forward(x, time_begin, time_end)` (:18-44), `LCT` (:46-257) and
`normalize_feature` (:273-286).  All arithmetic runs in libhiddenpose_hip.so
(hp_lct_* / hp_normalize_*); there is no torch or CPU fallback.

Differences from the reference, on purpose:
  * any batch size (the reference indexes 3-element time_begin/time_end lists, B <= 3);
  * constants live on the device of the input tensor (the reference hard-codes 'cuda');
  * partial time windows (tbe, ten) -- NlosPose itself only issues tbe = 0, ten = T (models/NlosPose.py:53) --
    are placed on the zero time axis by hp_lct_time_window before the transform;
  * mode 'bp' (back-projection: inverse filter conj(F) without the Wiener denominator, then the 5^3 Laplacian-of-Gaussian
    filter over a replication-padded volume and a zeroed first time slice, :93-94, :103-107, :246-253) is built.  In the
    reference this file asserts mode == 'lct' and its 'bp' branch calls a filterLaplacian it never imports; the twin
    models/tflct.py (same code, imports utils/helper.py:13-32) runs that branch and is what the goldens come from.
"""
from __future__ import annotations

import threading
import weakref

import numpy as np
import torch
from torch import nn

from . import _lib

_MATERIAL = {"diffuse": 0, "specular": 1}
_MODE = {"lct": 0, "bp": 1}


def filter_laplacian() -> np.ndarray:
    """utils/helper.py:13-32: 5x5x5 Laplacian of a Gaussian (std 1) in float32, mean removed.  Host constants, built
    with the reference's float32 NumPy expression order so that the 125 weights are bit-identical."""
    lim = 2
    d = np.arange(-lim, lim + 1, dtype=np.float32)
    y, x, z = np.meshgrid(d, d, d)
    r2 = x ** 2 + y ** 2 + z ** 2
    std2 = 1.0
    w = np.exp(-r2 / (2 * std2))
    w = w / np.sum(w)
    w1 = w * (r2 - 3 * std2)
    w1 = w1 / (std2 ** 2)
    return w1 - np.mean(w1)


class LCTPlan:
    """Owner of one hp_lct_plan (device constants: inverse PSF spectrum, band tables,
    twiddles).  Replaces LCT._parpareparam + todev (feature_propagation.py:71-184)."""

    def __init__(self, T: int, N: int, bin_len: float, wall_size: float, material: str, device: torch.device,
                 mode: str = "lct"):
        import ctypes as C

        self.T, self.N, self.device = T, N, torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HiddenPoseHipError("LCTPlan needs a HIP device (tensor on %s)" % self.device)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(_lib.lib().hp_lct_plan_create_mode(C.byref(h), T, N, float(bin_len), float(wall_size),
                                                      _MATERIAL[material], _MODE[mode], idx), "hp_lct_plan_create_mode")
        self.handle = h
        self._finalizer = weakref.finalize(self, _lib.lib().hp_lct_plan_destroy, h)

    def workspace_bytes(self, batch: int) -> int:
        return int(_lib.lib().hp_lct_workspace_bytes(self.handle, batch))

    def run(self, x: torch.Tensor, backward: bool) -> torch.Tensor:
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        batch = x.numel() // (self.T * self.N * self.N)
        y = torch.empty_like(x)
        nbytes = self.workspace_bytes(batch)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        fn = _lib.lib().hp_lct_backward if backward else _lib.lib().hp_lct_forward
        with torch.cuda.device(x.device):
            _lib.check(fn(self.handle, x.data_ptr(), y.data_ptr(), batch, ws.data_ptr(), nbytes,
                          _lib.current_stream_handle(x.device)), "hp_lct_backward" if backward else "hp_lct_forward")
        return y

    def invpsf(self):
        n = 8 * self.T * self.N * self.N
        re = np.empty(n, np.float32)
        im = np.empty(n, np.float32)
        _lib.check(_lib.lib().hp_lct_plan_get_invpsf(self.handle, re.ctypes.data, im.ctypes.data), "get_invpsf")
        shp = (2 * self.T, 2 * self.N, 2 * self.N)
        return re.reshape(shp), im.reshape(shp)


class _LCTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan):
        ctx.plan = plan
        return plan.run(x.contiguous(), backward=False)

    @staticmethod
    def backward(ctx, gy):
        return ctx.plan.run(gy.contiguous(), backward=True), None


class _Laplacian5(torch.autograd.Function):
    """'bp' epilogue (:246-253): ReplicationPad3d(2) -> conv3d(5^3 LoG) -> [:, :1] = 0 on (planes, T, H, W)."""

    @staticmethod
    def forward(ctx, v, w125):
        v = v.contiguous()
        p, t, h, w = v.shape
        y = torch.empty_like(v)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().hp_laplacian5_forward(v.data_ptr(), w125.data_ptr(), y.data_ptr(), p, t, h, w,
                                                        _lib.current_stream_handle(v.device)), "hp_laplacian5_forward")
        ctx.save_for_backward(w125)
        return y

    @staticmethod
    def backward(ctx, gy):
        (w125,) = ctx.saved_tensors
        gy = gy.contiguous()
        p, t, h, w = gy.shape
        gx = torch.empty_like(gy)
        with torch.cuda.device(gy.device):
            _lib.check(_lib.lib().hp_laplacian5_backward(gy.data_ptr(), w125.data_ptr(), gx.data_ptr(), p, t, h, w,
                                                         _lib.current_stream_handle(gy.device)), "hp_laplacian5_backward")
        return gx, None


class _TimeWindow(torch.autograd.Function):
    """models/feature_propagation.py:193-200: equal-length samples at per-sample offsets of a zero time axis."""

    @staticmethod
    def forward(ctx, x, tbes, T):
        import ctypes as C

        b, d, t, h, w = x.shape
        x = x.contiguous()
        y = torch.empty(b, d, T, h, w, dtype=torch.float32, device=x.device)
        arr = (C.c_int * b)(*[int(v) for v in tbes])
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_lct_time_window(y.data_ptr(), x.data_ptr(), b, d, t, T, h * w, arr, 0,
                                                     _lib.current_stream_handle(x.device)), "hp_lct_time_window")
        ctx.meta = (arr, t, T)
        return y

    @staticmethod
    def backward(ctx, gy):
        arr, t, T = ctx.meta
        b, d, _, h, w = gy.shape
        gy = gy.contiguous()
        gx = torch.empty(b, d, t, h, w, dtype=torch.float32, device=gy.device)
        with torch.cuda.device(gy.device):
            _lib.check(_lib.lib().hp_lct_time_window(gy.data_ptr(), gx.data_ptr(), b, d, t, T, h * w, arr, 1,
                                                     _lib.current_stream_handle(gy.device)), "hp_lct_time_window")
        return gx, None, None


class LCT(nn.Module):
    """models/feature_propagation.py:46-257 (modes 'lct' and 'bp')."""

    def __init__(self, image_size=256, time_size=128, bin_len=0.01, wall_size=2.0, mode="lct", material="diffuse"):
        super().__init__()
        assert mode in _MODE, f"{mode} is not spported. Feature propagation supports lct and bp"
        assert 2 ** int(np.log2(time_size)) == time_size, "time size should be a power of 2"
        assert material in _MATERIAL
        _lib.lib()  # the HIP extension is mandatory
        self.image_size, self.time_size = int(image_size), int(time_size)
        self.bin_len, self.wall_size, self.mode, self.material = bin_len, wall_size, mode, material
        self._plans = {}
        self._lapw = {}
        self._plock = threading.Lock()

    def todev(self, dev, dnum=1):
        """Reference API (:173-184).  Constants are created per device on first use;
        calling this merely pre-builds the plan for `dev`."""
        d = torch.device("cuda", dev) if isinstance(dev, int) else torch.device(dev)
        if d.type == "cuda":
            self.plan_for(d)
        return self

    def plan_for(self, device: torch.device) -> LCTPlan:
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        with self._plock:
            p = self._plans.get(device)
            if p is None:
                p = LCTPlan(self.time_size, self.image_size, self.bin_len, self.wall_size, self.material, device, self.mode)
                self._plans[device] = p
                if self.mode == "bp":
                    self._lapw[device] = torch.from_numpy(filter_laplacian().astype(np.float32)).reshape(-1).to(device)
        return p

    def forward(self, feture_bxdxtxhxw, tbes=None, tens=None):
        x = feture_bxdxtxhxw
        b, d, t, h, w = x.shape
        if not x.is_cuda:
            raise _lib.HiddenPoseHipError("LCT.forward needs a tensor on a HIP device; there is no CPU path")
        if tbes is not None:
            assert len(tbes) >= b and len(tens) >= b
            for tbe, ten in zip(tbes[:b], tens[:b]):
                assert tbe >= 0 and ten <= self.time_size
                assert ten - tbe == t, f"window [{tbe}, {ten}) does not hold {t} time bins"
            if any(tbe != 0 or ten != self.time_size for tbe, ten in zip(tbes[:b], tens[:b])):
                x = _TimeWindow.apply(x.float(), list(tbes[:b]), self.time_size)
                t = self.time_size
        assert t == self.time_size and h == w == self.image_size
        plan = self.plan_for(x.device)
        y = _LCTFunction.apply(x.reshape(b * d, t, h, w).float(), plan)
        if self.mode == "bp":
            y = _Laplacian5.apply(y, self._lapw[plan.device])
        return y.view(b, d, t, h, w)


class FeaturePropagation(nn.Module):
    """models/feature_propagation.py:18-44.  (B,C,T,H,W) -> (B,C,T,H,W)."""

    def __init__(self, image_size=256, time_size=512, bin_len=0.01, wall_size=2.0, mode="lct", material="diffuse",
                 dnum=1, dev="cpu"):
        super().__init__()
        assert mode in _MODE, f"{mode} is not spported. Feature propagation supports lct and bp"
        self.method = LCT(int(image_size), time_size, bin_len, wall_size, mode=mode, material=material)

    def forward(self, x, time_begin=None, time_end=None):
        return self.method(x, time_begin, time_end)


def normalize_feature(data_bxcxdxhxw):
    """models/feature_propagation.py:273-286 (no ReLU: its result is discarded on :274)."""
    from . import hip_ops

    return hip_ops.normalize_feature(data_bxcxdxhxw)


def normalize(data_bxcxdxhxw):
    """models/feature_propagation.py:260-270: (x - min) / (max(x - min) + 1e-15) per (b, c)."""
    from . import hip_ops

    return hip_ops.normalize_feature(data_bxcxdxhxw) * 0.1   # the kernel's gain is 10 (normalize_feature)


class VisibleNet(nn.Module):
    """models/feature_propagation.py:289-312: relu -> normalize -> x 1e5 -> top-4 along depth; values and depth
    coordinates (D-1-idx)/(D-1) concatenated on the channel axis.  (B,C,D,H,W) -> (B,2C,4,H,W).  The reference builds
    it only for its `posenet2d` backbone and never calls it (models/NlosPose.py:41-59); inference only."""

    def __init__(self, basedim, layernum=0):
        super().__init__()
        self.layernum = layernum

    @torch.no_grad()
    def forward(self, x):
        from . import hip_ops

        return hip_ops.visible_projection(x)
