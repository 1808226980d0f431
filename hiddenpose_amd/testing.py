"""Deterministic synthetic inputs and weight filler shared by tests, bench.py,
smoke() and the golden generator (tests/golden/make_goldens.py).

Nothing here is part of the hot path.  Neither the reference's trained weights
nor its datasets exist offline (reference README.md:36,111), so parity is
measured on synthetic transients with weights filled by a closed, seeded rule
that both the reference modules and this package's modules receive by
parameter *name* (the state_dict schemas are identical).
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng_for(name: str, salt: int = 0) -> np.random.Generator:
    seed = zlib.crc32(name.encode("utf-8")) ^ (salt * 0x9E3779B1 & 0xFFFFFFFF)
    return np.random.Generator(np.random.PCG64(seed))


def fill_value(name: str, shape, head_gain: float = 6.0, smooth: bool = False) -> torch.Tensor:
    """Value for the parameter/buffer called `name` with `shape`.

    smooth=True is the SECOND filler: the same weights, but every normalisation layer that feeds a
    ReLU gets gain 0.5 and bias +2 (BatchNorm of the regressor, GroupNorm of the U-Net), so that a normalised
    pre-activation is N(2, 0.5^2) and only ~3e-5 of them lie below the kink.  The default filler puts the kink in the
    middle of the distribution: half of all ReLU decisions hinge on the last bits of a pre-activation, and a float32
    evaluation of the REFERENCE differs from its own float64 evaluation by 4e-4 .. 3e-2 per gradient.  With the masks
    decided far from rounding noise the same comparison is ~1e-6, and an end-to-end gradient test can carry a 1e-3 bar
    that a wrong backward term cannot pass.

    conv / deconv weights : N(0, 2/fan_in)        (activations stay O(1))
    norm weights          : 1 + 0.1 N(0,1);  norm/conv biases: 0.1 N(0,1)
    running_mean          : 0.1 N(0,1);      running_var: 1 + 0.1 |N(0,1)|
    FeatureExtraction.weights keeps the reference's 8-tap box filter
    (models/feature_extraction.py:141-145) plus a small perturbation.
    The last 1x1x1 conv of the pose head is scaled by `head_gain` so the
    64^3 soft-max is peaked; with the reference's own init every joint
    decodes to the volume centre and any output would pass.
    """
    shape = tuple(int(s) for s in shape)
    g = _rng_for(name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        return torch.from_numpy((0.1 * g.standard_normal(shape)).astype(np.float32))
    if leaf == "running_var":
        return torch.from_numpy((1.0 + 0.1 * np.abs(g.standard_normal(shape))).astype(np.float32))
    if name == "feature_extraction.weights":
        w = np.zeros(shape, dtype=np.float32)
        w[:, :, 1:, 1:, 1:] = 1.0 / 8.0
        w += (0.02 * g.standard_normal(shape)).astype(np.float32)
        return torch.from_numpy(w)
    if smooth and len(shape) == 1 and (".bn" in name or ".downsample.1." in name or name.startswith("pose_net.bn1.")
                                        or ".head.features." in name or name.startswith("autoencoder.")):
        is_norm = leaf in ("weight", "bias") and not name.endswith("conv.bias") and not name.endswith("features.9.bias") \
            and not (name.startswith("autoencoder.") and (name.endswith(".0.bias") or name.endswith(".3.bias")))
        if is_norm:
            if leaf == "weight":
                gain = 0.25 if name.endswith("bn3.weight") else 0.5
                return torch.from_numpy((gain * (1.0 + 0.1 * g.standard_normal(shape))).astype(np.float32))
            # bn3 / downsample outputs are summed before their ReLU: half the offset each keeps the sum at ~+2
            off = 1.0 if (name.endswith("bn3.bias") or ".downsample.1." in name) else 2.0
            return torch.from_numpy((off + 0.1 * g.standard_normal(shape)).astype(np.float32))
    if len(shape) == 1:
        if leaf == "weight":
            # damp the residual branches so eval-mode activations (running stats ~ N(0,1),
            # i.e. no real normalisation) stay O(1) through the 16 bottlenecks
            gain = 0.25 if name.endswith("bn3.weight") else 0.7 if name.endswith("downsample.1.weight") else 1.0
            return torch.from_numpy((gain * (1.0 + 0.1 * g.standard_normal(shape))).astype(np.float32))
        return torch.from_numpy((0.1 * g.standard_normal(shape)).astype(np.float32))
    # conv (Cout,Cin,k,k,k) or transposed conv (Cin,Cout,k,k,k)
    fan_in = int(np.prod(shape[1:]))
    if ".head.features." in name and len(shape) == 5 and shape[2] == 4:
        # ConvTranspose3d k4 s2 p1: each output voxel sees 2^3 taps per input channel
        fan_in = shape[0] * 8
    std = np.sqrt(2.0 / fan_in)
    if name.endswith("head.features.9.weight"):
        std *= head_gain
    if name.endswith("pose_net.conv1.weight"):
        std *= 0.1  # stem sees inputs in [0, 10+]
    return torch.from_numpy((std * g.standard_normal(shape)).astype(np.float32))


@torch.no_grad()
def fill_module(module: torch.nn.Module, prefix: str = "", head_gain: float = 6.0, smooth: bool = False) -> None:
    """Overwrite every tensor of module.state_dict() with fill_value(prefix+key)."""
    sd = module.state_dict()
    for key, t in sd.items():
        if key.endswith((".scales", ".inv_freqs")):
            continue  # rotary-embedding constants of the Sformer head stay as constructed
        v = fill_value(prefix + key, t.shape, head_gain, smooth)
        t.copy_(v.to(dtype=t.dtype, device=t.device))


def synthetic_meas(batch: int, T: int, N: int, kind: str = "transient", seed: int = 410) -> torch.Tensor:
    """(B,1,T,N,N) float32 in [0,1].  kind='transient': Poisson counts around a
    paraboloid arrival surface; kind='uniform': U[0,1)."""
    out = torch.empty(batch, 1, T, N, N, dtype=torch.float32)
    for b in range(batch):
        g = torch.Generator().manual_seed(seed + b)
        if kind == "uniform":
            out[b, 0] = torch.rand(T, N, N, generator=g)
            continue
        hh = torch.linspace(-1.0, 1.0, N).view(1, N, 1)
        ww = torch.linspace(-1.0, 1.0, N).view(1, 1, N)
        cx = 0.3 * torch.rand(1, generator=g).item() - 0.15
        cy = 0.3 * torch.rand(1, generator=g).item() - 0.15
        t0 = T * (0.30 + 0.25 * ((hh - cy) ** 2 + (ww - cx) ** 2))
        tt = torch.arange(T, dtype=torch.float32).view(T, 1, 1)
        sigma = max(3.0 * T / 128.0, 1.5)
        lam = 20.0 * torch.exp(-((tt - t0) ** 2) / (2.0 * sigma * sigma)) + 0.05
        cnt = torch.poisson(lam, generator=g)
        out[b, 0] = cnt / cnt.max().clamp_min(1.0)
    return out


def synthetic_vol(batch: int, T: int, N: int, seed: int = 1) -> torch.Tensor:
    """Binary occupancy target (B,1,T,N,N), about 2 % ones."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(batch, 1, T, N, N, generator=g) < 0.02).to(torch.float32)


def synthetic_joints(batch: int, hm: int, seed: int = 2, num_joints: int = 24) -> torch.Tensor:
    """(B, J, 3) joint targets in heat-map voxel units inside the central 3/4."""
    g = torch.Generator().manual_seed(seed)
    lo, hi = hm / 8.0, hm * 7.0 / 8.0
    return lo + (hi - lo) * torch.rand(batch, num_joints, 3, generator=g)


def synthetic_joints_box(batch: int, whd, seed: int = 2, num_joints: int = 24) -> torch.Tensor:
    """(B, J, 3) joint targets (x, y, z) inside the central 3/4 of a heat-map of W x H x D = `whd` voxels (the decode's
    axis order: x runs along W, z along D); synthetic_joints() is the cubic case."""
    g = torch.Generator().manual_seed(seed)
    ext = torch.tensor([float(v) for v in whd])
    return ext / 8.0 + (ext * 0.75) * torch.rand(batch, num_joints, 3, generator=g)


def mpjpe(pred: torch.Tensor, ref: torch.Tensor, num_joints: int = 24) -> float:
    """Mean per-joint position error in heat-map voxels between two (B,3J) decodes
    (definition: SURVEY.md 8d; x31.25 gives mm for the 2 m wall / 64 voxels)."""
    p = pred.reshape(pred.shape[0], num_joints, 3).double()
    r = ref.reshape(ref.shape[0], num_joints, 3).double()
    return (p - r).norm(dim=2).mean().item()


def synthetic_rgbe(frames: int, H: int, W: int, seed: int = 0):
    """Seeded (frames*H, W, 4) uint8 Radiance RGBE pixels shaped like a transient histogram image: most pixels
    dark (exponent 0 or small), a band of bright returns; runs of equal bytes so that RLE has work to do."""
    import numpy as np

    g = np.random.Generator(np.random.PCG64(seed))
    img = np.zeros((frames * H, W, 4), np.uint8)
    mant = g.integers(0, 256, size=(frames * H, W, 3), dtype=np.uint8)
    expo = g.integers(118, 131, size=(frames * H, W), dtype=np.uint8)
    dark = g.random((frames * H, W)) < 0.35
    expo[dark] = 0
    mant[dark] = 0
    # horizontal runs: repeat every 4th column block in a third of the rows
    rows = g.random(frames * H) < 0.33
    mant[rows, 1::2] = mant[rows, 0::2][:, : mant[rows, 1::2].shape[1]]
    expo[rows, 1::2] = expo[rows, 0::2][:, : expo[rows, 1::2].shape[1]]
    img[..., :3] = mant
    img[..., 3] = expo
    return img
