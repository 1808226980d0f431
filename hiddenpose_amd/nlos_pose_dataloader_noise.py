"""Drop-in for utils/nlos_pose_dataloader_noise.py: nlos_pose_dataloader.NlosPoseDataset with the reference's noise
augmentation applied where the reference applies it (:92-94, :101-103, :114-116):

    imread -> BGR2GRAY of the RAW image (the first "/ max" is commented out at :92) -> addnoise_dataset -> / max
           -> '(t h) w -> t h w'[:512] -> time pairs -> DAWNSAMPLE_CNT box rounds                       (:106-117)

`addnoise_dataset` (:167-172) flattens the gray image to one long column, blurs along it with a Gaussian of sigma 10.61
(= 25 / 2.355) with a replicate border and replaces every value by a Poisson draw with that mean.  All of it runs on the
device: hp_ingest_rgbe_to_gray -> hp_noise_blur_poisson -> hp_ingest_image_to_meas (csrc/ingest_kernels.hip,
misc_kernels.hip).  The Poisson generator is counter based (sample i is a function of (seed, i) only), so a given
(noise_seed, index) gives the same noisy measurement on any launch geometry; bit parity with numpy.random.poisson is not
defined (different generators), everything around the draw is checked against the oracle (tests/test_ingest.py).

One deviation, deliberate: the reference's addnoise_dataset returns the FLATTENED array (cv2 hands back an (a*b, 1)
image), so its own '(t h) w' rearrange at :106 folds a one-pixel-wide image and the box rounds end in an empty width
axis -- the class cannot feed a network as written, which is why train.py imports the noise-free dataset.  Here the noisy
image gets its (a, b) shape back before the rearrange, which is what the surrounding code evidently intends.

`add_noise=False` is the noise-free dataset, bit for bit (it takes the base class's fused ingest)."""
from __future__ import annotations

import torch

from . import _lib, hip_ops
from .nlos_pose_dataloader import FRAMES, KEEP_FRAMES
from .nlos_pose_dataloader import NlosPoseDataset as _Base


def rgbe_to_noisy_meas(rgbe: torch.Tensor, downsample_cnt: int, seed: int, sigma: float = 10.61, poisson: bool = True,
                       frames: int = FRAMES, keep: int = KEEP_FRAMES) -> torch.Tensor:
    """(frames*H, W, 4) uint8 device tensor -> (keep/2^(cnt+1), H/2^cnt, W/2^cnt) fp32 through the noise variant of
    __getitem__ (:86-118).  `poisson=False` stops addnoise_dataset after the blur (deterministic; what the parity tests pin)."""
    if not rgbe.is_cuda:
        raise _lib.HiddenPoseHipError("rgbe_to_noisy_meas needs a HIP device tensor (no CPU path)")
    rows, W, four = rgbe.shape
    assert four == 4 and rgbe.dtype == torch.uint8 and rows % frames == 0
    H = rows // frames
    div = 1 << downsample_cnt
    L = _lib.lib()
    rgbe = rgbe.contiguous()
    st = torch.cuda.current_stream(rgbe.device).cuda_stream
    gray = torch.empty(rows, W, dtype=torch.float32, device=rgbe.device)
    maxima = torch.empty(2, dtype=torch.float32, device=rgbe.device)
    _lib.check(L.hp_ingest_rgbe_to_gray(rgbe.data_ptr(), rows * W, gray.data_ptr(), maxima.data_ptr(), st), "hp_ingest_rgbe_to_gray")
    if abs(float(maxima[0])) < 1e-10:  # :88 (one 4-byte read-back per sample)
        raise ValueError("wrong Meas File!")
    noisy = hip_ops.add_noise(gray, sigma, seed=seed, poisson=poisson)          # addnoise_dataset, same (a, b) shape
    meas = torch.empty(keep // (2 * div), H // div, W // div, dtype=torch.float32, device=rgbe.device)
    _lib.check(L.hp_ingest_image_to_meas(noisy.data_ptr(), frames, H, W, keep, downsample_cnt, 1 if poisson else 0, meas.data_ptr(),
                                         maxima[1:].data_ptr(), st), "hp_ingest_image_to_meas")
    return meas


class NlosPoseDataset(_Base):
    noise_sigma = 10.61

    def __init__(self, cfg, datapath=None, device=None, noise_seed: int = 0, add_noise: bool = True, poisson: bool = True):
        super().__init__(cfg, datapath, device=device)
        self.noise_seed, self.add_noise, self.poisson = int(noise_seed), bool(add_noise), bool(poisson)

    def addnoise_dataset(self, meas: torch.Tensor, index: int = 0) -> torch.Tensor:
        """(a, b) device tensor -> same shape, blurred along the flattened (a b) axis and Poisson sampled (:167-172)."""
        return hip_ops.add_noise(meas, self.noise_sigma, seed=self._seed(index), poisson=self.poisson)

    def _seed(self, index: int) -> int:
        return (self.noise_seed << 32) ^ int(index)

    def meas_to_device(self, rgbe: torch.Tensor, index: int = 0) -> torch.Tensor:
        if not self.add_noise:
            return super().meas_to_device(rgbe, index)
        return rgbe_to_noisy_meas(rgbe, self.downsample_cnt, self._seed(index), self.noise_sigma, self.poisson)
