"""Drop-in for utils/nlos_pose_dataloader_noise.py: the same dataset as nlos_pose_dataloader.NlosPoseDataset plus the
reference's (unused by train.py / test.py) noise augmentation `addnoise_dataset` (:167-172): the measurement image
is flattened to one long column, blurred along it with a Gaussian of sigma 10.61 (= 25 / 2.355 time bins) with a
replicate border, then replaced by a Poisson draw with that mean.  Both steps run in one HIP kernel
(hp_noise_blur_poisson); the Poisson generator is counter based (sample i is a function of (seed, i) only), so a
given seed gives the same noisy measurement on any launch geometry.  Bit parity with numpy.random.poisson is not
defined (different generators); the blur is deterministic and checked against the oracle."""
from __future__ import annotations

import torch

from . import hip_ops
from .nlos_pose_dataloader import NlosPoseDataset as _Base


class NlosPoseDataset(_Base):
    noise_sigma = 10.61

    def __init__(self, cfg, datapath=None, device=None, noise_seed: int = 0, add_noise: bool = True):
        super().__init__(cfg, datapath, device=device)
        self.noise_seed, self.add_noise = int(noise_seed), bool(add_noise)

    def addnoise_dataset(self, meas: torch.Tensor, index: int = 0) -> torch.Tensor:
        """(a, b) device tensor -> same shape, blurred along the flattened (a b) axis and Poisson sampled."""
        return hip_ops.add_noise(meas, self.noise_sigma, seed=(self.noise_seed << 32) ^ int(index), poisson=True)
