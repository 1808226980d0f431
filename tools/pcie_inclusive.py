"""PCIe-inclusive rates for DESIGN.md (bench.py's `value` keeps inputs resident in HBM):
(a) ingest of one dataset sample starting from the expanded RGBE bytes in pinned host memory,
(b) the headline train step with meas / vol / joints copied from pinned host memory every step."""
import sys
import time

sys.path.insert(0, ".")
import torch

from hiddenpose_amd import testing as hpt
from hiddenpose_amd.config import make_cfg
from hiddenpose_amd.NlosPose import NlosPose
from hiddenpose_amd.nlos_pose_dataloader import box_pyramid, rgbe_to_meas
from hiddenpose_amd.train_epoch import build_training, seed_everything, train_step

torch.cuda.set_device(0)
g = torch.Generator().manual_seed(1)
rg = torch.randint(0, 256, (600 * 256, 256, 4), dtype=torch.uint8, generator=g)
rg[..., 3] = rg[..., 3] % 12 + 120
rg = rg.pin_memory()
vol = (torch.rand(256, 256, 256, generator=g) < 0.02).float().pin_memory()


def ingest():
    return rgbe_to_meas(rg.cuda(non_blocking=True), 1), box_pyramid(vol.cuda(non_blocking=True), 1)


for _ in range(3):
    ingest()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ingest()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"ingest incl. host->device copy of 157 MB RGBE + 67 MB volume: {dt*1e3:.2f} ms/sample ({1/dt:.0f} samples/s)")

seed_everything(410)
T, N, B = 512, 128, 4
cfg = make_cfg(T, N)
model = NlosPose(cfg).cuda().train()
c, v, o, _ = build_training(cfg, model)
meas = hpt.synthetic_meas(B, T, N, "transient", seed=410).pin_memory()
vl = hpt.synthetic_vol(B, T, N, seed=1).pin_memory()
jt = hpt.synthetic_joints(B, T // 2, seed=2).pin_memory()


def step():
    return train_step(model, c, v, o, meas.cuda(non_blocking=True), vl.cuda(non_blocking=True), jt.cuda(non_blocking=True))


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"train step incl. host->device copy of the batch (2 x 134 MB): {dt*1e3:.1f} ms/step ({B/dt:.2f} samples/s)")
