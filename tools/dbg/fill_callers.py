"""Who launches the ~150 tiny fill kernels of a training step?  (dev tool: torch.profiler with Python stacks)
    python tools/dbg/fill_callers.py"""
import sys; sys.path.insert(0, '.')
import collections
import torch
from torch.profiler import profile, ProfilerActivity
from hiddenpose_amd import testing as hpt
from hiddenpose_amd.config import make_cfg
from hiddenpose_amd.NlosPose import NlosPose
from hiddenpose_amd.train_epoch import build_training, train_step

T, N, B = 128, 128, 2
dev = torch.device("cuda", 0)
cfg = make_cfg(T, N, device=0, conv_precision="fp32")
model = NlosPose(cfg).to(dev).train()
criterion, voxel_criterion, optimizer, _ = build_training(cfg, model)
meas = hpt.synthetic_meas(B, T, N, "transient", seed=410).to(dev)
vol = hpt.synthetic_vol(B, T, N, seed=1).to(dev)
joints = hpt.synthetic_joints(B, T // 2, seed=2).to(dev)
for _ in range(2):
    train_step(model, criterion, voxel_criterion, optimizer, meas, vol, joints, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    train_step(model, criterion, voxel_criterion, optimizer, meas, vol, joints, None)
    torch.cuda.synchronize()
cnt = collections.Counter()
names = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::zeros_like', 'aten::ones_like', 'aten::full', 'aten::new_zeros'):
        names[ev.name] += 1
        st = [s for s in (ev.stack or []) if 'hiddenpose_amd' in s or 'torch/optim' in s or 'autograd' in s][:3]
        cnt[(ev.name, tuple(st))] += 1
print(names)
for (k, st), v in cnt.most_common(25):
    print(v, k, ' <- '.join(st)[:260])
