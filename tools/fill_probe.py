import sys, torch
sys.path.insert(0, ".")
from hiddenpose_amd import testing as hpt
from hiddenpose_amd.config import make_cfg
from hiddenpose_amd.NlosPose import NlosPose
from hiddenpose_amd.train_epoch import build_training, seed_everything, train_step
T, N, B = 128, 128, 2
cfg = make_cfg(T, N, device=0); seed_everything(410)
model = NlosPose(cfg).cuda().train()
crit, vcrit, opt, _ = build_training(cfg, model)
meas = hpt.synthetic_meas(B, T, N, "transient", seed=410).cuda(); vol = hpt.synthetic_vol(B, T, N, seed=1).cuda(); joints = hpt.synthetic_joints(B, T // 2, seed=2).cuda()
for _ in range(2): train_step(model, crit, vcrit, opt, meas, vol, joints)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    train_step(model, crit, vcrit, opt, meas, vol, joints); torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for e in prof.events():
    if True:
        st = [s for s in (e.stack or []) if "hiddenpose_amd" in s or "train" in s or "autograd" in s][:2]
        cnt[e.name] += 1
for k, v in cnt.most_common(25): print(v, k)
