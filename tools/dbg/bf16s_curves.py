import sys; sys.path.insert(0, '.')
import numpy as np, torch
from hiddenpose_amd import testing as hpt
from hiddenpose_amd.config import make_cfg
from hiddenpose_amd.NlosPose import NlosPose
from hiddenpose_amd.train_epoch import build_training, seed_everything, train_step
meas = hpt.synthetic_meas(2, 128, 128, "transient", seed=1).cuda()
vol = hpt.synthetic_vol(2, 128, 128, seed=2).cuda()
joints = hpt.synthetic_joints(2, 64, seed=3).cuda()
def curve(prec, dconv):
    seed_everything(410)
    cfg = make_cfg(128, 128, conv_precision=prec); cfg.MODEL.DCONV_PRECISION = dconv
    model = NlosPose(cfg).cuda().train()
    criterion, voxel_criterion, optimizer, _ = build_training(cfg, model)
    c = [float(train_step(model, criterion, voxel_criterion, optimizer, meas, vol, joints)[0]) for _ in range(20)]
    del model, optimizer; torch.cuda.empty_cache()
    return np.array(c)
tag = sys.argv[1]
a = curve("fp32", "fp32")
print(tag, "fp32 last %.0f best5 %.0f" % (a[-1], a[-5:].min()), flush=True)
for dconv in ("fp32", "bf16"):
    for rep in range(3):
        b = curve("bf16s", dconv)
        print(tag, "bf16s dconv=%s: last %.0f (%+.1f%%) best-of-last-5 %.0f (%+.1f%%) max dev %.1f%%" % (dconv, b[-1], 100 * (b[-1] / a[-1] - 1), b[-5:].min(), 100 * (b[-5:].min() / a[-5:].min() - 1), 100 * np.abs(b / a - 1).max()), flush=True)
