import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from hiddenpose_amd import hip_ops as ops, testing as hpt
from hiddenpose_amd.unet3d import UNet3d
from util import rel_l2
B, T, N = 2, 32, 32
un = UNet3d(1, 4); hpt.fill_module(un, "autoencoder."); un = un.cuda()
u0 = (hpt.synthetic_meas(B, T, N, "uniform", seed=103) * 10.0).cuda()
def run(mode, grad):
    p = ops.set_dconv_precision(mode)
    try:
        if grad:
            u = u0.clone().requires_grad_(True)
            return un(u).detach()
        with torch.no_grad():
            return un(u0)
    finally:
        ops.set_dconv_precision(p)
for grad in (False, True):
    y = {m: run(m, grad) for m in ("fp32", "bf16", "bf16emu")}
    print("grad", grad, "bf16 vs emu %.2e  bf16 vs fp32 %.2e  emu vs fp32 %.2e" % (rel_l2(y["bf16"], y["bf16emu"]), rel_l2(y["bf16"], y["fp32"]), rel_l2(y["bf16emu"], y["fp32"])))
# chain with trace: record every node's output in both modes
orig = ops.conv3_gn_relu
trace = {}
def rec(mode):
    def f(x, w, b, gw, gb, groups, eps, out=None):
        y = orig(x, w, b, gw, gb, groups, eps, out)
        trace.setdefault(mode, []).append((x.clone(), y.clone()))
        return y
    return f
for m in ("bf16", "bf16emu"):
    ops.conv3_gn_relu = rec(m)
    run(m, False)
ops.conv3_gn_relu = orig
for i, ((xa, ya), (xb, yb)) in enumerate(zip(trace["bf16"], trace["bf16emu"])):
    nflip = (xa.bfloat16() != xb.bfloat16()).float().mean().item()
    print(i, tuple(xa.shape), "in diff %.2e out diff %.2e  fraction of inputs rounding differently %.2e" % (rel_l2(xa, xb), rel_l2(ya, yb), nflip))
