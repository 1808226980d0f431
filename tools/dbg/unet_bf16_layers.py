"""Diagnostic: every conv -> GroupNorm -> ReLU node of a U-Net forward, bf16 kernel vs exact kernel on rounded operands."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from hiddenpose_amd import hip_ops as ops, testing as hpt
from hiddenpose_amd.unet3d import UNet3d
from util import rel_l2
B, T, N = 2, 32, 32
un = UNet3d(1, 4); hpt.fill_module(un, "autoencoder."); un = un.cuda()
u0 = (hpt.synthetic_meas(B, T, N, "uniform", seed=103) * 10.0).cuda()
orig = ops.conv3_gn_relu
def both(x, w, b, gw, gb, groups, eps, out=None):
    out = {}
    for m in ("fp32", "bf16", "bf16emu"):
        p = ops.set_dconv_precision(m)
        out[m] = orig(x, w, b, gw, gb, groups, eps)
        ops.set_dconv_precision(p)
    # raw convolution too
    raw = {}
    for m in ("bf16", "bf16emu"):
        p = ops.set_dconv_precision(m)
        raw[m] = ops._DConv3.apply(x, w, b, False)
        ops.set_dconv_precision(p)
    xr, wr = x.bfloat16().float(), w.bfloat16().float()
    ref = torch.nn.functional.conv3d(xr.double().cpu(), wr.double().cpu(), b.double().cpu(), padding=1)
    print(f"{tuple(x.shape)} -> {w.shape[0]}: gn bf16 vs emu {rel_l2(out['bf16'], out['bf16emu']):.2e}  bf16 vs fp32 {rel_l2(out['bf16'], out['fp32']):.2e} | "
          f"raw conv bf16 vs emu {rel_l2(raw['bf16'], raw['bf16emu']):.2e}  bf16 vs f64(rounded) {rel_l2(raw['bf16'], ref):.2e}  emu vs f64 {rel_l2(raw['bf16emu'], ref):.2e}"
          f"  groups {groups}", flush=True)
    return out["fp32"]
ops.conv3_gn_relu = both
import hiddenpose_amd.unet3d as U
with torch.no_grad():
    un(u0)
