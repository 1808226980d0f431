"""GPU parity of the thin-channel direct 3^3 convolutions (hp_dconv3_*) against float64 CPU
evaluations of the reference operators: ReplicationPad3d(1)+Conv3d (feature_extraction.py:147-158),
zero-padded conv3d (:167) and UNet3d's Conv3d(k3,p1) (unet3d.py:15-23)."""
import pytest
import torch
import torch.nn.functional as F

from hiddenpose_amd import hip_ops as ops
from util import rel_l2

pytestmark = pytest.mark.gpu

CASES = [(1, 1, True, (2, 9, 10, 37)), (1, 1, False, (2, 8, 8, 32)), (1, 4, False, (1, 8, 16, 32)),
         (4, 4, False, (2, 5, 9, 33)), (8, 4, False, (1, 8, 8, 16)), (4, 8, False, (1, 8, 8, 16)),
         (16, 32, False, (2, 4, 4, 4)), (64, 16, False, (1, 4, 6, 4)), (32, 8, False, (1, 4, 8, 8)),
         (32, 32, False, (1, 2, 2, 2)), (16, 4, False, (1, 8, 8, 8)), (1, 1, True, (1, 1, 3, 2)),
         # channel counts off the 4-wide MFMA blocks, x-tiles past 64, z ranges split over several workgroups
         (3, 5, False, (1, 4, 5, 70)), (6, 2, True, (2, 3, 9, 66)), (4, 4, False, (1, 40, 8, 16)), (1, 4, True, (1, 37, 4, 8))]


@pytest.mark.parametrize("cin,cout,rep,dims", CASES)
def test_dconv3(cin, cout, rep, dims):
    g = torch.Generator().manual_seed(cin * 100 + cout)
    B, D, H, W = dims
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = F.conv3d(F.pad(xd, (1,) * 6, mode="replicate"), wd, bd) if rep else F.conv3d(xd, wd, bd, padding=1)
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy.double()).sum().backward()
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops._DConv3.apply(xg, wg, bg, rep)
    (y * gy.cuda()).sum().backward()
    assert rel_l2(y, ref) < 2e-6
    assert rel_l2(xg.grad, xd.grad) < 2e-6
    assert rel_l2(wg.grad, wd.grad) < 1e-5
    assert rel_l2(bg.grad, bd.grad) < 1e-5
