"""GPU parity of the auxiliary rows of SURVEY 8(f): VisibleNet (reference golden), the noise augmentation (blur vs
the oracle, Poisson draws by their statistics) and the fused weighted-MSE joint loss (row L2)."""
import numpy as np
import pytest
import torch

from hiddenpose_amd import hip_ops as ops
from oracle import nlospose_oracle as O
from util import rel_l2

pytestmark = pytest.mark.gpu


def test_visible_net_vs_reference_golden(golden):
    from hiddenpose_amd.feature_propagation import VisibleNet

    g = golden("visible_net.npz")
    y = VisibleNet(basedim=3)(torch.from_numpy(g["x"]).cuda())
    assert y.shape == g["y"].shape
    c = g["x"].shape[1]
    assert rel_l2(y[:, :c], g["y"][:, :c]) < 1e-6            # the four largest values, descending
    # their depth coordinates, exactly -- wherever the value is positive (zeros left by the ReLU tie, and torch.topk's
    # order among equal values is unspecified)
    pos = g["y"][:, :c] > 0
    assert pos.mean() > 0.9
    assert np.array_equal(y[:, c:].cpu().numpy()[pos], g["y"][:, c:][pos])


def test_weighted_mse_forward_backward():
    gen = torch.Generator().manual_seed(4)
    p = (torch.rand(3, 72, generator=gen) * 60).requires_grad_(True)
    t = torch.rand(3, 72, generator=gen) * 60
    w = (torch.rand(3, 72, generator=gen) > 0.2).float()
    ref = ((p.double() - t.double()) ** 2 * w.double()).sum() / 3
    ref.backward()
    pg = p.detach().cuda().requires_grad_(True)
    from hiddenpose_amd.criterion import weighted_mse_loss

    out = weighted_mse_loss(pg, t.cuda(), w.cuda(), True)
    (out * 1.5).backward()
    assert abs(out.item() / ref.item() - 1) < 1e-6
    assert rel_l2(pg.grad, 1.5 * p.grad) < 1e-6
    assert abs(weighted_mse_loss(pg.detach(), t.cuda(), w.cuda(), False).item() / (3 * ref.item()) - 1) < 1e-6


def test_noise_blur_matches_oracle_and_poisson_statistics():
    gen = torch.Generator().manual_seed(8)
    meas = torch.rand(300, 77, generator=gen) * 3.0
    blur = ops.add_noise(meas.cuda(), 10.61, seed=1, poisson=False)
    ref = O.blur_flat_replicate(meas.numpy(), 10.61)
    assert rel_l2(blur, ref) < 1e-6
    # Poisson: one draw per sample with the blurred value as its mean; deterministic in (seed, index)
    lam = torch.cat([torch.full((200000,), 0.7), torch.full((200000,), 6.0), torch.full((200000,), 40.0)]).cuda()
    import ctypes as C

    from hiddenpose_amd import _lib

    one = torch.ones(1, device="cuda")
    y = torch.empty_like(lam)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().hp_noise_blur_poisson(lam.data_ptr(), y.data_ptr(), lam.numel(), one.data_ptr(), 0, 1, 1234, st), "noise")
    y2 = torch.empty_like(lam)
    _lib.check(_lib.lib().hp_noise_blur_poisson(lam.data_ptr(), y2.data_ptr(), lam.numel(), one.data_ptr(), 0, 1, 1234, st), "noise")
    assert torch.equal(y, y2)
    assert torch.equal(y, y.round()) and float(y.min()) >= 0
    for k, m in enumerate((0.7, 6.0, 40.0)):
        s = y[k * 200000:(k + 1) * 200000].double()
        assert abs(s.mean().item() / m - 1) < 0.01, (m, s.mean().item())
        assert abs(s.var().item() / m - 1) < 0.03, (m, s.var().item())
    y3 = torch.empty_like(lam)
    _lib.check(_lib.lib().hp_noise_blur_poisson(lam.data_ptr(), y3.data_ptr(), lam.numel(), one.data_ptr(), 0, 1, 99, st), "noise")
    assert not torch.equal(y, y3)
