"""GPU parity of the memory-bound stages (GroupNorm+ReLU, pools, trilinear upsampling+concat, 1x1 conv,
leaky/add, normalize_feature, soft-argmax, BCE+Dice) against float64 CPU evaluations of the reference
operators, and of the FeatureExtraction / normalize / UNet3d stages against the reference goldens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hiddenpose_amd import hip_ops as ops
from hiddenpose_amd import testing as hpt
from oracle import nlospose_oracle as O
from util import rel_l2

pytestmark = pytest.mark.gpu


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("B,C,dims", [(2, 4, (8, 8, 8)), (1, 32, (2, 2, 2)), (2, 8, (3, 5, 6)), (1, 16, (4, 4, 4))])
def test_groupnorm_relu(B, C, dims):
    g = gen(C)
    z = torch.randn(B, C, *dims, generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    gy = torch.randn(B, C, *dims, generator=g)
    zd, gd, bd = (t.double().requires_grad_(True) for t in (z, gamma, beta))
    ref = F.relu(F.group_norm(zd, 4, gd, bd, 1e-5))
    (ref * gy.double()).sum().backward()
    zg, gg, bg = (t.cuda().requires_grad_(True) for t in (z, gamma, beta))
    y = ops._GroupNormRelu.apply(zg, gg, bg, 4, 1e-5)
    (y * gy.cuda()).sum().backward()
    assert rel_l2(y, ref) < 1e-6
    assert rel_l2(zg.grad, zd.grad) < 1e-5
    assert rel_l2(gg.grad, gd.grad) < 1e-5 and rel_l2(bg.grad, bd.grad) < 1e-5


def test_maxpool2_first_max_semantics():
    g = gen(1)
    x = torch.randn(2, 3, 4, 6, 8, generator=g)
    x[0, 0, :2, :2, :2] = 0.0  # a window of ties: gradient must go to its first element only
    xd = x.double().requires_grad_(True)
    ref = F.max_pool3d(xd, 2, 2)
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = ops.max_pool3d_2(xg)
    (y * gy.cuda()).sum().backward()
    assert rel_l2(y, ref) == 0.0
    assert rel_l2(xg.grad, xd.grad) < 1e-7


# (4,4,4), (8,2,16): the separable passes with shift decodes; (3,6,12): their general (division) decode; (2,5,6): separable
# forward, fused adjoint (W % 4 != 0); (1,2,3): both fused (odd W)
@pytest.mark.parametrize("dims", [(4, 4, 4), (1, 2, 3), (8, 2, 16), (3, 6, 12), (2, 5, 6)])
def test_upsample_cat(dims):
    g = gen(2)
    x1 = torch.randn(2, 3, *dims, generator=g)
    skip = torch.randn(2, 5, *(2 * d for d in dims), generator=g)
    a, b = x1.double().requires_grad_(True), skip.double().requires_grad_(True)
    ref = torch.cat([b, F.interpolate(a, scale_factor=2, mode="trilinear", align_corners=True)], dim=1)
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy.double()).sum().backward()
    ag, bg = x1.cuda().requires_grad_(True), skip.cuda().requires_grad_(True)
    y = ops.upsample_cat(ag, bg)
    (y * gy.cuda()).sum().backward()
    assert rel_l2(y, ref) < 1e-6
    assert rel_l2(ag.grad, a.grad) < 1e-6 and rel_l2(bg.grad, b.grad) == 0.0


def test_conv1x1_out():
    g = gen(3)
    x, w, b = torch.randn(2, 4, 4, 6, 8, generator=g), torch.randn(1, 4, 1, 1, 1, generator=g), torch.randn(1, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = F.conv3d(xd, wd, bd)
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy.double()).sum().backward()
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.conv3d(xg, wg, bg)
    (y * gy.cuda()).sum().backward()
    assert rel_l2(y, ref) < 1e-6 and rel_l2(xg.grad, xd.grad) < 1e-6
    assert rel_l2(wg.grad, wd.grad) < 1e-5
    # the bias gradient is a cancelling sum of 768 N(0,1) values: compare against the sum's natural scale
    assert abs(bg.grad.item() - bd.grad.item()) < 1e-5 * np.sqrt(gy.numel())


def test_leaky_add_and_add():
    g = gen(4)
    a, b = torch.randn(2, 1, 4, 4, 8, generator=g), torch.randn(2, 1, 4, 4, 8, generator=g)
    ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.leaky_relu(ad + bd, 0.2) + F.leaky_relu(ad, 0.2) + (ad + bd)
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy.double()).sum().backward()
    ag, bg = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.leaky_add(ag, bg, 0.2) + ops.leaky_add(ag, None, 0.2) + ops.add(ag, bg)
    (y * gy.cuda()).sum().backward()
    assert rel_l2(y, ref) < 1e-6 and rel_l2(ag.grad, ad.grad) < 1e-6 and rel_l2(bg.grad, bd.grad) < 1e-6


def test_normalize_feature_vs_golden(golden):
    g = golden("parts_io.npz")
    B, T, N = 2, 32, 32
    gy = (hpt.synthetic_meas(B, T, N, "uniform", seed=101) - 0.5).cuda()
    z = ((hpt.synthetic_meas(B, T, N, "uniform", seed=102) - 0.3) * 1e-4).cuda().requires_grad_(True)
    nz = ops.normalize_feature(z)
    (nz * gy).sum().backward()
    assert rel_l2(nz, g["norm_y"]) < 1e-6
    assert rel_l2(z.grad, g["norm_gx"]) < 1e-4
    assert float(nz.detach().min()) == 0.0 and abs(float(nz.detach().max()) - 10.0) < 1e-5


def test_softargmax_forward_backward():
    g = gen(5)
    h = torch.randn(2, 24, 4, 5, 6, generator=g) * 3
    hd = h.double().requires_grad_(True)
    ref = O.softmax_integral(hd, 24, 6, 5, 4)
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy.double()).sum().backward()
    hg = h.cuda().requires_grad_(True)
    y = ops.softmax_integral(hg, 24, 6, 5, 4)
    (y * gy.cuda()).sum().backward()
    assert rel_l2(y, ref) < 1e-6
    assert rel_l2(hg.grad, hd.grad) < 1e-5


def test_bce_dice_forward_backward():
    g = gen(6)
    x = torch.randn(2, 4096, generator=g) * 3
    t = (torch.rand(2, 4096, generator=g) < 0.05).float()
    xd = x.double().requires_grad_(True)
    ref = O.bce_dice_loss(xd, t.double())
    (ref * 1.7).backward()
    xg = x.cuda().requires_grad_(True)
    loss = ops.bce_dice(xg, t.cuda())
    (loss * 1.7).backward()
    assert abs(loss.item() - ref.item()) < 1e-6 * abs(ref.item())
    assert rel_l2(xg.grad, xd.grad) < 1e-5


def test_feature_extraction_and_unet_vs_golden(golden):
    from hiddenpose_amd.feature_extraction import FeatureExtraction
    from hiddenpose_amd.unet3d import UNet3d

    g = golden("parts_io.npz")
    B, T, N = 2, 32, 32
    gy = (hpt.synthetic_meas(B, T, N, "uniform", seed=101) - 0.5).cuda()
    fe = FeatureExtraction(basedim=1, in_channels=1, stride=1)
    hpt.fill_module(fe, "feature_extraction.")
    fe = fe.cuda()
    x = hpt.synthetic_meas(B, T, N, "transient", seed=410).cuda().requires_grad_(True)
    y = fe(x)
    (y * gy).sum().backward()
    assert rel_l2(y, g["fe_y"]) < 1e-5 and rel_l2(x.grad, g["fe_gx"]) < 1e-5
    for k, p in fe.named_parameters():
        assert rel_l2(p.grad, g["fe_g_" + k]) < 1e-4, k

    un = UNet3d(in_channels=1, n_channels=4)
    hpt.fill_module(un, "autoencoder.")
    un = un.cuda()
    u = (hpt.synthetic_meas(B, T, N, "uniform", seed=103) * 10.0).cuda().requires_grad_(True)
    uy = un(u)
    (uy * gy).sum().backward()
    assert rel_l2(uy, g["unet_y"]) < 1e-4 and rel_l2(u.grad, g["unet_gx"]) < 1e-3
    named = dict(un.named_parameters())
    for k in ["conv.double_conv.0.weight", "conv.double_conv.1.weight", "enc4.encoder.1.double_conv.3.weight",
              "dec1.conv.double_conv.0.weight", "dec4.conv.double_conv.4.bias", "out.conv.weight", "out.conv.bias"]:
        assert rel_l2(named[k].grad, g["unet_g_" + k]) < 2e-3, k


@pytest.mark.parametrize("dims", [(2, 32, 32), (1, 128, 128)])
def test_unet_bf16_kernels_equal_exact_kernels_on_rounded_operands(dims, capsys):
    """HP_PRECISION_BF16 in situ: every conv -> GroupNorm -> ReLU node of a U-Net (all 18 layer shapes of the network, forward
    + backward) with the bf16 matrix-core kernels (v_mfma_f32_4x4x4_16b_bf16) against the EXACT kernels fed operands rounded
    to bf16 beforehand (hip_ops "bf16emu"): same products, same fp32 accumulation, only the summation order differs.  The
    comparison is per node on IDENTICAL inputs (those of the fp32 chain): two bf16 chains cannot be compared end to end,
    because operand rounding is a step function -- a 1e-7 difference in a node's output flips the rounding of ~3e-5 of the
    next node's operands by a full bf16 ulp, and the difference grows as sqrt(delta * ulp) per node until it sits at the ulp
    level (measured: 6e-8 -> 1.7e-5 -> 8e-5 -> ... -> 6e-3 over the 18 nodes).  For the same reason (and because the ReLU
    mask of the backward is a step function of z) the data gradient is checked on the node's convolution alone, for one
    given output gradient."""
    from hiddenpose_amd.unet3d import UNet3d

    B, T, N = dims
    un = UNet3d(in_channels=1, n_channels=4)
    hpt.fill_module(un, "autoencoder.")
    un = un.cuda()
    u0 = (hpt.synthetic_meas(B, T, N, "uniform", seed=103) * 10.0).cuda()
    nodes = []
    orig = ops.conv3_gn_relu

    def record(x, w, b, gw, gb, groups, eps, out=None):
        nodes.append((x.detach(), w.detach(), b.detach(), gw.detach(), gb.detach(), groups, eps))
        return orig(x, w, b, gw, gb, groups, eps, out)

    ops.conv3_gn_relu = record
    try:
        with torch.no_grad():
            y_fp32 = un(u0)
            ops.conv3_gn_relu = orig
            prev = ops.set_dconv_precision("bf16")
            try:
                y_bf16 = un(u0)
            finally:
                ops.set_dconv_precision(prev)
    finally:
        ops.conv3_gn_relu = orig
    assert len(nodes) == 18
    worst = [0.0, 0.0]
    gen = torch.Generator().manual_seed(3)
    for x, w, b, gw, gb, groups, eps in nodes:
        gy = torch.randn(x.shape[0], w.shape[0], *x.shape[2:], generator=gen).cuda()
        out = {}
        for mode in ("bf16", "bf16emu"):
            prev = ops.set_dconv_precision(mode)
            try:
                with torch.no_grad():
                    y = orig(x, w, b, gw, gb, groups, eps)          # convolution + statistics epilogue + GroupNorm + ReLU
                xl = x.clone().requires_grad_(True)
                (ops._DConv3.apply(xl, w, b, False) * gy).sum().backward()   # the convolution's data gradient for ONE given gy
            finally:
                ops.set_dconv_precision(prev)
            out[mode] = (y, xl.grad)
        errs = (rel_l2(out["bf16"][0], out["bf16emu"][0]), rel_l2(out["bf16"][1], out["bf16emu"][1]))
        worst = [max(p, q) for p, q in zip(worst, errs)]
        assert errs[0] < 1e-6 and errs[1] < 2e-6, (tuple(x.shape), w.shape[0], errs)
    with capsys.disabled():
        print(f"\n[unet bf16] {dims}: 18 nodes, kernel vs rounded-operand emulation: y {worst[0]:.1e}, dx {worst[1]:.1e};  "
              f"whole U-Net, bf16 vs fp32 arithmetic: y {rel_l2(y_bf16, y_fp32):.2e}")
    assert 1e-4 < rel_l2(y_bf16, y_fp32) < 6e-2   # the mode really rounds; ~2.5e-3 per node, no blow-up


@pytest.mark.parametrize("dims", [(2, 3, 4, 6, 8), (1, 4, 8, 8, 16)])
def test_pool_and_skip_sums_both_gradients_in_the_pool_backward(dims):
    """hip_ops.pool_and_skip: (x, max_pool3d(x, 2)) as one node -- the gradient of x is pool_backward(g_pool) + g_skip, the
    skip gradient read in place from a larger (concatenation-shaped) gradient tensor; also with only one of the two used."""
    B, C, D, H, W = dims
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, C, D, H, W, generator=g)
    up = torch.randn(B, 2, D // 2, H // 2, W // 2, generator=g)          # a coarse tensor to concatenate behind the skip
    wcat = torch.randn(B, C + 2, D, H, W, generator=g)                   # weights of the loss on the concatenation
    wpool = torch.randn(B, C, D // 2, H // 2, W // 2, generator=g)
    xd = x.double().requires_grad_(True)
    cat = torch.cat([xd, F.interpolate(up.double(), scale_factor=2, mode="trilinear", align_corners=True)], dim=1)
    ((cat * wcat.double()).sum() + (F.max_pool3d(xd, 2) * wpool.double()).sum()).backward()
    xg = x.cuda().requires_grad_(True)
    skip, pooled = ops.pool_and_skip(xg)
    assert torch.equal(skip, xg.detach()) and rel_l2(pooled, F.max_pool3d(x.double(), 2)) == 0
    (((ops.upsample_cat(up.cuda(), skip)) * wcat.cuda()).sum() + (pooled * wpool.cuda()).sum()).backward()
    assert rel_l2(xg.grad, xd.grad) < 1e-6
    for use in ("skip", "pool"):   # one-sided uses
        xg2 = x.cuda().requires_grad_(True)
        s2, p2 = ops.pool_and_skip(xg2)
        ((s2 * wcat[:, :C].cuda()).sum() if use == "skip" else (p2 * wpool.cuda()).sum()).backward()
        xd2 = x.double().requires_grad_(True)
        ((xd2 * wcat[:, :C].double()).sum() if use == "skip" else (F.max_pool3d(xd2, 2) * wpool.double()).sum()).backward()
        assert rel_l2(xg2.grad, xd2.grad) < 1e-6
