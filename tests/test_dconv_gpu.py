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
         (3, 5, False, (1, 4, 5, 70)), (6, 2, True, (2, 3, 9, 66)), (4, 4, False, (1, 40, 8, 16)), (1, 4, True, (1, 37, 4, 8)),
         # single-channel stencil kernel: 16-byte row loads + lane exchange (W % 4 == 0), two x blocks, z chunks
         (1, 1, True, (1, 6, 9, 64)), (1, 1, True, (2, 70, 5, 260)), (1, 1, False, (1, 11, 6, 264)), (1, 1, True, (1, 3, 4, 8))]


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


BF16_CASES = [c for c in CASES if not (c[0] == 1 and c[1] == 1)] + [(8, 8, False, (1, 9, 8, 70)), (12, 4, True, (1, 5, 9, 16)),
                                                                  (20, 8, False, (2, 6, 4, 8)), (4, 4, True, (1, 6, 9, 68))]


@pytest.mark.parametrize("cin,cout,rep,dims", BF16_CASES)
def test_dconv3_bf16_operands(cin, cout, rep, dims):
    """HP_PRECISION_BF16 (hip_ops.set_dconv_precision("bf16"), v_mfma_f32_4x4x4_16b_bf16): with x, w and the incoming
    gradient ON the bf16 grid every product is exact, so forward and data gradient equal the float64 reference up to fp32
    accumulation order -- the bars of the exact kernel; with arbitrary fp32 operands the error is the operand rounding
    (2^-9 relative per operand).  The weight gradient takes the bf16 kernel where W % 4 == 0 and cin > 1 (K = four consecutive
    voxels per instruction; bias gradient summed exactly) and the exact kernel elsewhere."""
    g = torch.Generator().manual_seed(cin * 100 + cout + 1)
    B, D, H, W = dims
    grid = lambda t: t.bfloat16().float()
    x = grid(torch.randn(B, cin, D, H, W, generator=g))
    w = grid(torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.2)
    b = torch.randn(cout, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = F.conv3d(F.pad(xd, (1,) * 6, mode="replicate"), wd, bd) if rep else F.conv3d(xd, wd, bd, padding=1)
    gy = grid(torch.randn(ref.shape, generator=g))
    (ref * gy.double()).sum().backward()
    prev = ops.set_dconv_precision("bf16")
    try:
        xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
        y = ops._DConv3.apply(xg, wg, bg, rep)
        (y * gy.cuda()).sum().backward()
        assert rel_l2(y, ref) < 2e-6
        assert rel_l2(xg.grad, xd.grad) < 2e-6
        assert rel_l2(wg.grad, wd.grad) < 1e-5
        assert rel_l2(bg.grad, bd.grad) < 1e-5
        # arbitrary fp32 operands: bounded by the operand rounding
        x2 = torch.randn(B, cin, D, H, W, generator=g)
        w2 = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.2
        ref2 = F.conv3d(F.pad(x2.double(), (1,) * 6, mode="replicate"), w2.double(), b.double()) if rep else \
            F.conv3d(x2.double(), w2.double(), b.double(), padding=1)
        y2 = ops._DConv3.apply(x2.cuda(), w2.cuda(), b.cuda(), rep)
        assert 1e-5 < rel_l2(y2, ref2) < 6e-3
        # ... and equals the exact kernel on operands rounded beforehand (the "bf16emu" checker of test_stages_gpu.py)
        gy2 = torch.randn(ref2.shape, generator=g).cuda()
        grads = {}
        for mode in ("bf16", "bf16emu"):
            ops.set_dconv_precision(mode)
            xl, wl = x2.cuda().requires_grad_(True), w2.cuda().requires_grad_(True)
            yy = ops._DConv3.apply(xl, wl, b.cuda(), rep)
            (yy * gy2).sum().backward()
            grads[mode] = (yy.detach(), xl.grad, wl.grad)
        for a, e, bar in zip(grads["bf16"], grads["bf16emu"], (2e-6, 2e-6, 1e-5)):
            assert rel_l2(a, e) < bar
    finally:
        ops.set_dconv_precision(prev)
    # and the switch is really off again: exact result on off-grid operands
    y3 = ops._DConv3.apply(x2.cuda(), w2.cuda(), b.cuda(), rep)
    assert rel_l2(y3, ref2) < 2e-6


def test_conv_groupnorm_relu_bf16_operands():
    """The DoubleConv half in bf16 arithmetic: statistics come from the bf16 kernel's epilogue, the backward's data gradient
    runs on the same matrix-core path."""
    g = torch.Generator().manual_seed(5)
    grid = lambda t: t.bfloat16().float()
    cin, cout, (B, D, H, W) = 8, 4, (2, 7, 8, 66)
    x = grid(torch.randn(B, cin, D, H, W, generator=g))
    w = grid(torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.2)
    b = torch.randn(cout, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
    xd, wd, bd, gd, btd = (t.double().requires_grad_(True) for t in (x, w, b, gamma, beta))
    ref = F.relu(F.group_norm(F.conv3d(xd, wd, bd, padding=1), 4, gd, btd, 1e-5))
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy.double()).sum().backward()
    prev = ops.set_dconv_precision("bf16")
    try:
        xg, wg, bg, gg, btg = (t.cuda().requires_grad_(True) for t in (x, w, b, gamma, beta))
        y = ops.conv3_gn_relu(xg, wg, bg, gg, btg, 4, 1e-5)
        (y * gy.cuda()).sum().backward()
    finally:
        ops.set_dconv_precision(prev)
    assert rel_l2(y, ref) < 3e-6
    # dz (the GroupNorm backward's output) is not on the bf16 grid: the data gradient carries its rounding
    assert rel_l2(xg.grad, xd.grad) < 6e-3
    assert rel_l2(wg.grad, wd.grad) < 2e-5
    assert rel_l2(gg.grad, gd.grad) < 2e-5 and rel_l2(btg.grad, btd.grad) < 2e-5


@pytest.mark.parametrize("cin,cout,rep,dims", [(1, 1, True, (2, 9, 10, 37)), (1, 1, False, (1, 12, 16, 64)), (4, 4, True, (1, 6, 9, 68))])
def test_dconv3_fused_residual_leaky(cin, cout, rep, dims):
    """y = leaky(conv(x) + b + res, 0.2): ResConv3D's second half (feature_extraction.py:228-256) in one kernel."""
    g = torch.Generator().manual_seed(7 + cin)
    B, D, H, W = dims
    x = torch.randn(B, cin, D, H, W, generator=g)
    r = torch.randn(B, cout, D, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    xd, wd, bd, rd = (t.double().requires_grad_(True) for t in (x, w, b, r))
    conv = F.conv3d(F.pad(xd, (1,) * 6, mode="replicate"), wd, bd) if rep else F.conv3d(xd, wd, bd, padding=1)
    ref = F.leaky_relu(conv + rd, 0.2)
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy.double()).sum().backward()
    xg, wg, bg, rg = (t.cuda().requires_grad_(True) for t in (x, w, b, r))
    y = ops._DConv3.apply(xg, wg, bg, rep, rg, 0.2)
    (y * gy.cuda()).sum().backward()
    assert rel_l2(y, ref) < 2e-6
    assert rel_l2(xg.grad, xd.grad) < 2e-6 and rel_l2(rg.grad, rd.grad) < 2e-6
    assert rel_l2(wg.grad, wd.grad) < 1e-5 and rel_l2(bg.grad, bd.grad) < 1e-5


@pytest.mark.parametrize("cin,cout,dims", [(1, 4, (2, 8, 16, 32)), (4, 4, (1, 20, 9, 66)), (8, 4, (2, 5, 8, 16)), (16, 32, (1, 4, 4, 4)),
                                           (64, 16, (1, 4, 6, 4)), (4, 8, (1, 37, 8, 8))])
def test_conv_groupnorm_relu_one_node(cin, cout, dims):
    """relu(GroupNorm(conv(x))) (DoubleConv half, unet3d.py:14-24) with the statistics taken in the convolution's
    epilogue and the backward's ReLU mask rebuilt from z."""
    g = torch.Generator().manual_seed(11 + cin + cout)
    B, D, H, W = dims
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
    xd, wd, bd, gd, btd = (t.double().requires_grad_(True) for t in (x, w, b, gamma, beta))
    ref = F.relu(F.group_norm(F.conv3d(xd, wd, bd, padding=1), 4, gd, btd, 1e-5))
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy.double()).sum().backward()
    xg, wg, bg, gg, btg = (t.cuda().requires_grad_(True) for t in (x, w, b, gamma, beta))
    y = ops.conv3_gn_relu(xg, wg, bg, gg, btg, 4, 1e-5)
    (y * gy.cuda()).sum().backward()
    assert rel_l2(y, ref) < 3e-6
    assert rel_l2(xg.grad, xd.grad) < 2e-5
    assert rel_l2(wg.grad, wd.grad) < 2e-5
    # a bias in front of GroupNorm with one channel per group has NO effect (exact gradient 0): absolute bar, scaled by
    # the mass of the terms that cancel
    scale = float(gy.abs().sum()) * float(gamma.abs().max())
    assert float((bg.grad.cpu().double() - bd.grad).abs().max()) < 1e-6 * scale
    assert rel_l2(gg.grad, gd.grad) < 2e-5 and rel_l2(btg.grad, btd.grad) < 2e-5


@pytest.mark.gpu
def test_thin_channel_weight_gradients_on_the_second_stream_equal_the_serial_ones():
    """Round 4: the thin-channel weight gradients (FeatureExtraction / U-Net) run on the second stream like the regressor's
    (hip_ops._dconv3_grads): leaf parameters whose .grad autograd simply adopts; operands held until the main stream has waited
    for the kernel; joined when backward ends.  Same kernels, same order of summation: the gradients are bit-equal to the
    one-stream run -- also with a parameter that already holds a gradient (accumulation: that layer stays on the main stream)."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 4, 12, 16, 64, generator=g).cuda()
    ws = [(torch.randn(8, 4, 3, 3, 3, generator=g) * 0.2).cuda(), (torch.randn(4, 8, 3, 3, 3, generator=g) * 0.2).cuda()]
    bs = [torch.randn(8, generator=g).cuda(), torch.randn(4, generator=g).cuda()]
    gam, bet = (1 + 0.3 * torch.randn(8, generator=g)).cuda(), (0.3 * torch.randn(8, generator=g)).cuda()
    gy = torch.randn(2, 4, 12, 16, 64, generator=g).cuda()

    def run(on, pre_grad):
        prev = ops._DCONV_WGRAD_STREAM
        ops._DCONV_WGRAD_STREAM = on
        try:
            ps = [t.clone().requires_grad_(True) for t in (ws[0], bs[0], gam, bet, ws[1], bs[1])]
            if pre_grad:
                ps[4].grad = torch.full_like(ps[4], 0.5)
            xi = x.clone().requires_grad_(True)
            for _ in range(2):          # twice: the hold queue and the end-of-backward join are exercised across passes
                y = ops.conv3d_reppad(ops.conv3_gn_relu(xi, ps[0], ps[1], ps[2], ps[3], 4, 1e-5), ps[4], ps[5], residual=xi, slope=0.2)
                (y * gy).sum().backward()
            torch.cuda.synchronize()
            return [xi.grad.clone()] + [p.grad.clone() for p in ps]
        finally:
            ops._DCONV_WGRAD_STREAM = prev

    for pre in (False, True):
        n0 = ops._side_dconv_calls[0]
        a = run(True, pre)
        n1 = ops._side_dconv_calls[0]
        b = run(False, pre)
        # first pass: both layers (or the one without an accumulated gradient) take the second stream; second pass: none does
        assert n1 - n0 == (1 if pre else 2) and ops._side_dconv_calls[0] == n1
        for u, v in zip(a, b):
            assert torch.equal(u, v)


@pytest.mark.gpu
def test_a_thin_channel_weight_used_twice_stays_on_the_main_stream():
    """A weight that appears twice in one graph has its two gradients SUMMED by autograd on the main stream: neither may be written
    on the second stream (hip_ops._count_use / _shared_use, as for the regressor's convolutions).  Bit-equal to the one-stream run,
    over two passes (the use counts return to zero)."""
    g = torch.Generator().manual_seed(78)
    x = torch.randn(1, 4, 10, 16, 64, generator=g).cuda()
    w0 = (torch.randn(4, 4, 3, 3, 3, generator=g) * 0.2).cuda()
    b0 = torch.randn(4, generator=g).cuda()
    gy = torch.randn(1, 4, 10, 16, 64, generator=g).cuda()

    def run(on):
        prev = ops._DCONV_WGRAD_STREAM
        ops._DCONV_WGRAD_STREAM = on
        try:
            out = []
            for _ in range(2):
                w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
                xi = x.clone().requires_grad_(True)
                y = ops.conv3d_reppad(ops.conv3d_reppad(xi, w, b, slope=0.2), w, b, residual=xi)
                (y * gy).sum().backward()
                torch.cuda.synchronize()
                assert not getattr(w, "_hp_uses", {}) and not getattr(w, "_hp_shared", False)   # every use consumed
                out += [xi.grad.clone(), w.grad.clone(), b.grad.clone()]
            return out
        finally:
            ops._DCONV_WGRAD_STREAM = prev

    n0 = ops._side_dconv_calls[0]
    a = run(True)
    assert ops._side_dconv_calls[0] == n0       # the shared weight never took the second stream
    for u, v in zip(a, run(False)):
        assert torch.equal(u, v)
