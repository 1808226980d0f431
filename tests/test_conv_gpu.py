"""GPU parity of the pose-regressor building blocks (C ABI: hp_conv3d_*, hp_bn_*, hp_maxpool3d_*)
against float64 CPU evaluations of the same reference operators (nn.Conv3d / ConvTranspose3d /
BatchNorm3d / MaxPool3d as used by models/posenet3d_50.py) and the reference goldens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hiddenpose_amd import hip_ops as ops
from hiddenpose_amd import testing as hpt
from util import rel_l2

pytestmark = pytest.mark.gpu


def cl(x):  # (B,C,D,H,W) -> channels-last (B,D,H,W,C)
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


CASES = [
    # name, Cin, Cout, k, stride, pad, transposed, (B, D, H, W)
    ("k1", 64, 64, 1, 1, 0, False, (2, 4, 6, 8)),
    ("k1_wide", 256, 96, 1, 1, 0, False, (1, 4, 4, 8)),
    ("k3", 64, 64, 3, 1, 1, False, (2, 6, 4, 8)),
    ("k3_s2", 128, 160, 3, 2, 1, False, (1, 8, 4, 6)),
    ("k1_s2", 64, 256, 1, 2, 0, False, (2, 4, 4, 8)),
    ("deconv", 64, 32, 4, 2, 1, True, (2, 3, 4, 2)),
    ("stem", 1, 64, 7, 1, 3, False, (1, 10, 12, 9)),
    ("head", 256, 24, 1, 1, 0, False, (1, 4, 4, 4)),
    # long and wide: >= 131072 voxels, 256 output channels (forward; data gradient)
    ("big_fwd256", 32, 256, 1, 1, 0, False, (1, 32, 64, 64)),
    ("big_dgrad256", 256, 32, 1, 1, 0, False, (1, 33, 64, 64)),
    # >= 65536 voxels, 64 channels on one side and 256 on the other: the one-block-per-split weight gradient
    ("k1_thin_n", 64, 256, 1, 1, 0, False, (1, 16, 64, 65)),
    ("k1_thin_c", 256, 64, 1, 1, 0, False, (1, 16, 64, 65)),
]


def _bf16_grid(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3", "bf16x6"])
@pytest.mark.parametrize("name,cin,cout,k,s,p,tr,dims", CASES, ids=[c[0] for c in CASES])
def test_conv_forward_and_gradients(name, cin, cout, k, s, p, tr, dims, precision):
    """fp32: against the float64 operator.  bf16 (HP_PRECISION_BF16): operands that already lie on the bf16
    grid are not changed by the kernel's rounding and their products are exact in fp32, so the same float64
    operator is the expected value up to fp32 accumulation order -- any indexing slip in the bf16 fragment
    layout or the transposing LDS reads shows as an O(1) error.  bf16x3 / bf16x6 take arbitrary fp32
    operands: 2 planes keep 16 significant bits (tolerance 3e-5), 3 planes all 24 (held to the fp32 bar)."""
    import ctypes as C

    from hiddenpose_amd import _lib

    g = torch.Generator().manual_seed(sum(map(ord, name)))
    B, D, H, W = dims
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = torch.randn((cin, cout, k, k, k) if tr else (cout, cin, k, k, k), generator=g) / np.sqrt(cin * k ** 3 / s ** 3)
    if precision == "bf16":
        x, w = _bf16_grid(x), _bf16_grid(w)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv_transpose3d(xd, wd, stride=s, padding=p) if tr else F.conv3d(xd, wd, stride=s, padding=p)
    gy = torch.randn(ref.shape, generator=g)
    if precision == "bf16":
        gy = _bf16_grid(gy)
    (ref * gy.double()).sum().backward()

    L = _lib.lib()
    xc = cl(x).cuda()
    wc = w.cuda()
    prev = ops.set_conv_precision(precision)
    try:
        desc = ops._desc(xc, cout, k, s, p, tr)
    finally:
        ops.set_conv_precision(prev)
    assert desc.precision == {"fp32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3}[precision]
    tol_f, tol_g = (3e-5, 3e-5) if precision == "bf16x3" else (2e-6, 5e-6)
    st = ops._stream(xc)
    wf, _ = ops._pack(desc, wc, True, False)
    do, ho, wo = ops._out_dims(desc)
    assert (do, ho, wo) == tuple(ref.shape[2:])
    y = torch.empty(B, do, ho, wo, cout, device="cuda")
    stats = torch.empty(_lib.STATS_SLOTS * 2 * cout, dtype=torch.float64, device="cuda")
    _lib.check(L.hp_conv3d_forward(C.byref(desc), xc.data_ptr(), wf.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st), "fwd")
    stats = stats.view(_lib.STATS_SLOTS, 2 * cout).sum(0)      # the epilogue's partial vectors (HP_STATS_SLOTS)
    assert rel_l2(ncdhw(y), ref) < tol_f
    refcl = cl(ref.detach())
    assert rel_l2(stats[:cout], refcl.reshape(-1, cout).sum(0)) < 10 * tol_f
    assert rel_l2(stats[cout:], (refcl.reshape(-1, cout) ** 2).sum(0)) < 10 * tol_f
    dx, dw = ops._conv_grads(desc, xc, wc, cl(gy).cuda(), True)
    assert rel_l2(ncdhw(dx), xd.grad) < tol_g
    assert rel_l2(dw, wd.grad) < tol_g


def test_stem_data_gradient_interior_and_edge_patches():
    """The stem's data gradient on a volume with whole interior patches next to partial ones in every direction
    (4 x 4 x 8-voxel patches on 13 x 18 x 43 voxels, two samples, the launcher's own z split), all 343 taps random:
    against float64 autograd."""
    g = torch.Generator().manual_seed(78)
    B, D, H, W = 2, 13, 18, 43
    w = torch.randn(64, 1, 7, 7, 7, generator=g) / np.sqrt(343.0)
    gy = torch.randn(B, 64, D, H, W, generator=g)
    xd = torch.zeros(B, 1, D, H, W, dtype=torch.float64, requires_grad=True)
    (F.conv3d(xd, w.double(), padding=3) * gy.double()).sum().backward()
    xc = cl(torch.zeros(B, 1, D, H, W)).cuda()
    desc = ops._desc(xc, 64, 7, 1, 3, False)
    dx, _ = ops._conv_grads(desc, xc, w.cuda(), cl(gy).cuda(), True)
    assert rel_l2(ncdhw(dx), xd.grad) < 5e-6


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("zsplit", [1, 2, 0])
def test_stem_data_gradient_walks_patches_along_z(precision, zsplit, monkeypatch):
    """The stem's data gradient keeps its output patch as a ring of planes while a workgroup walks several 4-plane
    patches along z (HP_STEM_DGRAD_ZSPLIT forces runs of 6 / 3 patches on this small volume, 0 = the launcher's own choice;
    depth 22 ends in a partial patch, H and W are not tile multiples: voxel rows outside the volume are fetched with
    out-of-range buffer offsets, z slices outside with an empty descriptor; all 343 taps carry random weights, so the
    49th tap row that rides in the spare columns of chunks 0..6 is checked like any other).  fp32: the exact kernel; bf16: the patch GEMM on the bf16 matrix cores with
    operands on the bf16 grid (exact products).  Also the stem forward (with its BatchNorm statistics) and the stem weight
    gradient of either mode on the same case."""
    if zsplit:
        monkeypatch.setenv("HP_STEM_DGRAD_ZSPLIT", str(zsplit))
    else:
        monkeypatch.delenv("HP_STEM_DGRAD_ZSPLIT", raising=False)   # the size heuristic: one patch per workgroup on this volume
    g = torch.Generator().manual_seed(77)
    B, D, H, W = 2, 22, 7, 11
    x = torch.randn(B, 1, D, H, W, generator=g)
    w = torch.randn(64, 1, 7, 7, 7, generator=g) / np.sqrt(343.0)
    gy = torch.randn(B, 64, D, H, W, generator=g)
    if precision == "bf16":
        x, w, gy = _bf16_grid(x), _bf16_grid(w), _bf16_grid(gy)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv3d(xd, wd, padding=3)
    (ref * gy.double()).sum().backward()
    prev = ops.set_conv_precision(precision)
    try:
        import ctypes as C

        from hiddenpose_amd import _lib
        xc = cl(x).cuda()
        desc = ops._desc(xc, 64, 7, 1, 3, False)
        # forward at the same odd extents (depth 22: the 16-plane tiles of the bf16-mode stem kernel end in a partial tile)
        wf, _ = ops._pack(desc, w.cuda(), True, False)
        y = torch.empty(B, D, H, W, 64, device="cuda")
        stats = torch.empty(_lib.STATS_SLOTS * 128, dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib().hp_conv3d_forward(C.byref(desc), xc.data_ptr(), wf.data_ptr(), None, y.data_ptr(), stats.data_ptr(),
                                                ops._stream(xc)), "fwd")
        dx, dw = ops._conv_grads(desc, xc, w.cuda(), cl(gy).cuda(), True)
    finally:
        ops.set_conv_precision(prev)
    assert rel_l2(ncdhw(y), ref) < 2e-6
    refcl = cl(ref.detach())
    stats = stats.view(_lib.STATS_SLOTS, 128).sum(0)
    assert rel_l2(stats[:64], refcl.reshape(-1, 64).sum(0)) < 2e-5 and rel_l2(stats[64:], (refcl.reshape(-1, 64) ** 2).sum(0)) < 2e-5
    assert rel_l2(ncdhw(dx), xd.grad) < 5e-6
    assert rel_l2(dw, wd.grad) < 5e-6


@pytest.mark.parametrize("train,relu,with_res", [(True, True, True), (True, True, False), (False, True, True), (True, False, False)])
def test_conv_bn_act_unit(train, relu, with_res):
    g = torch.Generator().manual_seed(11)
    B, C1, C2, D = 2, 64, 96, 6
    conv = torch.nn.Conv3d(C1, C2, 3, padding=1, bias=False)
    bn = torch.nn.BatchNorm3d(C2)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.05)
        bn.weight.copy_(1 + 0.2 * torch.randn(C2, generator=g))
        bn.bias.copy_(0.2 * torch.randn(C2, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(C2, generator=g))
        bn.running_var.copy_(1 + 0.1 * torch.rand(C2, generator=g))
    x = torch.randn(B, C1, D, D, D, generator=g)
    res = torch.randn(B, C2, D, D, D, generator=g) if with_res else None
    gy = torch.randn(B, C2, D, D, D, generator=g)

    import copy

    convr, bnr = copy.deepcopy(conv).double(), copy.deepcopy(bn).double()
    bnr.train(train)
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    o = bnr(convr(xr))
    if with_res:
        o = o + rr
    if relu:
        o = F.relu(o)
    (o * gy.double()).sum().backward()

    convg, bng = conv.cuda(), bn.cuda()
    bng.train(train)
    xg = cl(x).cuda().requires_grad_(True)
    rg = cl(res).cuda().requires_grad_(True) if with_res else None
    y = ops.conv_bn_act(xg, convg, bng, relu=relu, residual=rg)
    (y * cl(gy).cuda()).sum().backward()
    assert rel_l2(ncdhw(y), o) < 1e-5
    assert rel_l2(ncdhw(xg.grad), xr.grad) < 1e-4
    assert rel_l2(convg.weight.grad, convr.weight.grad) < 1e-4
    assert rel_l2(bng.weight.grad, bnr.weight.grad) < 1e-4
    assert rel_l2(bng.bias.grad, bnr.bias.grad) < 1e-4
    if with_res:
        assert rel_l2(ncdhw(rg.grad), rr.grad) < 1e-6
    assert rel_l2(bng.running_mean, bnr.running_mean) < 1e-5
    assert rel_l2(bng.running_var, bnr.running_var) < 1e-5
    assert int(bng.num_batches_tracked) == int(bnr.num_batches_tracked)


def test_maxpool_k3s2():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 8, 6, 8, 4, generator=g)
    xr = x.double().requires_grad_(True)
    o = F.max_pool3d(xr, 3, 2, 1)
    gy = torch.randn(o.shape, generator=g)
    (o * gy.double()).sum().backward()
    xg = cl(x).cuda().requires_grad_(True)
    y = ops._MaxPool3CL.apply(xg)
    (y * cl(gy).cuda()).sum().backward()
    assert rel_l2(ncdhw(y), o) == 0.0
    assert rel_l2(ncdhw(xg.grad), xr.grad) < 1e-7


@pytest.mark.parametrize("dims,train", [((2, 8, 8, 16), True), ((1, 6, 4, 6), True), ((1, 4, 8, 72), False)],
                         ids=["rows16", "ragged_rows", "eval"])
def test_stem_conv_bn_relu_pool_unit(dims, train):
    """Stem unit (posenet3d_50.py:252-257) against the float64 operators: full 32-voxel row tiles of the tiled
    BN + pool backward, a ragged tile (W = 6), and eval-mode statistics."""
    import copy

    g = torch.Generator().manual_seed(31)
    B, D, H, W = dims
    conv = torch.nn.Conv3d(1, 64, 7, padding=3, bias=False)
    bn = torch.nn.BatchNorm3d(64)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.05)
        bn.weight.copy_(1 + 0.2 * torch.randn(64, generator=g))
        bn.bias.copy_(0.2 * torch.randn(64, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(64, generator=g))
        bn.running_var.copy_(1 + 0.1 * torch.rand(64, generator=g))
    x = torch.randn(B, 1, D, H, W, generator=g)
    convr, bnr = copy.deepcopy(conv).double(), copy.deepcopy(bn).double()
    bnr.train(train)
    xr = x.double().requires_grad_(True)
    o = F.max_pool3d(F.relu(bnr(convr(xr))), 3, 2, 1)
    gy = torch.randn(o.shape, generator=g)
    (o * gy.double()).sum().backward()

    convg, bng = conv.cuda(), bn.cuda()
    bng.train(train)
    xg = x.cuda().requires_grad_(True)
    y = ops.stem_conv_bn_relu_pool(xg, convg, bng)
    (y * cl(gy).cuda()).sum().backward()
    assert rel_l2(ncdhw(y), o) < 1e-5
    assert rel_l2(xg.grad, xr.grad) < 1e-4
    assert rel_l2(convg.weight.grad, convr.weight.grad) < 1e-4
    assert rel_l2(bng.weight.grad, bnr.weight.grad) < 1e-4
    assert rel_l2(bng.bias.grad, bnr.bias.grad) < 1e-4


@pytest.mark.parametrize("C", [64, 32], ids=["tiled_c64", "general_c32"])
def test_stem_bn_relu_pool_abi(C):
    """hp_stem_bn_relu_pool_forward / _backward through the C ABI: the 64-channel tiled kernels and the general
    ones (any channel count) against float64 BatchNorm3d -> ReLU -> MaxPool3d(3,2,1) autograd."""
    import ctypes as Ct

    from hiddenpose_amd import _lib

    L = _lib.lib()
    g = torch.Generator().manual_seed(77)
    B, D, H, W = 2, 6, 4, 40
    z = torch.randn(B, C, D, H, W, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    zr = z.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    o = F.max_pool3d(F.relu(F.batch_norm(zr, None, None, gr, br, True, 0.1, 1e-5)), 3, 2, 1)
    gy = torch.randn(o.shape, generator=g)
    (o * gy.double()).sum().backward()
    zc = cl(z).cuda()
    mean = zc.reshape(-1, C).double().mean(0)
    var = zc.reshape(-1, C).double().var(0, unbiased=False)
    mean_f, rstd_f = mean.float(), (1.0 / torch.sqrt(var + 1e-5)).float()
    gam, bet = gamma.cuda(), beta.cuda()
    pooled = torch.empty(B, D // 2, H // 2, W // 2, C, device="cuda")
    ws = torch.empty(int(L.hp_stem_bn_pool_workspace_bytes(C)) // 4 + 4, device="cuda")
    st = ops._stream(zc)
    _lib.check(L.hp_stem_bn_relu_pool_forward(zc.data_ptr(), pooled.data_ptr(), B, D, H, W, C, mean_f.data_ptr(), rstd_f.data_ptr(),
                                              gam.data_ptr(), bet.data_ptr(), ws.data_ptr(), st), "fwd")
    assert rel_l2(ncdhw(pooled), o) < 1e-5
    dz, dgam, dbet = torch.empty_like(zc), torch.empty_like(gam), torch.empty_like(gam)
    dpc = cl(gy).cuda()
    _lib.check(L.hp_stem_bn_relu_pool_backward(zc.data_ptr(), pooled.data_ptr(), dpc.data_ptr(), dz.data_ptr(), B, D, H, W, C,
                                               mean_f.data_ptr(), rstd_f.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1,
                                               dgam.data_ptr(), dbet.data_ptr(), ws.data_ptr(), st), "bwd")
    assert rel_l2(ncdhw(dz), zr.grad) < 1e-4
    assert rel_l2(dgam, gr.grad) < 1e-4
    assert rel_l2(dbet, br.grad) < 1e-4


@pytest.mark.parametrize("stride,downsample,train", [(1, False, True), (1, True, True), (2, True, True), (2, True, False), (1, False, False)],
                         ids=["identity", "shortcut_conv", "shortcut_s2", "shortcut_s2_eval", "identity_eval"])
def test_bottleneck_forward_backward(stride, downsample, train):
    """Bottleneck (posenet3d_50.py:59-95) in train mode against the same block built from float64 torch modules:
    output, input gradient and every parameter gradient.  Covers the shortcut-gradient plumbing (GradLink with a
    masked addend for the identity shortcut, ResLink for the shortcut convolution)."""
    import copy

    from hiddenpose_amd.posenet3d_50 import Bottleneck

    g = torch.Generator().manual_seed(5 + stride)
    planes = 32  # the stride-2 data gradients need 32-channel multiples
    cin = planes * 4 if not downsample else 48
    ds = None
    if downsample:
        ds = torch.nn.Sequential(torch.nn.Conv3d(cin, planes * 4, 1, stride=stride, bias=False), torch.nn.BatchNorm3d(planes * 4))
    blk = Bottleneck(cin, planes, stride, ds)
    with torch.no_grad():
        for prm in blk.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * (0.2 if prm.dim() > 1 else 0.3) + (1.0 if prm.dim() == 1 else 0.0))
    B, D = 2, 8
    x = torch.randn(B, cin, D, D, D, generator=g)

    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(1 + 0.2 * torch.rand(m.running_var.shape, generator=g))
    ref = copy.deepcopy(blk).double().train(train)
    xr = x.double().requires_grad_(True)
    pre = torch.relu(xr * 1.0)  # a non-leaf input, as inside the network

    def ref_fwd(m, t):
        o = torch.relu(m.bn1(m.conv1(t)))
        o = torch.relu(m.bn2(m.conv2(o)))
        o = m.bn3(m.conv3(o))
        return torch.relu(o + (m.downsample(t) if m.downsample is not None else t))

    yr = ref_fwd(ref, pre)
    gy = torch.randn(yr.shape, generator=g)
    (yr * gy.double()).sum().backward()

    blk = blk.cuda().train(train)
    xg = cl(x).cuda().requires_grad_(True)
    y = blk(torch.relu(xg * 1.0))
    (y * cl(gy).cuda()).sum().backward()
    assert rel_l2(ncdhw(y), yr) < 1e-5
    assert rel_l2(ncdhw(xg.grad), xr.grad) < 1e-4
    for (n, pg), (_, pr) in zip(blk.named_parameters(), ref.named_parameters()):
        assert rel_l2(pg.grad, pr.grad) < 2e-4, n


def test_posenet_vs_reference_golden(golden):
    from hiddenpose_amd.posenet3d_50 import get_pose_net_50

    g = golden("posenet_io.npz")
    net = get_pose_net_50()
    hpt.fill_module(net, "pose_net.")
    net = net.cuda()
    x = (hpt.synthetic_meas(1, 32, 32, "uniform", seed=104) * 10.0).cuda()
    net.eval()
    with torch.no_grad():
        y = net(x)
    assert rel_l2(y, g["eval_y"]) < 1e-4
    net.train()
    x2 = (hpt.synthetic_meas(2, 32, 32, "uniform", seed=105) * 10.0).cuda().requires_grad_(True)
    yt = net(x2)
    gy = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal(tuple(yt.shape)).astype(np.float32)).cuda()
    (yt * gy).sum().backward()
    assert rel_l2(yt, g["train_y"]) < 1e-3
    assert rel_l2(x2.grad, g["train_gx"]) < 1e-2
    sd = net.state_dict()
    assert rel_l2(sd["bn1.running_mean"], g["train_bn1_running_mean"]) < 1e-4
    assert rel_l2(sd["bn1.running_var"], g["train_bn1_running_var"]) < 1e-4
    assert rel_l2(sd["head.features.7.running_var"], g["train_head7_running_var"]) < 1e-3
    named = dict(net.named_parameters())
    for k in ["conv1.weight", "bn1.weight", "layer1.0.conv2.weight", "layer1.0.downsample.0.weight", "layer4.2.bn3.bias",
              "head.features.9.weight"]:
        if "train_g_" + k in g:
            assert rel_l2(named[k].grad, g["train_g_" + k]) < 1e-2, k
        else:
            idx = g["train_gidx_" + k]
            gn = named[k].grad.cpu().numpy().reshape(-1)
            assert rel_l2(gn[idx], g["train_gs_" + k]) < 1e-2, k
            assert abs(np.sqrt((gn.astype(np.float64) ** 2).sum()) / float(g["train_gl2_" + k]) - 1) < 1e-2, k


def test_weight_gradients_on_side_stream_match_serial():
    """hip_ops.set_wgrad_async: weight gradients queued on a second stream are complete when backward() returns
    (end-of-backward callback) and equal the serial ones up to the summation order of the split-K atomics."""
    from hiddenpose_amd.posenet3d_50 import get_pose_net_50

    net = get_pose_net_50()
    hpt.fill_module(net, "pose_net.")
    net = net.cuda().train()
    x = (hpt.synthetic_meas(2, 32, 32, "uniform", seed=105) * 10.0).cuda()

    def grads(async_on):
        prev = ops.set_wgrad_async(async_on)
        try:
            net.zero_grad(set_to_none=True)
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm3d):
                    m.reset_running_stats()
            y = net(x)
            (y * y).mean().backward()
            return {k: p.grad.clone() for k, p in net.named_parameters()}
        finally:
            ops.set_wgrad_async(prev)

    ref = grads(False)
    got = grads(True)
    for k in ref:
        assert rel_l2(got[k], ref[k]) < 1e-4, k


BF16S_CASES = [c for c in CASES if c[0] != "stem"] + [
    # power-of-two grids with rows a multiple of a weight-gradient step (32 voxels): the wave-uniform row addressing of the
    # bf16-storage weight gradient, with taps that leave the row on both sides, a strided and a transposed case
    ("k3_runx", 64, 64, 3, 1, 1, False, (2, 4, 8, 32)),
    ("k3_s2_runx", 64, 128, 3, 2, 1, False, (1, 4, 8, 64)),
    ("deconv_runx", 64, 64, 4, 2, 1, True, (1, 2, 4, 32)),
]


@pytest.mark.parametrize("name,cin,cout,k,s,p,tr,dims", BF16S_CASES, ids=[c[0] for c in BF16S_CASES])
def test_conv_bf16_storage(name, cin, cout, k, s, p, tr, dims):
    """BASELINE configs[2] storage: the same kernels with bf16 ACTIVATION tensors in memory (hp_conv_desc.io) -- the
    gathered tensor read as 8-element bf16 runs (64-deep K tiles when the channel count allows, bf16 packed weights),
    the written tensor optionally bf16.  Operands on the bf16 grid make the arithmetic exact up to accumulation order, so
    the float64 operator is the expected value: fp32 outputs to the fp32 bar, bf16 outputs to one bf16 rounding."""
    import ctypes as C

    from hiddenpose_amd import _lib

    g = torch.Generator().manual_seed(sum(map(ord, name)) + 1)
    B, D, H, W = dims
    x = _bf16_grid(torch.randn(B, cin, D, H, W, generator=g))
    w = _bf16_grid(torch.randn((cin, cout, k, k, k) if tr else (cout, cin, k, k, k), generator=g) / np.sqrt(cin * k ** 3 / s ** 3))
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv_transpose3d(xd, wd, stride=s, padding=p) if tr else F.conv3d(xd, wd, stride=s, padding=p)
    gy = _bf16_grid(torch.randn(ref.shape, generator=g))
    (ref * gy.double()).sum().backward()

    L = _lib.lib()
    xh = cl(x).cuda().bfloat16()
    wc = w.cuda()
    prev = ops.set_conv_precision("bf16s")
    try:
        desc = ops._desc(xh, cout, k, s, p, tr)
    finally:
        ops.set_conv_precision(prev)
    assert desc.precision == 1
    st = ops._stream(xh)
    wh = ops._w_half(desc, True, cin)
    assert wh == (cin % 64 == 0)
    wf, _ = ops._pack(desc, wc, True, False, wh)
    assert wf.dtype == (torch.bfloat16 if wh else torch.float32)
    do, ho, wo = ops._out_dims(desc)
    stats = torch.empty(_lib.STATS_SLOTS * 2 * cout, dtype=torch.float64, device="cuda")
    for yh in (False, True):
        y = torch.empty(B, do, ho, wo, cout, device="cuda", dtype=torch.bfloat16 if yh else torch.float32)
        desc.io = ops.HP_IO_X | (ops.HP_IO_W if wh else 0) | (ops.HP_IO_Y if yh else 0)
        _lib.check(L.hp_conv3d_forward(C.byref(desc), xh.data_ptr(), wf.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st), "fwd")
        assert rel_l2(ncdhw(y.float()), ref) < (4e-3 if yh else 2e-6)
        refcl = cl(ref.detach())
        assert rel_l2(stats.view(_lib.STATS_SLOTS, 2 * cout).sum(0)[:cout], refcl.reshape(-1, cout).sum(0)) < 2e-5   # from the fp32 accumulators
    dzh = cl(gy).cuda().bfloat16()
    dx, dw = ops._conv_grads(desc, xh, wc, dzh, True)
    assert dx.dtype == torch.bfloat16 and dw.dtype == torch.float32
    assert rel_l2(ncdhw(dx.float()), xd.grad) < 4e-3
    assert rel_l2(dw, wd.grad) < 5e-6
    # an addend of the data gradient (second contribution to the same tensor) in bf16 as well
    add = _bf16_grid(torch.randn(xd.shape, generator=g))
    if not (k == 1 and s == 2):   # the strided 1^3 data gradient accumulates in place (covered by the model tests)
        dx2, _ = ops._conv_grads(desc, xh, wc, dzh, True, cl(add).cuda().bfloat16())
        assert rel_l2(ncdhw(dx2.float()), xd.grad + add.double()) < 4e-3


@pytest.mark.parametrize("stride,downsample", [(1, False), (2, True)], ids=["identity", "downsample_s2"])
def test_bottleneck_bf16_storage(stride, downsample):
    """A whole Bottleneck with bf16 activation storage (64 / 256 channels: every convolution takes the direct-to-LDS
    tiles, the identity shortcut the 16-byte masked-addend epilogue, the stride-2 block the parity-class data gradient)
    against the same block in float64.  Each of the ~10 tensors between input and output is rounded to bf16 once
    (2^-9) and the ReLU masks move with it, so the bars are a few per cent -- an indexing slip anywhere in the tile
    loads, the swizzle or the epilogue shows as an O(1) error."""
    import copy

    from hiddenpose_amd.posenet3d_50 import Bottleneck

    g = torch.Generator().manual_seed(31 + stride)
    planes = 64
    cin = 128 if downsample else planes * 4
    ds = None
    if downsample:
        ds = torch.nn.Sequential(torch.nn.Conv3d(cin, planes * 4, 1, stride=stride, bias=False), torch.nn.BatchNorm3d(planes * 4))
    blk = Bottleneck(cin, planes, stride, ds)
    with torch.no_grad():
        for prm in blk.parameters():
            prm.copy_(_bf16_grid(torch.randn(prm.shape, generator=g) * (0.08 if prm.dim() > 1 else 0.3) + (1.0 if prm.dim() == 1 else 0.0)))
    B, D = 2, 8
    x = _bf16_grid(torch.randn(B, cin, D, D, D, generator=g))
    ref = copy.deepcopy(blk).double().train(True)
    xr = x.double().requires_grad_(True)
    pre = torch.relu(xr * 1.0)
    o = torch.relu(ref.bn1(ref.conv1(pre)))
    o = torch.relu(ref.bn2(ref.conv2(o)))
    o = ref.bn3(ref.conv3(o))
    yr = torch.relu(o + (ref.downsample(pre) if ref.downsample is not None else pre))
    gy = _bf16_grid(torch.randn(yr.shape, generator=g))
    (yr * gy.double()).sum().backward()

    blk = blk.cuda().train(True)
    prev = ops.set_conv_precision("bf16s")
    try:
        xg = cl(x).cuda().requires_grad_(True)
        y = blk(torch.relu(xg * 1.0).to(torch.bfloat16))
        assert y.dtype == torch.bfloat16
        (y.float() * cl(gy).cuda()).sum().backward()
    finally:
        ops.set_conv_precision(prev)
    assert rel_l2(ncdhw(y.float()), yr) < 2e-2
    assert rel_l2(ncdhw(xg.grad), xr.grad) < 8e-2
    for (n, pg), (_, pr) in zip(blk.named_parameters(), ref.named_parameters()):
        assert rel_l2(pg.grad, pr.grad) < 0.15, n   # measured <= 0.08 (BatchNorm biases: sums over flipped ReLU masks)


@pytest.mark.parametrize("with_res", [True, False])
def test_conv_bn_act_unit_bf16_storage(with_res):
    """One conv + BatchNorm + ReLU (+ residual) unit with bf16 activation storage against the same unit in the bf16
    mode with fp32 storage (identical arithmetic; only the roundings of z, y, dz, dx differ: one bf16 ulp each)."""
    g = torch.Generator().manual_seed(12)
    B, C1, C2, D = 2, 64, 128, 6
    conv = torch.nn.Conv3d(C1, C2, 3, padding=1, bias=False).cuda()
    bn = torch.nn.BatchNorm3d(C2).cuda()
    with torch.no_grad():
        conv.weight.copy_(_bf16_grid(torch.randn(conv.weight.shape, generator=g) * 0.05))
        bn.weight.copy_(1 + 0.2 * torch.randn(C2, generator=g))
        bn.bias.copy_(0.2 * torch.randn(C2, generator=g))
    x = _bf16_grid(torch.randn(B, D, D, D, C1, generator=g)).cuda()
    res = _bf16_grid(torch.randn(B, D, D, D, C2, generator=g)).cuda() if with_res else None
    gy = _bf16_grid(torch.randn(B, D, D, D, C2, generator=g)).cuda()
    out = {}
    for mode in ("bf16", "bf16s"):
        conv.zero_grad()
        bn.zero_grad()
        bn.train()
        dt = torch.bfloat16 if mode == "bf16s" else torch.float32
        xi = x.detach().to(dt).clone().requires_grad_(True)
        ri = res.detach().to(dt).clone().requires_grad_(True) if with_res else None
        prev = ops.set_conv_precision(mode)
        try:
            y = ops.conv_bn_act(xi, conv, bn, relu=True, residual=ri)
        finally:
            ops.set_conv_precision(prev)
        assert y.dtype == dt
        (y.float() * gy).sum().backward()
        out[mode] = (y.detach().float(), xi.grad.float(), ri.grad.float() if with_res else None, conv.weight.grad.clone(),
                     bn.weight.grad.clone(), bn.bias.grad.clone())
    a, b = out["bf16"], out["bf16s"]
    assert rel_l2(b[0], a[0]) < 6e-3 and rel_l2(b[1], a[1]) < 3e-2
    if with_res:   # g = dy (.) [y > 0]: z rounded to bf16 moves outputs within 2^-9 of zero across it (mask flips)
        assert rel_l2(b[2], a[2]) < 4e-2
    assert rel_l2(b[3], a[3]) < 3e-2 and rel_l2(b[4], a[4]) < 3e-2 and rel_l2(b[5], a[5]) < 3e-2


def test_side_stream_weight_gradient_with_a_tensor_hook_or_shared_weight():
    """with weight gradients on the side stream (set_wgrad_async) a weight that carries a tensor hook, or that is
    used twice in one graph (autograd then SUMS the two gradients on the main stream), must not be handed to autograd before
    the side stream has written it: such weights fall back to the main-stream path.  Results equal the serial mode."""
    from torch import nn

    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 6, 6, 8, 64, generator=g).cuda()

    def run(async_on):
        conv = nn.Conv3d(64, 64, 3, padding=1, bias=False).cuda()
        bn = nn.BatchNorm3d(64).cuda()
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=torch.Generator().manual_seed(9)).cuda() * 0.05)
        seen = []
        conv.weight.register_hook(lambda gr: seen.append(float(gr.double().norm())))   # reads the gradient on the main stream
        prev = ops.set_wgrad_async(async_on)
        try:
            y = ops.conv_bn_act(ops.conv_bn_act(x, conv, bn), conv, bn)               # the same weight twice in one graph
            y.square().mean().backward()
            torch.cuda.synchronize()
        finally:
            ops.set_wgrad_async(prev)
        return conv.weight.grad.clone(), seen

    g0, s0 = run(False)
    g1, s1 = run(True)
    assert rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-5
    assert len(s0) == len(s1) and all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(s0, s1))


def test_a_convolution_weight_used_twice_in_one_graph_stays_on_the_main_stream():
    """The shared-weight rule without a hook in play (round 4: the use counts are taken by the wrappers conv_bn_act /
    deconv_bn_relu, where grad mode is the caller's -- inside Function.forward it is always off, so the counts of rounds 3-4 never
    counted): the same Conv3d twice in one graph -> autograd SUMS the two weight gradients on the main stream, so neither may be
    written on the second stream; a weight used once takes it.  Gradients equal the one-stream run."""
    from torch import nn

    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 6, 6, 8, 64, generator=g).cuda()
    w0 = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).cuda()

    def run(async_on, twice):
        conv, bn = nn.Conv3d(64, 64, 3, padding=1, bias=False).cuda(), nn.BatchNorm3d(64).cuda()
        with torch.no_grad():
            conv.weight.copy_(w0)
        prev = ops.set_wgrad_async(async_on)
        n0 = ops._side_conv_calls[0]
        try:
            y = ops.conv_bn_act(x, conv, bn)
            if twice:
                y = ops.conv_bn_act(y, conv, bn)
            y.square().mean().backward()
            torch.cuda.synchronize()
        finally:
            ops.set_wgrad_async(prev)
        assert not getattr(conv.weight, "_hp_uses", {}) and not getattr(conv.weight, "_hp_shared", False)
        return conv.weight.grad.clone(), ops._side_conv_calls[0] - n0

    for twice in (True, False):
        (g1, n1), (g0, n0_) = run(True, twice), run(False, twice)
        assert n0_ == 0 and n1 == (0 if twice else 1), (twice, n1, n0_)
        assert rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("ca,cb,kb,transposed,dims", [
    (64, 64, 3, False, (2, 4, 8, 8)),      # conv1 -> conv2 of a Bottleneck: 64-column tile, 3^3 consumer
    (64, 256, 1, False, (2, 4, 8, 8)),     # conv2 -> conv3: 64-column tile, 1^3 consumer
    (128, 128, 3, False, (1, 8, 8, 8)),    # 128-column tile
    (256, 64, 1, False, (1, 4, 8, 8)),     # two N tiles of 128
    (256, 256, 4, True, (1, 4, 4, 8)),     # deconvolution consumer (DeconvHead)
    (256, 24, 1, False, (2, 4, 4, 8)),     # the head's 1^3 convolution behind the last deconvolution's BatchNorm (K = 24: flat loads)
    (96, 64, 1, False, (1, 4, 8, 8)),      # 96 channels: not a whole number of 64-column tiles -> falls back, same result
    (64, 64, 3, False, (1, 3, 5, 8)),      # 120 rows: not whole M tiles -> falls back
])
def test_bn_backward_sums_taken_by_the_consumers_data_gradient(ca, cb, kb, transposed, dims):
    """hip_ops.BnLink (round 4): unit a = conv + BatchNorm + ReLU without residual, consumed by ONE convolution b.  b's data
    gradient IS a's incoming gradient, so hp_conv3d_backward_data_bnsums takes a's two BatchNorm-backward sums from the tile in
    hand (one read of z_a) and hp_bn_backward_presummed skips the reduction pass.  Every gradient of the chain equals the
    un-linked evaluation (same arithmetic, different summation order: 1e-5) and the float64 reference; geometries the whole-tile
    kernels do not cover fall back without a trace (sums stay None)."""
    g = torch.Generator().manual_seed(100 + ca + cb + kb)
    B, D, H, W = dims
    conv_a, bn_a = torch.nn.Conv3d(32, ca, 1, bias=False), torch.nn.BatchNorm3d(ca)
    conv_b = (torch.nn.ConvTranspose3d(ca, cb, 4, stride=2, padding=1, bias=False) if transposed
              else torch.nn.Conv3d(ca, cb, kb, padding=kb // 2, bias=False))
    bn_b = torch.nn.BatchNorm3d(cb)
    with torch.no_grad():
        for m in (conv_a, conv_b):
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.05)
        for bn in (bn_a, bn_b):
            bn.weight.copy_(1 + 0.2 * torch.randn(bn.weight.shape, generator=g))
            bn.bias.copy_(0.2 * torch.randn(bn.bias.shape, generator=g))
    x = torch.randn(B, 32, D, H, W, generator=g)
    import copy

    ref = [copy.deepcopy(m).double().train() for m in (conv_a, bn_a, conv_b, bn_b)]
    xr = x.double().requires_grad_(True)
    o = F.relu(ref[3](ref[2](F.relu(ref[1](ref[0](xr))))))
    gy = torch.randn(o.shape, generator=g)
    (o * gy.double()).sum().backward()

    mods = [m.cuda().train() for m in (conv_a, bn_a, conv_b, bn_b)]
    results, taken = [], []
    for linked in (True, False):
        for m in mods:
            m.zero_grad()
        xg = cl(x).cuda().requires_grad_(True)
        link = ops.BnLink() if linked else None
        before = ops._bn_fused_calls[0]
        ya = ops.conv_bn_act(xg, mods[0], mods[1], relu=True, bn_out=link)
        yb = (ops.deconv_bn_relu(ya, mods[2], mods[3], bn_in=link) if transposed
              else ops.conv_bn_act(ya, mods[2], mods[3], relu=True, bn_in=link))
        (yb * cl(gy).cuda()).sum().backward()
        results.append([ncdhw(xg.grad)] + [p.grad.clone() for m in mods for p in m.parameters()])
        taken.append(ops._bn_fused_calls[0] - before)
        if linked:
            assert link.src is not None and link.sums is None      # handed over and consumed (or never produced)
    whole = (B * D * H * W) % 128 == 0 and ca % 64 == 0 and (ca <= 64 or ca % 128 == 0)
    assert taken == [1 if whole else 0, 0], (taken, whole)
    refs = [xr.grad] + [p.grad for m in ref for p in m.parameters()]
    for a, b, r in zip(results[0], results[1], refs):
        assert rel_l2(a, b) < 1e-5          # linked vs un-linked: the same sums in another order
        assert rel_l2(a, r) < 1e-4          # and the float64 reference


def test_block_output_units_hand_their_sums_to_the_next_blocks_conv1():
    """Three Bottlenecks in an nn.Sequential (a shortcut-convolution block, then two identity blocks; 64 planes, 256 channels,
    1024 voxels) followed by a deconvolution unit, float64 reference built from the same modules.  Besides bn1 / bn2 of every
    block (test above), the OUTPUT units of blocks 1 and 2 (bn3 + identity shortcut) hand their BatchNorm-backward sums to the
    data gradient in which the block-output gradient becomes complete -- conv1 of the next block (masked shortcut addend
    included), the deconvolution for the last one -- through hip_ops.offer_res_bn / take_res_bn; block 0's output unit
    (paired with the shortcut's BatchNorm in the dual pass) does not.  With the hand-overs off the same gradients come out.
    (The block-output hand-over is built and tested but OFF by default, HP_BN_FUSE_RES: its consumers are the HBM-bound 1^3
    data gradients with 4 x planes output channels, and the extra read of z costs them what the saved pass gains.)"""
    import copy

    from hiddenpose_amd.posenet3d_50 import Bottleneck

    g = torch.Generator().manual_seed(77)
    planes, cin = 64, 128
    ds = torch.nn.Sequential(torch.nn.Conv3d(cin, planes * 4, 1, bias=False), torch.nn.BatchNorm3d(planes * 4))
    net = torch.nn.Sequential(Bottleneck(cin, planes, 1, ds), Bottleneck(planes * 4, planes), Bottleneck(planes * 4, planes))
    dec, dbn = torch.nn.ConvTranspose3d(planes * 4, 128, 4, stride=2, padding=1, bias=False), torch.nn.BatchNorm3d(128)
    with torch.no_grad():
        for prm in list(net.parameters()) + list(dec.parameters()) + list(dbn.parameters()):
            prm.copy_(torch.randn(prm.shape, generator=g) * (0.08 if prm.dim() > 1 else 0.3) + (1.0 if prm.dim() == 1 else 0.0))
    B, D = 2, 8
    x = torch.randn(B, cin, D, D, D, generator=g)
    refn, refd, refb = copy.deepcopy(net).double().train(), copy.deepcopy(dec).double(), copy.deepcopy(dbn).double().train()
    xr = x.double().requires_grad_(True)
    t = torch.relu(xr * 1.0)
    for m in refn:
        o = torch.relu(m.bn1(m.conv1(t)))
        o = torch.relu(m.bn2(m.conv2(o)))
        o = m.bn3(m.conv3(o))
        t = torch.relu(o + (m.downsample(t) if m.downsample is not None else t))
    yr = torch.relu(refb(refd(t)))
    gy = torch.randn(yr.shape, generator=g)
    (yr * gy.double()).sum().backward()
    ref_grads = [xr.grad] + [p.grad for p in list(refn.parameters()) + list(refd.parameters()) + list(refb.parameters())]

    net, dec, dbn = net.cuda().train(), dec.cuda(), dbn.cuda().train()
    out = []
    for fuse in (True, False):
        prev = (ops._BN_FUSE, ops._BN_FUSE_RES)
        ops._BN_FUSE, ops._BN_FUSE_RES = fuse, fuse      # the block-output hand-over is opt-in (see hip_ops._BN_FUSE_RES)
        try:
            for p in list(net.parameters()) + list(dec.parameters()) + list(dbn.parameters()):
                p.grad = None
            before = ops._bn_fused_calls[0]
            xg = cl(x).cuda().requires_grad_(True)
            h = net(torch.relu(xg * 1.0))
            y = ops.deconv_bn_relu(h, dec, dbn, bn_in=ops.take_res_bn(h))
            (y * cl(gy).cuda()).sum().backward()
            out.append(([ncdhw(xg.grad)] + [p.grad.clone() for p in list(net.parameters()) + list(dec.parameters()) + list(dbn.parameters())],
                        ops._bn_fused_calls[0] - before))
        finally:
            ops._BN_FUSE, ops._BN_FUSE_RES = prev
    # 3 blocks x (bn1, bn2) + the output units of blocks 1 and 2 (block 0's is the dual pass with its shortcut BatchNorm)
    assert out[0][1] == 8 and out[1][1] == 0, (out[0][1], out[1][1])
    for a, b, r in zip(out[0][0], out[1][0], ref_grads):
        assert rel_l2(a, b) < 2e-5          # with and without the hand-over: the same sums in another order
        assert rel_l2(a, r) < 2e-2          # float64: ten train-mode BatchNorms over 1024 values and their ReLU masks deep (1e-3 .. 5e-3)
