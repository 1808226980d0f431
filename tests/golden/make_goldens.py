#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by importing the reference
(read-only checkout at /root/reference) through ref_shims.install().

Run in the dev container only:   python tests/golden/make_goldens.py [section ...]
Sections: lctwin ingest sformer schema consts consts_hr lct parts posenet e2e e2e128 e2e512 softargmax specular bp e2e128train e2e128train_smooth highres e2e512train e2e512train_b1 e2e512train_repro
(default: all; highres needs ~45 GB of RAM and ~15 minutes, e2e512train ~55 GB and ~15 minutes)

Inputs come from hiddenpose_amd.testing (seeded, closed form); weights from
its filler keyed by state_dict name, so tests rebuild identical inputs and
weights without committing them.  Only reference OUTPUTS are stored.
"""
from __future__ import annotations

import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402

ref_shims.install()

from hiddenpose_amd import testing as hpt  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)

BIN_LEN = {(32, 16): 0.16, (32, 32): 0.16, (128, 128): 0.04, (512, 128): 0.01, (64, 32): 0.08, (1024, 256): 0.005}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


def sample_idx(n_total: int, n: int, seed: int) -> np.ndarray:
    return np.random.Generator(np.random.PCG64(seed)).choice(n_total, size=n, replace=False).astype(np.int64)


def sec_consts():
    import models.feature_propagation as fp

    out = {}
    for (T, N) in [(32, 16), (128, 128), (512, 128)]:
        t0 = time.time()
        lct = fp.LCT(N, T, BIN_LEN[(T, N)], 2.0)
        tag = f"T{T}_N{N}"
        mtx = lct.mtx_MxM.numpy()
        r, c = np.nonzero(mtx)
        out[f"{tag}_mtx_rows"] = r.astype(np.int32)
        out[f"{tag}_mtx_cols"] = c.astype(np.int32)
        out[f"{tag}_mtx_vals"] = mtx[r, c]
        out[f"{tag}_gridz"] = lct.gridz_1xMx1x1.numpy().reshape(-1)
        slope = lct.width / lct.trange
        out[f"{tag}_slope"] = np.float64(slope)
        psf = lct._definePsf(N, T, slope)
        ind = (psf > 0).astype(np.uint8)
        out[f"{tag}_psf_sha1"] = np.frombuffer(hashlib.sha1(ind.tobytes()).digest(), dtype=np.uint8)
        out[f"{tag}_psf_nnz"] = np.int64(ind.sum())
        out[f"{tag}_psf_val"] = np.float32(psf.max())
        # z index of the single 1 in every (x,y) column, after the roll (layout (2M,2N,2N))
        out[f"{tag}_psf_zidx"] = ind.argmax(axis=0).astype(np.int16)
        re = lct.invpsf_real.numpy()[0]
        im = lct.invpsf_imag.numpy()[0]
        if T * N * N <= 32 * 16 * 16:
            out[f"{tag}_invpsf_re"] = re
            out[f"{tag}_invpsf_im"] = im
        else:
            idx = sample_idx(re.size, 512, 7)
            out[f"{tag}_invpsf_idx"] = idx
            out[f"{tag}_invpsf_re_s"] = re.reshape(-1)[idx]
            out[f"{tag}_invpsf_im_s"] = im.reshape(-1)[idx]
        out[f"{tag}_invpsf_l2"] = np.float64(np.sqrt((re.astype(np.float64) ** 2 + im.astype(np.float64) ** 2).sum()))
        print(f"  consts {tag}: {time.time()-t0:.1f}s nnz(mtx)={len(r)}")
    save("lct_consts.npz", **out)


def sec_consts_hr():
    """The PSF of BASELINE configs[3] (N = 256, M = 1024) from the reference's OWN _definePsf (feature_propagation.py:141-171):
    ~10 float32 arrays of (512, 512, 2048) = 2.1 GB each live at once (about 20 GB peak; the fftn of it, which the LCT
    constructor would add, is not needed to pin the indicator).  Stored: sha1 of the uint8 indicator, the z index of the 1 in
    every (x, y) column after the roll, the count, the value, the slope."""
    import models.feature_propagation as fp

    T, N = 1024, 256
    t0 = time.time()
    bin_len, wall = BIN_LEN[(T, N)], 2.0
    slope = (wall / 2.0) / (T * 3e8 * (bin_len / 3e8))      # :72-78, as LCT._parpareparam evaluates it
    psf = fp.LCT._definePsf(None, N, T, slope)              # the method never touches self
    ind = (psf > 0).astype(np.uint8)
    out = {"T1024_N256_slope": np.float64(slope),
           "T1024_N256_psf_sha1": np.frombuffer(hashlib.sha1(np.ascontiguousarray(ind).tobytes()).digest(), dtype=np.uint8),
           "T1024_N256_psf_nnz": np.int64(ind.sum(dtype=np.int64)), "T1024_N256_psf_val": np.float32(psf.max()),
           "T1024_N256_psf_zidx": ind.argmax(axis=0).astype(np.int16)}
    save("lct_consts_hr.npz", **out)
    print(f"  consts T1024_N256: {time.time()-t0:.1f}s nnz {int(out['T1024_N256_psf_nnz'])} val {float(out['T1024_N256_psf_val']):.6g}")


def sec_lct():
    import models.feature_propagation as fp

    out = {}
    # small: full tensors
    T, N, B = 32, 16, 2
    lct = fp.LCT(N, T, BIN_LEN[(T, N)], 2.0)
    x = hpt.synthetic_meas(B, T, N, "uniform", seed=0).requires_grad_(True)
    y = lct(x, [0] * B, [T] * B)
    gy = hpt.synthetic_meas(B, T, N, "uniform", seed=100) - 0.5
    (y * gy).sum().backward()
    out["small_y"] = y.detach().numpy()
    out["small_gx"] = x.grad.numpy()
    # B=3 with the transient generator (odd batch)
    x3 = hpt.synthetic_meas(3, T, N, "transient", seed=410)
    out["small3_y"] = lct(x3, [0] * 3, [T] * 3).detach().numpy()
    # large: sampled
    for (T, N) in [(128, 128), (512, 128)]:
        lct = fp.LCT(N, T, BIN_LEN[(T, N)], 2.0)
        x = hpt.synthetic_meas(1, T, N, "transient", seed=410).requires_grad_(True)
        t0 = time.time()
        y = lct(x, [0], [T])
        gy = hpt.synthetic_meas(1, T, N, "uniform", seed=100) - 0.5
        (y * gy).sum().backward()
        tag = f"T{T}_N{N}"
        idx = sample_idx(y.numel(), 1024, 11)
        yn = y.detach().numpy().reshape(-1)
        gn = x.grad.numpy().reshape(-1)
        out[f"{tag}_idx"] = idx
        out[f"{tag}_y_s"] = yn[idx]
        out[f"{tag}_gx_s"] = gn[idx]
        out[f"{tag}_y_l2"] = np.float64(np.sqrt((yn.astype(np.float64) ** 2).sum()))
        out[f"{tag}_gx_l2"] = np.float64(np.sqrt((gn.astype(np.float64) ** 2).sum()))
        out[f"{tag}_y_minmax"] = np.array([yn.min(), yn.max()], dtype=np.float32)
        print(f"  lct {tag}: fwd+bwd {time.time()-t0:.1f}s")
    save("lct_io.npz", **out)


def sec_parts():
    """FeatureExtraction, normalize_feature, UNet3d on (B=2, T=32, N=32)."""
    from models.feature_extraction import FeatureExtraction
    from models.feature_propagation import normalize_feature
    from unet.unet3d import UNet3d

    B, T, N = 2, 32, 32
    out = {}
    x = hpt.synthetic_meas(B, T, N, "transient", seed=410)

    fe = FeatureExtraction(basedim=1, in_channels=1, stride=1)
    hpt.fill_module(fe, "feature_extraction.")
    xi = x.clone().requires_grad_(True)
    y = fe(xi)
    gy = hpt.synthetic_meas(B, T, N, "uniform", seed=101) - 0.5
    (y * gy).sum().backward()
    out["fe_y"] = y.detach().numpy()
    out["fe_gx"] = xi.grad.numpy()
    for k, p in fe.named_parameters():
        out["fe_g_" + k] = p.grad.numpy()

    z = (hpt.synthetic_meas(B, T, N, "uniform", seed=102) - 0.3) * 1e-4
    zi = z.clone().requires_grad_(True)
    nz = normalize_feature(zi)
    (nz * gy).sum().backward()
    out["norm_y"] = nz.detach().numpy()
    out["norm_gx"] = zi.grad.numpy()

    un = UNet3d(in_channels=1, n_channels=4)
    hpt.fill_module(un, "autoencoder.")
    ui = (hpt.synthetic_meas(B, T, N, "uniform", seed=103) * 10.0).requires_grad_(True)
    uy = un(ui)
    (uy * gy).sum().backward()
    out["unet_y"] = uy.detach().numpy()
    out["unet_gx"] = ui.grad.numpy()
    for k in ["conv.double_conv.0.weight", "conv.double_conv.1.weight", "enc4.encoder.1.double_conv.3.weight",
              "dec1.conv.double_conv.0.weight", "dec4.conv.double_conv.4.bias", "out.conv.weight", "out.conv.bias"]:
        out["unet_g_" + k] = dict(un.named_parameters())[k].grad.numpy()
    save("parts_io.npz", **out)


def sec_posenet():
    from models.posenet3d_50 import get_pose_net_50

    out = {}
    net = get_pose_net_50()
    hpt.fill_module(net, "pose_net.")
    x = hpt.synthetic_meas(1, 32, 32, "uniform", seed=104) * 10.0
    net.eval()
    with torch.no_grad():
        out["eval_y"] = net(x).numpy()
    net.train()
    x2 = hpt.synthetic_meas(2, 32, 32, "uniform", seed=105) * 10.0
    xi = x2.clone().requires_grad_(True)
    y = net(xi)
    gy = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal(tuple(y.shape)).astype(np.float32))
    (y * gy).sum().backward()
    out["train_y"] = y.detach().numpy()
    out["train_gx"] = xi.grad.numpy()
    sd = net.state_dict()
    out["train_bn1_running_mean"] = sd["bn1.running_mean"].numpy()
    out["train_bn1_running_var"] = sd["bn1.running_var"].numpy()
    out["train_head7_running_var"] = sd["head.features.7.running_var"].numpy()
    named = dict(net.named_parameters())
    for k in ["conv1.weight", "bn1.weight", "layer1.0.conv2.weight", "layer1.0.downsample.0.weight",
              "layer2.0.conv2.weight", "layer3.5.conv3.weight", "layer4.2.bn3.bias",
              "head.features.0.weight", "head.features.6.weight", "head.features.9.weight", "head.features.9.bias"]:
        g = named[k].grad.numpy()
        if g.size > 70000:
            idx = sample_idx(g.size, 4096, 13)
            out["train_gidx_" + k] = idx
            out["train_gs_" + k] = g.reshape(-1)[idx]
            out["train_gl2_" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        else:
            out["train_g_" + k] = g
    save("posenet_io.npz", **out)


E2E_PARAMS = ["feature_extraction.weights", "feature_extraction.conv1.1.weight", "feature_extraction.conv1.3.tmp.4.bias",
              "autoencoder.conv.double_conv.0.weight", "autoencoder.out.conv.bias", "pose_net.conv1.weight",
              "pose_net.layer2.0.conv2.weight", "pose_net.head.features.9.bias", "pose_net.bn1.weight"]


def _e2e(T, N, B, full):
    from models.NlosPose import NlosPose
    from utils.criterion import BCEDiceLoss, L2JointLocationLoss, softmax_integral_tensor

    cfg = ref_shims.make_cfg(T, N, BIN_LEN[(T, N)])
    model = NlosPose(cfg)
    hpt.fill_module(model)
    out = {}
    meas = hpt.synthetic_meas(B, T, N, "transient", seed=410)
    vol = hpt.synthetic_vol(B, T, N)
    joints = hpt.synthetic_joints(B, T // 2).reshape(B, -1)
    hm = (N // 2, N // 2, T // 2)

    model.eval()
    t0 = time.time()
    with torch.no_grad():
        heat_e, refine_e = model(meas)
        j_e = softmax_integral_tensor(heat_e, 24, True, hm[0], hm[1], hm[2])
    print(f"  e2e T{T} N{N} B{B}: eval fwd {time.time()-t0:.1f}s  softmax peak {torch.softmax(heat_e.reshape(B,24,-1),2).max().item():.3g}")
    out["eval_joints"] = j_e.numpy()
    out["eval_heat_l2_per_joint"] = heat_e.reshape(B, 24, -1).double().norm(dim=2).numpy()
    if full:
        out["eval_heat"] = heat_e.numpy()
        out["eval_refine"] = refine_e.numpy()
    else:
        out["eval_heat_sub"] = heat_e[:, :, ::8, ::8, ::8].numpy()
        out["eval_refine_sub"] = refine_e[:, :, ::8, ::8, ::8].numpy()
        out["eval_refine_l2"] = np.float64(refine_e.double().norm().item())

    if full:
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        heat, refine = model(meas)
        jl = L2JointLocationLoss(output_3d=True)(heat, joints, torch.ones_like(joints))
        vl = BCEDiceLoss()(refine.reshape(B, -1), vol.reshape(B, -1))
        loss = jl + vl
        opt.zero_grad()
        loss.backward()
        out["train_heat"] = heat.detach().numpy()
        out["train_refine"] = refine.detach().numpy()
        out["train_joints"] = softmax_integral_tensor(heat.detach(), 24, True, hm[0], hm[1], hm[2]).numpy()
        out["train_joint_loss"] = np.float64(jl.item())
        out["train_voxel_loss"] = np.float64(vl.item())
        named = dict(model.named_parameters())
        for k in E2E_PARAMS:
            out["grad_" + k] = named[k].grad.numpy().copy()
        opt.step()
        for k in E2E_PARAMS:
            out["adam1_" + k] = named[k].detach().numpy().copy()
        out["adam1_bn1_running_mean"] = model.state_dict()["pose_net.bn1.running_mean"].numpy()
    return out


def sec_e2e():
    save("e2e_T32_N32.npz", **_e2e(32, 32, 2, True))


def sec_e2e128():
    save("e2e_T128_N128.npz", **_e2e(128, 128, 1, False))


def sec_e2e512():
    """The benchmark cube (BASELINE metric: 128x128x512), eval forward of the reference, batch 1."""
    save("e2e_T512_N128.npz", **_e2e(512, 128, 1, False))


def sec_softargmax():
    """The reference's own known-answer demo (utils/criterion.py:420-437): -1000
    background, +1 at one voxel per joint => decode returns that voxel, loss ~ 0."""
    from utils.criterion import L2JointLocationLoss, softmax_integral_tensor

    B, J, D = 1, 24, 5
    inp = torch.zeros(B, J, D, D, D) - 1000
    for j in range(J):
        inp[0, j, 0, 0, 0] = 1 if j != 0 else -1000
    inp[0, 0, 1, 1, 1] = 1.0
    gt = torch.zeros(B, J, 3)
    gt[0, 0] = torch.tensor([1.0, 1.0, 1.0])
    gt = gt.reshape(B, J * 3)
    pred = softmax_integral_tensor(inp, J, True, D, D, D)
    loss = L2JointLocationLoss(output_3d=True)(inp, gt, torch.ones_like(gt))
    # a non-trivial random case too, anisotropic dims (W,H,D)=(6,5,4)
    g = torch.Generator().manual_seed(3)
    r = torch.randn(2, 24, 4, 5, 6, generator=g) * 3
    rp = softmax_integral_tensor(r, 24, True, 6, 5, 4)
    save("softargmax.npz", demo_pred=pred.numpy(), demo_loss=np.float64(loss.item()), rand_pred=rp.numpy())


def sec_specular():
    """material='specular' (models/feature_propagation.py:213-217: g^2 fall-off instead of g^4), forward + input gradient."""
    import models.feature_propagation as fp

    T, N, B = 32, 16, 2
    lct = fp.LCT(N, T, BIN_LEN[(T, N)], 2.0, material="specular")
    x = hpt.synthetic_meas(B, T, N, "uniform", seed=0).requires_grad_(True)
    y = lct(x, [0] * B, [T] * B)
    gy = hpt.synthetic_meas(B, T, N, "uniform", seed=100) - 0.5
    (y * gy).sum().backward()
    save("lct_specular.npz", y=y.detach().numpy(), gx=x.grad.numpy())
    print(f"  specular: |y| {y.norm().item():.4g} |gx| {x.grad.norm().item():.4g}")


def sec_bp():
    """mode 'bp' (models/feature_propagation.py:93-94,103-107,246-253) from the reference's runnable twin models/tflct.py
    (method='bp': conj-only inverse filter, ReplicationPad3d(2) + 5^3 Laplacian of Gaussian from utils/helper.py:13-32,
    first time slice zeroed).  tflct.lct fixes its time size at 128 whatever `crop` says (:19), so T = 128, N = 16.
    Forward + input gradient, and the 125 filter weights themselves."""
    import models.tflct as tf
    from utils.helper import filterLaplacian

    T, N, B = 128, 16, 2
    lct = tf.lct(spatial=N, crop=T, bin_len=0.04, wall_size=2.0, method="bp")
    lct.todev("cpu", 1)
    x = hpt.synthetic_meas(B, T, N, "uniform", seed=0).requires_grad_(True)
    y = lct(x, [0] * B, [T] * B)
    gy = hpt.synthetic_meas(B, T, N, "uniform", seed=100) - 0.5
    (y * gy).sum().backward()
    # the Wiener twin of the same class on the same input: pins that tflct's 'lct' branch equals feature_propagation's
    lct_w = tf.lct(spatial=N, crop=T, bin_len=0.04, wall_size=2.0, method="lct")
    lct_w.todev("cpu", 1)
    yw = lct_w(x.detach(), [0] * B, [T] * B)
    save("lct_bp.npz", y=y.detach().numpy(), gx=x.grad.numpy(), lapw=filterLaplacian().astype(np.float32), y_lct=yw.numpy())
    print(f"  bp: |y| {y.norm().item():.4g} |gx| {x.grad.norm().item():.4g} |y_lct| {yw.norm().item():.4g}")


def sec_visible():
    """VisibleNet (models/feature_propagation.py:289-312) on a volume without ties among its four largest values."""
    from models.feature_propagation import VisibleNet

    gen = torch.Generator().manual_seed(21)
    x = torch.randn(2, 3, 12, 9, 10, generator=gen)
    y = VisibleNet(basedim=3)(x)
    save("visible_net.npz", x=x.numpy(), y=y.numpy())
    print(f"  VisibleNet: {tuple(x.shape)} -> {tuple(y.shape)}")


E2E128_PARAMS = ["feature_extraction.weights", "feature_extraction.conv1.1.weight", "feature_extraction.conv1.3.tmp.4.bias",
                 "autoencoder.conv.double_conv.0.weight", "autoencoder.dec4.conv.double_conv.3.weight", "autoencoder.out.conv.bias",
                 "pose_net.conv1.weight", "pose_net.bn1.weight", "pose_net.layer1.0.conv2.weight", "pose_net.layer2.0.conv2.weight",
                 "pose_net.layer3.2.conv1.weight", "pose_net.layer4.1.conv3.weight", "pose_net.head.features.0.weight",
                 "pose_net.head.features.9.weight", "pose_net.head.features.9.bias"]


def sec_e2e128train(smooth=False):
    """smooth=True: the same step with hiddenpose_amd.testing's second filler (normalisation layers in front of a ReLU get
    gain 0.5 / bias +2: ReLU decisions far from rounding noise) -> e2e_T128_N128_train_smooth.npz, the golden whose
    gradients carry a 1e-3 bar.
    The reference's train step (utils/train_epoch.py:38-76) at its NATIVE shape 128^3, batch 2: losses, joints,
    sampled heat-maps, 15 named gradients (sampled + L2), post-Adam values and BatchNorm running statistics.  This is
    the well-conditioned size for gradient parity (at T = N = 32 layer4's BatchNorm normalises over 2 values)."""
    from models.NlosPose import NlosPose
    from utils.criterion import BCEDiceLoss, L2JointLocationLoss, softmax_integral_tensor

    T = N = 128
    B = 2
    cfg = ref_shims.make_cfg(T, N, BIN_LEN[(T, N)])
    model = NlosPose(cfg)
    hpt.fill_module(model, smooth=smooth)
    meas = hpt.synthetic_meas(B, T, N, "transient", seed=410)
    vol = hpt.synthetic_vol(B, T, N)
    joints = hpt.synthetic_joints(B, T // 2).reshape(B, -1)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    t0 = time.time()
    heat, refine = model(meas)
    jl = L2JointLocationLoss(output_3d=True)(heat, joints, torch.ones_like(joints))
    vl = BCEDiceLoss()(refine.reshape(B, -1), vol.reshape(B, -1))
    loss = jl + vl
    opt.zero_grad()
    loss.backward()
    print(f"  e2e128train: fwd+bwd {time.time()-t0:.1f}s  joint loss {jl.item():.6g}  voxel loss {vl.item():.6g}")
    out = {"joint_loss": np.float64(jl.item()), "voxel_loss": np.float64(vl.item()),
           "joints": softmax_integral_tensor(heat.detach(), 24, True, 64, 64, 64).numpy(),
           "heat_l2_per_joint": heat.detach().reshape(B, 24, -1).double().norm(dim=2).numpy(),
           "heat_sub": heat.detach()[:, :, ::8, ::8, ::8].numpy(), "refine_sub": refine.detach()[:, :, ::8, ::8, ::8].numpy(),
           "refine_l2": np.float64(refine.detach().double().norm().item())}
    named = dict(model.named_parameters())
    for k in E2E128_PARAMS:
        g = named[k].grad.numpy()
        out["gl2_" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        if g.size > 8192:
            idx = sample_idx(g.size, 4096, 17)
            out["gidx_" + k] = idx
            out["gs_" + k] = g.reshape(-1)[idx]
        else:
            out["g_" + k] = g.copy()
    opt.step()
    for k in E2E128_PARAMS:
        v = named[k].detach().numpy()
        out["adam1_" + k] = v.reshape(-1)[out["gidx_" + k]].copy() if ("gidx_" + k) in out else v.copy()
    sd = model.state_dict()
    out["bn1_running_mean"] = sd["pose_net.bn1.running_mean"].numpy()
    out["bn1_running_var"] = sd["pose_net.bn1.running_var"].numpy()
    out["l4_bn3_running_var"] = sd["pose_net.layer4.2.bn3.running_var"].numpy()

    # Conditioning of this step: the SAME reference step in float64.  |g32 - g64| / |g64| per parameter is what fp32
    # rounding alone does to each gradient in the reference itself (ReLU masks and batch statistics re-decided per
    # layer amplify last-bit differences); a kernel cannot be held to a tighter bar than the reference holds itself.
    del heat, refine, loss, opt
    model64 = NlosPose(cfg)
    hpt.fill_module(model64, smooth=smooth)
    model64 = model64.double().train()
    lctm = model64.feature_propagation.method
    for a in ("gridz_1xMx1x1_todev", "mtx_MxM_todev", "mtxi_MxM_todev", "invpsf_real_todev", "invpsf_imag_todev", "datapad_Dx2Tx2Hx2W"):
        setattr(lctm, a, getattr(lctm, a).double())
    t0 = time.time()
    h64, r64 = model64(meas.double())
    l64 = (L2JointLocationLoss(output_3d=True)(h64, joints.double(), torch.ones_like(joints).double())
           + BCEDiceLoss()(r64.reshape(B, -1), vol.double().reshape(B, -1)))
    l64.backward()
    print(f"  e2e128train float64: fwd+bwd {time.time()-t0:.1f}s  loss {l64.item():.9g}")
    named64 = dict(model64.named_parameters())
    for k in E2E128_PARAMS:
        g64 = named64[k].grad.numpy()
        g32 = named[k].grad.numpy().astype(np.float64)
        out["spread_" + k] = np.float64(np.linalg.norm(g32 - g64) / max(np.linalg.norm(g64), 1e-300))
        out["g64l2_" + k] = np.float64(np.linalg.norm(g64))
        if ("gidx_" + k) in out:
            out["gs64_" + k] = g64.reshape(-1)[out["gidx_" + k]]
        else:
            out["g64_" + k] = g64.copy()
    out["loss64"] = np.float64(l64.item())
    print("  fp32-vs-fp64 gradient spread of the reference: " + ", ".join(f"{k.split('.', 1)[1]} {float(out['spread_' + k]):.1e}" for k in E2E128_PARAMS))
    save("e2e_T128_N128_train_smooth.npz" if smooth else "e2e_T128_N128_train.npz", **out)


def _ckpt_wrap(model):
    """Stem (conv1 -> bn1 -> relu -> maxpool) and every Bottleneck of layer1 / layer2 of the reference's posenet under
    torch.utils.checkpoint (non-reentrant): the SAME reference modules are called; their forward is recomputed inside
    backward instead of being kept.  Train-mode BatchNorm recomputes identical batch statistics, so outputs and gradients are
    those of the plain step; only the running statistics receive a second momentum update (never stored from these runs)."""
    from torch.utils.checkpoint import checkpoint

    class _Ckpt(torch.nn.Module):
        def __init__(self, fn):
            super().__init__()
            self.fn = fn

        def forward(self, x):
            return checkpoint(self.fn, x, use_reentrant=False)

    model.train()          # before the stem modules leave the module tree below
    pn = model.pose_net
    conv1, bn1, relu, pool = pn.conv1, pn.bn1, pn.relu, pn.maxpool
    pn.conv1 = _Ckpt(lambda x: pool(relu(bn1(conv1(x)))))
    pn.bn1 = pn.relu = pn.maxpool = torch.nn.Identity()
    for layer in (pn.layer1, pn.layer2):
        for i in range(len(layer)):
            layer[i] = _Ckpt(layer[i])
    assert bn1.training and pn.layer1[0].fn.bn1.training


def _e2e512_step(B, dtype):
    """One train-mode forward + both losses + backward of the reference at 128 x 128 x 512, smooth filler."""
    import resource

    from models.NlosPose import NlosPose
    from utils.criterion import BCEDiceLoss, L2JointLocationLoss, softmax_integral_tensor

    T, N = 512, 128
    cfg = ref_shims.make_cfg(T, N, BIN_LEN[(T, N)])
    model = NlosPose(cfg)
    hpt.fill_module(model, smooth=True)
    if dtype == torch.float64:
        model = model.double()
        lctm = model.feature_propagation.method
        for a in ("gridz_1xMx1x1_todev", "mtx_MxM_todev", "mtxi_MxM_todev", "invpsf_real_todev", "invpsf_imag_todev", "datapad_Dx2Tx2Hx2W"):
            setattr(lctm, a, getattr(lctm, a).double())
    named = dict(model.named_parameters())
    _ckpt_wrap(model)
    meas = hpt.synthetic_meas(B, T, N, "transient", seed=410).to(dtype)
    vol = hpt.synthetic_vol(B, T, N).to(dtype)
    joints = hpt.synthetic_joints_box(B, (N // 2, N // 2, T // 2)).reshape(B, -1).to(dtype)
    t0 = time.time()
    heat, refine = model(meas)
    jl = L2JointLocationLoss(output_3d=True)(heat, joints, torch.ones_like(joints))
    vl = BCEDiceLoss()(refine.reshape(B, -1), vol.reshape(B, -1))
    (jl + vl).backward()
    print(f"  e2e512 step B={B} {dtype}: fwd+bwd {time.time()-t0:.1f}s  joint loss {jl.item():.9g}  voxel loss {vl.item():.9g}  "
          f"maxrss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/2**20:.1f} GiB", flush=True)
    hd = heat.detach()
    out = {"joint_loss": np.float64(jl.item()), "voxel_loss": np.float64(vl.item()),
           "joints": softmax_integral_tensor(hd, 24, True, N // 2, N // 2, T // 2).numpy(),
           "heat_l2_per_joint": hd.reshape(B, 24, -1).double().norm(dim=2).numpy(),
           "heat_sub": hd[:, :, ::8, ::8, ::8].numpy(), "refine_sub": refine.detach()[:, :, ::8, ::8, ::8].numpy(),
           "refine_l2": np.float64(refine.detach().double().norm().item())}
    grads = {k: named[k].grad.numpy().copy() for k in E2E128_PARAMS}
    return out, grads


def _store_grads(out, grads, prefix_l2, prefix_s, prefix_full):
    for k, g in grads.items():
        out[prefix_l2 + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        if g.size > 8192:
            idx = sample_idx(g.size, 4096, 17)
            out["gidx_" + k] = idx
            out[prefix_s + k] = g.reshape(-1)[idx]
        else:
            out[prefix_full + k] = g.copy()


def sec_e2e512train():
    """The reference's train step at the BENCHMARK cube (128 x 128 x 512, BASELINE configs[1]'s shape), batch 2, smooth
    filler -> e2e_T512_N128_train_smooth.npz: losses, joints, sampled heat-maps / refined volume, the 15 named gradients.
    The step's saved activations need ~80 GB as the reference runs it; this container has 64, hence _ckpt_wrap (27 GiB
    peak).  float32 only: the float64 twin of this batch needs > 51 GB for the stem BatchNorm's backward alone (three
    17 GB tensors) -- see e2e512train_b1 for the float64 yardstick at this volume."""
    out, grads = _e2e512_step(2, torch.float32)
    _store_grads(out, grads, "gl2_", "gs_", "g_")
    save("e2e_T512_N128_train_smooth.npz", **out)


def sec_e2e512train_repro():
    """How reproducible is the reference's OWN float32 step at batch 2?  The same step as e2e512train with 3 instead of 8
    intra-op threads (another partition of every reduction PyTorch parallelises): repro_* = |g(3 threads) - g(8 threads)| /
    |g(8 threads)| per named gradient, added to e2e_T512_N128_train_smooth.npz.  A float32 golden cannot pin a gradient more
    tightly than the golden's own arithmetic reproduces it."""
    path = os.path.join(HERE, "e2e_T512_N128_train_smooth.npz")
    old = dict(np.load(path))
    torch.set_num_threads(3)
    out, grads = _e2e512_step(2, torch.float32)
    torch.set_num_threads(8)
    for k, g in grads.items():
        ref = old["gs_" + k] if ("gs_" + k) in old else old["g_" + k]
        got = g.reshape(-1)[old["gidx_" + k]] if ("gidx_" + k) in old else g
        old["repro_" + k] = np.float64(np.linalg.norm(got.astype(np.float64) - ref) / max(np.linalg.norm(ref.astype(np.float64)), 1e-300))
    old["repro_joint_loss"] = np.float64(abs(float(out["joint_loss"]) / float(old["joint_loss"]) - 1))
    print("  float32 reproducibility of the reference at 512x128x128, B=2 (3 vs 8 threads): " +
          ", ".join(f"{k.split('.', 1)[1]} {float(old['repro_' + k]):.1e}" for k in E2E128_PARAMS))
    save("e2e_T512_N128_train_smooth.npz", **old)


def sec_e2e512train_b1():
    """The same step at batch 1 in float32 AND float64 -> e2e_T512_N128_train_smooth_b1.npz: the float64 gradients are the
    yardstick at the benchmark volume (train-mode BatchNorm over one sample's voxels is well defined: >= 2048 values per
    channel at layer4), spread_* = how far the reference's own float32 gradients lie from them."""
    out, g32 = _e2e512_step(1, torch.float32)
    _store_grads(out, g32, "gl2_", "gs_", "g_")
    o64, g64 = _e2e512_step(1, torch.float64)
    out["loss64"] = np.float64(o64["joint_loss"] + o64["voxel_loss"])
    out["joint_loss64"], out["voxel_loss64"] = o64["joint_loss"], o64["voxel_loss"]
    out["joints64"] = o64["joints"]
    out["heat_sub64"], out["refine_sub64"] = o64["heat_sub"].astype(np.float32), o64["refine_sub"].astype(np.float32)
    for k in E2E128_PARAMS:
        a, b = g32[k].astype(np.float64), g64[k]
        out["spread_" + k] = np.float64(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
        out["g64l2_" + k] = np.float64(np.linalg.norm(b))
        if ("gidx_" + k) in out:
            out["gs64_" + k] = b.reshape(-1)[out["gidx_" + k]]
        else:
            out["g64_" + k] = b.copy()
    print("  fp32-vs-fp64 gradient spread of the reference at 512x128x128, B=1: " +
          ", ".join(f"{k.split('.', 1)[1]} {float(out['spread_' + k]):.1e}" for k in E2E128_PARAMS))
    save("e2e_T512_N128_train_smooth_b1.npz", **out)


def _lean_lct(T, N, bin_len):
    """Memory-lean restatement of LCT.forward / its adjoint for sizes where the reference's own constructor does not
    fit this container (its (2N,2N,2M) meshgrids + complex128 spectrum need > 60 GB at 1024 x 256 x 256): the PSF comes
    from the pinned oracle (bit-exact with the reference's, tests/test_oracle_golden.py), spectra are kept as
    half-spectra (the padded volume is real and invpsf Hermitian, so R2C/C2R is the same operator).  Checked against
    the reference's LCT at (T,N) = (64,32) right here before it is used."""
    import scipy.fft as sfft

    from oracle import nlospose_oracle as O

    slope = (2.0 / 2.0) / (T * bin_len)
    psf = O.define_psf(N, T, slope)                           # (2T,2N,2N) float32
    f = sfft.rfftn(psf.astype(np.float64), workers=8)         # complex128 half spectrum
    del psf
    inv = np.conjugate(f) / (1 / 1e-1 + f.real ** 2 + f.imag ** 2)
    del f
    inv = inv.astype(np.complex64)
    mtx = torch.from_numpy(O.resampling_operator(T)).to_sparse_csr()
    mtxi = torch.from_numpy(np.ascontiguousarray(O.resampling_operator(T).T)).to_sparse_csr()
    g4 = (torch.arange(T, dtype=torch.float32) / (T - 1)).view(T, 1) ** 4

    def run(x, adjoint):                                       # x (T,N,N) float32 tensor
        v = x.reshape(T, N * N)
        if not adjoint:
            v = v * g4
        v = torch.sparse.mm(mtx, v).reshape(T, N, N).numpy()
        pad = np.zeros((2 * T, 2 * N, 2 * N), np.float32)
        pad[:T, :N, :N] = v
        fr = sfft.rfftn(pad, workers=8)
        del pad
        fr *= np.conjugate(inv) if adjoint else inv
        re = sfft.irfftn(fr, s=(2 * T, 2 * N, 2 * N), workers=8)[:T, :N, :N]
        del fr
        o = torch.sparse.mm(mtxi, torch.from_numpy(np.ascontiguousarray(re, dtype=np.float32)).reshape(T, N * N))
        if adjoint:
            o = o * g4
        return o.reshape(T, N, N)

    class Lean(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return torch.stack([run(v, False) for v in x.reshape(-1, T, N, N)]).view_as(x)

        @staticmethod
        def backward(ctx, gy):
            return torch.stack([run(v, True) for v in gy.reshape(-1, T, N, N)]).view_as(gy)

    return Lean.apply


def sec_highres():
    """BASELINE configs[3]: 256 x 256 x 1024, FeatureExtraction -> LCT -> normalize_feature -> UNet3d, batch 1,
    forward and gradients.  FeatureExtraction, normalize_feature and UNet3d are the REFERENCE's modules; the LCT is
    `_lean_lct` (see there; validated against the reference's LCT below)."""
    import models.feature_propagation as fp
    from models.feature_extraction import FeatureExtraction
    from models.feature_propagation import normalize_feature
    from unet.unet3d import UNet3d

    # 1. the lean LCT against the reference's own, forward and adjoint, at a size the reference can build
    Ts, Ns = 64, 32
    lean = _lean_lct(Ts, Ns, BIN_LEN[(Ts, Ns)])
    ref = fp.LCT(Ns, Ts, BIN_LEN[(Ts, Ns)], 2.0)
    xa = hpt.synthetic_meas(1, Ts, Ns, "uniform", seed=3).requires_grad_(True)
    xb = xa.detach().clone().requires_grad_(True)
    ga = hpt.synthetic_meas(1, Ts, Ns, "uniform", seed=4) - 0.5
    ya, yb = ref(xa, [0], [Ts]), lean(xb)
    (ya * ga).sum().backward()
    (yb * ga).sum().backward()
    e1 = ((ya - yb).norm() / ya.norm()).item()
    e2 = ((xa.grad - xb.grad).norm() / xa.grad.norm()).item()
    print(f"  lean LCT vs reference LCT at {Ts}x{Ns}x{Ns}: forward rel-L2 {e1:.2e}, adjoint rel-L2 {e2:.2e}")
    assert e1 < 2e-6 and e2 < 2e-6

    # 2. the full-size pipeline
    T, N, B = 1024, 256, 1
    t0 = time.time()
    lct = _lean_lct(T, N, 5.12 / T)
    print(f"  highres: lean constants {time.time()-t0:.1f}s")
    fe = FeatureExtraction(basedim=1, in_channels=1, stride=1)
    hpt.fill_module(fe, "feature_extraction.")
    un = UNet3d(in_channels=1, n_channels=4)
    hpt.fill_module(un, "autoencoder.")
    meas = hpt.synthetic_meas(B, T, N, "transient", seed=410).requires_grad_(True)
    t0 = time.time()
    a = fe(meas)
    l = lct(a)
    f = normalize_feature(l)
    r = un(f)
    print(f"  highres: forward {time.time()-t0:.1f}s")
    # a smooth, seeded cotangent (no 268 MB random tensor to regenerate in the test): loss = mean(r^2) + mean(f)
    loss = r.square().mean() + f.mean()
    t0 = time.time()
    loss.backward()
    print(f"  highres: backward {time.time()-t0:.1f}s  loss {loss.item():.6g}")
    out = {"loss": np.float64(loss.item())}
    idx = sample_idx(T * N * N, 2048, 23)
    out["idx"] = idx
    for tag, t in (("fe", a), ("lct", l), ("feature", f), ("refine", r), ("gmeas", meas.grad)):
        v = t.detach().numpy().reshape(-1)
        out[tag + "_s"] = v[idx]
        out[tag + "_l2"] = np.float64(np.sqrt((v.astype(np.float64) ** 2).sum()))
    for k, p_ in list(fe.named_parameters()):
        out["g_fe." + k] = p_.grad.numpy().copy()
    named = dict(un.named_parameters())
    for k in ["conv.double_conv.0.weight", "conv.double_conv.1.weight", "conv.double_conv.3.weight", "enc2.encoder.1.double_conv.0.weight",
              "enc4.encoder.1.double_conv.3.weight", "dec1.conv.double_conv.0.weight", "dec4.conv.double_conv.0.weight",
              "dec4.conv.double_conv.4.bias", "out.conv.weight", "out.conv.bias"]:
        out["g_un." + k] = named[k].grad.numpy().copy()
    save("highres_T1024_N256.npz", **out)


SFORMER_CFGS = {
    "small": dict(dim=64, num_frames=4, num_joints=24, image_size=32, patch_size=8, channels=1, depth=2, heads=4,
                  dim_head=16, out_dim=128),
    "mid": dict(dim=128, num_frames=3, num_joints=24, image_size=64, patch_size=4, channels=1, depth=2, heads=4,
                dim_head=32, out_dim=512),
}


def sec_sformer():
    """NlosPoseSformer (models/NlosPoseSformer.py), the orphan RoPE transformer head of BASELINE config 5."""
    import contextlib
    import io

    from models.NlosPoseSformer import NlosPoseSformer

    out = {}
    for tag, kw in SFORMER_CFGS.items():
        with contextlib.redirect_stdout(io.StringIO()):
            m = NlosPoseSformer(**kw)
        hpt.fill_module(m, "sformer.")
        m.eval()
        g = torch.Generator().manual_seed(77)
        video = torch.rand(2, kw["num_frames"], kw["channels"], kw["image_size"], kw["image_size"], generator=g)
        with torch.no_grad():
            y = m(video)
        out[tag + "_y"] = y.numpy()
        print(f"  sformer {tag}: out {tuple(y.shape)} std {y.std().item():.3g}")
    save("sformer_io.npz", **out)


TIMESFORMER_CFGS = {
    "plain": dict(dim=64, num_frames=4, num_classes=10, image_size=32, patch_size=8, channels=1, depth=2, heads=4, dim_head=16),
    "shift": dict(dim=96, num_frames=3, num_classes=10, image_size=32, patch_size=4, channels=2, depth=2, heads=2, dim_head=32,
                  shift_tokens=True),
}
TOKENPOSE_CFGS = {
    "sinefull": dict(feature_size=[16, 16], patch_size=[4, 4], num_keypoints=6, dim=48, depth=2, heads=2, mlp_dim=96,
                     heatmap_dim=64, heatmap_size=[8, 8], channels=4, pos_embedding_type="sine-full", hidden_heatmap_dim=64),
    "learnable": dict(feature_size=[16, 24], patch_size=[4, 4], num_keypoints=5, dim=64, depth=1, heads=4, mlp_dim=128,
                      heatmap_dim=48, heatmap_size=[8, 6], channels=3, pos_embedding_type="learnable", hidden_heatmap_dim=64),
}


def sec_xformers():
    """The orphan transformer heads of SURVEY 8(f) rank 1: TimeSformer (models/transformer.py:152-257, with and without
    token shift) and TokenPose_L_base (models/tokenpose.py:66-227)."""
    import contextlib
    import io
    import json

    from models.tokenpose import TokenPose_L_base
    from models.transformer import TimeSformer

    out, schema = {}, {}
    for tag, kw in TIMESFORMER_CFGS.items():
        m = TimeSformer(**kw)
        hpt.fill_module(m, "timesformer.")
        with torch.no_grad():
            m.cls_token.copy_(hpt.fill_value("timesformer.cls_token", m.cls_token.shape))
        m.eval()
        g = torch.Generator().manual_seed(78)
        video = torch.rand(2, kw["num_frames"], kw["channels"], kw["image_size"], kw["image_size"], generator=g)
        with torch.no_grad():
            y = m(video)
        out["ts_" + tag + "_y"] = y.numpy()
        schema["ts_" + tag] = {k: list(v.shape) for k, v in m.state_dict().items()}
        print(f"  TimeSformer {tag}: out {tuple(y.shape)} std {y.std().item():.3g}")
    for tag, kw in TOKENPOSE_CFGS.items():
        with contextlib.redirect_stdout(io.StringIO()):
            m = TokenPose_L_base(**kw)
        hpt.fill_module(m, "tokenpose.")
        m.eval()
        g = torch.Generator().manual_seed(79)
        feat = torch.rand(2, kw["channels"], kw["feature_size"][0], kw["feature_size"][1], generator=g)
        with torch.no_grad():
            y = m(feat)
        out["tp_" + tag + "_y"] = y.numpy()
        schema["tp_" + tag] = {k: list(v.shape) for k, v in m.state_dict().items()}
        print(f"  TokenPose {tag}: out {tuple(y.shape)} std {y.std().item():.3g}")
    save("xformers_io.npz", **out)
    with open(os.path.join(HERE, "xformers_schema.json"), "w") as f:
        json.dump(schema, f, indent=0)


def sec_schema():
    """state_dict key names and shapes of the reference NlosPose (checkpoint contract)."""
    import json

    from models.NlosPose import NlosPose

    model = NlosPose(ref_shims.make_cfg(32, 32, BIN_LEN[(32, 32)]))
    sd = model.state_dict()
    schema = {k: list(v.shape) for k, v in sd.items()}
    with open(os.path.join(HERE, "state_dict_schema.json"), "w") as f:
        json.dump(schema, f, indent=0, sort_keys=False)
    print(f"  wrote state_dict_schema.json: {len(schema)} tensors, "
          f"{sum(int(np.prod(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k)} parameters")


def sec_lctwin():
    """LCT.forward with partial time windows (models/feature_propagation.py:186-200): samples of equal length
    placed at different offsets inside the zero-padded time axis."""
    import models.feature_propagation as fp

    T, N = 32, 16
    lct = fp.LCT(N, T, BIN_LEN[(T, N)], 2.0)
    g = torch.Generator().manual_seed(12)
    x = torch.rand(2, 1, 24, N, N, generator=g, requires_grad=True)
    gy = torch.rand(2, 1, T, N, N, generator=g)
    tbes, tens = [2, 5], [26, 29]
    y = lct(x, tbes, tens)
    (y * gy).sum().backward()
    save("lct_window.npz", x=x.detach().numpy(), gy=gy.numpy(), tbes=np.array(tbes), tens=np.array(tens), y=y.detach().numpy(),
         gx=x.grad.numpy())
    print(f"  window: y {tuple(y.shape)} |y| {y.norm().item():.4g} |gx| {x.grad.norm().item():.4g}")


def sec_ingest():
    """utils/loadrealdata.py:6-15 and NlosPoseDataset.__getitem__ (utils/nlos_pose_dataloader.py:71-144).
    cv2 is absent from this image: imread / cvtColor are the oracle's restatement of OpenCV's RGBE reader and
    float BGR2GRAY, so ingest_getitem.npz pins every other line of __getitem__ (see oracle/ingest_oracle.py)."""
    import tempfile

    from scipy.io import savemat

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import ingest_oracle as io

    from utils.loadrealdata import load_realdata

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        g = np.random.Generator(np.random.PCG64(5))
        for tag, dt, cnt in (("f32", np.float32, 1), ("f64", np.float64, 2)):
            data = g.random((8, 12, 32)).astype(dt)  # (h, w, t)
            path = os.path.join(tmp, f"real_{tag}.mat")
            savemat(path, {"data_new": data})
            y = load_realdata(path, downsample_cnt=cnt)
            out[f"real_{tag}_in"] = data
            out[f"real_{tag}_out"] = y.numpy()
            out[f"real_{tag}_cnt"] = cnt
            print(f"  load_realdata {tag}: {data.shape} -> {tuple(y.shape)} {y.dtype}")
        save("ingest_realdata.npz", **out)

        # dataset tree: <root>/<pose>/<phase>/{meas,vol,joints}/<id>.{hdr,mat,joints}
        cv2 = sys.modules["cv2"]
        cv2.imread = lambda path, flag: io.rgbe_to_bgr_float(io.rgbe_read(open(path, "rb").read()))
        cv2.cvtColor = lambda img, code: io.bgr2gray_f32(img)
        cv2.COLOR_BGR2GRAY = 6
        from utils.nlos_pose_dataloader import NlosPoseDataset

        H = W = 8
        root = os.path.join(tmp, "data")
        for sub in ("meas", "vol", "joints"):
            os.makedirs(os.path.join(root, "pose0", "train", sub))
        rgbe = hpt.synthetic_rgbe(600, H, W, seed=7)
        open(os.path.join(root, "pose0", "train", "meas", "person_3.hdr"), "wb").write(io.rgbe_write(rgbe, rle=True))
        vol = (g.random((8, 8, 8)) < 0.3).astype(np.float64)
        savemat(os.path.join(root, "pose0", "train", "vol", "person_3.mat"), {"vol": vol})
        joints = g.random((24, 3)) - 0.5
        np.savetxt(os.path.join(root, "pose0", "train", "joints", "person_3.joints"), joints)
        cfg = ref_shims._AttrDict()
        cfg.DATASET = ref_shims._AttrDict(VOL_SIZE=[256, 256, 256], DAWNSAMPLE_CNT=1, PHASE="train")
        cfg.MODEL = ref_shims._AttrDict(HEATMAP_SIZE=[64, 64, 64])
        ds = NlosPoseDataset(cfg, root)
        meas, v, j, pid = ds[0]
        print(f"  __getitem__: meas {meas.shape} {meas.dtype}, vol {v.shape} {v.dtype}, joints {j.shape}, id {pid}")
        save("ingest_getitem.npz", vol_in=vol, joints_in=np.loadtxt(os.path.join(root, "pose0", "train", "joints", "person_3.joints")),
             meas=meas, vol=v, joints=j, person_id=np.array(pid))


SECTIONS = {"lctwin": sec_lctwin, "ingest": sec_ingest, "sformer": sec_sformer, "schema": sec_schema, "consts": sec_consts, "lct": sec_lct, "parts": sec_parts, "posenet": sec_posenet,
            "e2e": sec_e2e, "e2e128": sec_e2e128, "e2e512": sec_e2e512, "softargmax": sec_softargmax,
            "specular": sec_specular, "bp": sec_bp, "consts_hr": sec_consts_hr, "visible": sec_visible, "xformers": sec_xformers, "e2e128train": sec_e2e128train, "e2e128train_smooth": lambda: sec_e2e128train(True), "highres": sec_highres,
            "e2e512train": sec_e2e512train, "e2e512train_b1": sec_e2e512train_b1,
            "e2e512train_repro": sec_e2e512train_repro}

if __name__ == "__main__":
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        print(f"[{s}]")
        t0 = time.time()
        SECTIONS[s]()
        print(f"[{s}] done in {time.time()-t0:.1f}s")
