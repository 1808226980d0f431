"""Ingest row: the oracle against goldens captured from the reference's own loaders, the
library's host-side Radiance decoder against the oracle (CPU, no device), and the device path bit-exactly
against the oracle (GPU)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

from hiddenpose_amd import testing as hpt

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ingest_oracle as io  # noqa: E402


def test_oracle_load_realdata_matches_reference_golden(golden):
    g = golden("ingest_realdata.npz")
    for tag in ("f32", "f64"):
        y = io.load_realdata(g[f"real_{tag}_in"], int(g[f"real_{tag}_cnt"]))
        assert y.dtype == g[f"real_{tag}_out"].dtype
        assert np.array_equal(y, g[f"real_{tag}_out"])


def test_oracle_getitem_matches_reference_golden(golden):
    g = golden("ingest_getitem.npz")
    rgbe = hpt.synthetic_rgbe(600, 8, 8, seed=7)
    for rle in (True, False):
        meas = io.meas_from_hdr(io.rgbe_write(rgbe, rle=rle), 600, 512, 1)
        assert np.array_equal(meas[None], g["meas"])
    assert np.array_equal(io.vol_pyramid(g["vol_in"], 1)[None], g["vol"])
    assert np.array_equal(io.remap_joints(g["joints_in"], 256, 64), g["joints"])


def test_oracle_rejects_dark_file():
    rgbe = np.zeros((600 * 8, 8, 4), np.uint8)
    with pytest.raises(ValueError, match="wrong Meas File"):
        io.meas_from_hdr(io.rgbe_write(rgbe, rle=True))


@pytest.mark.parametrize("rle", [True, False])
def test_host_rgbe_decoder_matches_oracle(rle, tmp_path):
    from hiddenpose_amd.nlos_pose_dataloader import decode_hdr

    rgbe = hpt.synthetic_rgbe(40, 8, 24, seed=11)
    path = tmp_path / "a.hdr"
    path.write_bytes(io.rgbe_write(rgbe, rle=rle))
    got = decode_hdr(str(path))
    assert np.array_equal(got, rgbe)
    assert np.array_equal(io.rgbe_read(path.read_bytes()), rgbe)


def test_host_rgbe_decoder_errors(tmp_path):
    from hiddenpose_amd import _lib
    from hiddenpose_amd.nlos_pose_dataloader import decode_hdr

    bad = tmp_path / "bad.hdr"
    bad.write_bytes(b"P6\n1 1\n255\n\0\0\0")
    with pytest.raises(_lib.HiddenPoseHipError, match="signature"):
        decode_hdr(str(bad))
    trunc = tmp_path / "trunc.hdr"
    trunc.write_bytes(io.rgbe_write(hpt.synthetic_rgbe(4, 8, 16, seed=1), rle=True)[:-7])
    with pytest.raises(_lib.HiddenPoseHipError):
        decode_hdr(str(trunc))


def test_device_ops_refuse_cpu_tensors():
    from hiddenpose_amd import _lib
    from hiddenpose_amd.loadrealdata import realdata_to_meas
    from hiddenpose_amd.nlos_pose_dataloader import box_pyramid, rgbe_to_meas

    with pytest.raises(_lib.HiddenPoseHipError):
        rgbe_to_meas(torch.zeros(600 * 2, 2, 4, dtype=torch.uint8), 1)
    with pytest.raises(_lib.HiddenPoseHipError):
        box_pyramid(torch.zeros(2, 2, 2), 1)
    with pytest.raises(_lib.HiddenPoseHipError):
        realdata_to_meas(torch.zeros(2, 2, 2), 1)


# ---------------------------------------------------------------- device
@pytest.mark.gpu
@pytest.mark.parametrize("cnt", [0, 1, 2])
def test_rgbe_to_meas_bit_exact(cnt):
    from hiddenpose_amd.nlos_pose_dataloader import rgbe_to_meas

    rgbe = hpt.synthetic_rgbe(600, 8, 16, seed=3 + cnt)
    want = io.meas_from_bgr(io.rgbe_to_bgr_float(rgbe), 600, 512, cnt)
    got = rgbe_to_meas(torch.from_numpy(rgbe).cuda(), cnt)
    assert got.shape == want.shape
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_dataset_getitem_vs_reference_golden(golden, tmp_path):
    """The whole Dataset.__getitem__ on synthetic files against the reference's own output."""
    from scipy.io import savemat

    from hiddenpose_amd.config import make_cfg
    from hiddenpose_amd.nlos_pose_dataloader import NlosPoseDataset

    g = golden("ingest_getitem.npz")
    base = tmp_path / "pose0" / "train"
    for sub in ("meas", "vol", "joints"):
        (base / sub).mkdir(parents=True)
    (base / "meas" / "person_3.hdr").write_bytes(io.rgbe_write(hpt.synthetic_rgbe(600, 8, 8, seed=7), rle=True))
    (base / "meas" / "dark_9.hdr").write_bytes(io.rgbe_write(np.zeros((4800, 8, 4), np.uint8), rle=True))
    for stem in ("person_3", "dark_9"):
        savemat(str(base / "vol" / f"{stem}.mat"), {"vol": g["vol_in"]})
        np.savetxt(str(base / "joints" / f"{stem}.joints"), g["joints_in"])
    cfg = make_cfg(128, 128)
    assert cfg.DATASET.VOL_SIZE == [256, 256, 256] and cfg.DATASET.PHASE == "train" and cfg.DATASET.DAWNSAMPLE_CNT == 1
    ds = NlosPoseDataset(cfg, str(tmp_path))
    assert len(ds) == 2
    order = [os.path.basename(f) for f in ds.measFiles]
    # make the readable sample index 0 (the reference's fallback target), whatever os.listdir returned
    if order[0] != "person_3.hdr":
        for lst in (ds.measFiles, ds.volFiles, ds.jointsFiles):
            lst.reverse()
    meas, vol, joints, pid = ds[0]
    assert pid == "person_3" and meas.is_cuda and vol.is_cuda
    assert np.array_equal(meas.cpu().numpy(), g["meas"])
    assert np.array_equal(vol.cpu().numpy(), g["vol"])
    assert np.array_equal(joints, g["joints"])
    meas2, _, _, pid2 = ds[1]  # all-dark file -> replaced by sample 0, as at :84-104
    assert pid2 == "person_3" and torch.equal(meas2, meas)
    assert ds.wrongMeasFiles and ds.wrongMeasFiles[0].endswith("dark_9.hdr")


@pytest.mark.gpu
def test_load_realdata_vs_reference_golden(golden, tmp_path):
    from scipy.io import savemat

    from hiddenpose_amd.loadrealdata import load_realdata

    g = golden("ingest_realdata.npz")
    path = str(tmp_path / "real.mat")
    savemat(path, {"data_new": g["real_f32_in"]})
    y = load_realdata(path, downsample_cnt=int(g["real_f32_cnt"]))
    assert np.array_equal(y.cpu().numpy(), g["real_f32_out"])


@pytest.mark.gpu
def test_full_size_ingest_properties():
    """600 x 256 x 256 (the dataset's real size): constant image -> constant 1 volume; range and shape."""
    from hiddenpose_amd.nlos_pose_dataloader import rgbe_to_meas

    rgbe = torch.zeros(600 * 256, 256, 4, dtype=torch.uint8, device="cuda")
    rgbe[..., 0], rgbe[..., 1], rgbe[..., 2], rgbe[..., 3] = 200, 100, 50, 128
    meas = rgbe_to_meas(rgbe, 1)
    assert meas.shape == (128, 128, 128)
    assert torch.all(meas == 1.0)
    rnd = torch.randint(0, 256, (600 * 256, 256, 4), dtype=torch.uint8, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    rnd[..., 3] = rnd[..., 3] % 8 + 124
    m = rgbe_to_meas(rnd, 1)
    assert 0.0 <= float(m.min()) and float(m.max()) <= 1.0 and torch.isfinite(m).all()


# ---------------------------------------------------------------- noise variant (utils/nlos_pose_dataloader_noise.py)
def test_oracle_noise_path_properties():
    """Oracle of the noise dataset's __getitem__ arithmetic (:86-118): blur-only leg is float32 and within [0, 1] with
    maximum 1 before the pyramid; the Poisson leg goes through int64 counts and float64 division; a constant image stays
    constant through the blur (replicate border, taps sum to 1)."""
    rgbe = hpt.synthetic_rgbe(600, 8, 8, seed=5)
    bgr = io.rgbe_to_bgr_float(rgbe)
    blur = io.meas_from_bgr_noise(bgr, None, 600, 512, 1)
    assert blur.dtype == np.float32 and blur.shape == (128, 4, 4) and 0.0 <= blur.min() and blur.max() <= 1.0
    gray = io.bgr2gray_f32(bgr)
    scaled = (gray * np.float32(40.0 / gray.max())).astype(np.float32)   # means large enough for non-trivial counts
    counts = io.addnoise_dataset(scaled, np.random.default_rng(0))
    assert counts.dtype == np.int64 and counts.shape == gray.shape and counts.min() >= 0
    b = io.addnoise_dataset(scaled, None)
    assert abs(counts.mean() - b.mean()) < 0.02 * b.mean()
    const = io.addnoise_dataset(np.full((64, 8), 3.25, np.float32), None)
    assert np.allclose(const, 3.25, rtol=2e-6)
    with pytest.raises(ValueError, match="wrong Meas File"):
        io.meas_from_bgr_noise(np.zeros((4800, 8, 3), np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("cnt", [0, 1, 2])
def test_noisy_ingest_blur_leg_vs_oracle(cnt):
    """gray of the raw image -> blur -> / max -> crop -> pyramid on the device against the oracle (Poisson off: the
    deterministic part of the noise path).  The gray image and the pyramid are exact float32; the blur is an 87-tap sum in
    another order: 2e-6."""
    from hiddenpose_amd.nlos_pose_dataloader_noise import rgbe_to_noisy_meas

    rgbe = hpt.synthetic_rgbe(600, 8, 16, seed=21 + cnt)
    want = io.meas_from_bgr_noise(io.rgbe_to_bgr_float(rgbe), None, 600, 512, cnt)
    got = rgbe_to_noisy_meas(torch.from_numpy(rgbe).cuda(), cnt, seed=1, poisson=False).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.gpu
def test_noisy_ingest_poisson_leg_statistics_and_exact_normalisation():
    """With the Poisson draw on: (1) the device's counts fed to the ORACLE's '/ max', crop and pyramid (float64, cast at the
    end) reproduce the device's measurement bit for bit -- everything after the draw is exact; (2) the draw itself is
    statistical: integer, non-negative, reproducible per seed, mean within 1 % of the blurred image's mean."""
    from hiddenpose_amd import _lib
    from hiddenpose_amd import hip_ops as ops
    from hiddenpose_amd.nlos_pose_dataloader_noise import rgbe_to_noisy_meas

    rgbe = hpt.synthetic_rgbe(600, 8, 16, seed=33)
    rgbe[..., 3] = rgbe[..., 3] % 4 + 138          # decoded values of order 10..1000: counts well above 0 / 1
    dev = torch.from_numpy(rgbe).cuda()
    got = rgbe_to_noisy_meas(dev, 1, seed=77, poisson=True)
    again = rgbe_to_noisy_meas(dev, 1, seed=77, poisson=True)
    other = rgbe_to_noisy_meas(dev, 1, seed=78, poisson=True)
    assert torch.equal(got, again) and not torch.equal(got, other)
    # the same counts through the oracle's tail
    L = _lib.lib()
    gray = torch.empty(4800, 16, device="cuda")
    mx = torch.empty(2, device="cuda")
    _lib.check(L.hp_ingest_rgbe_to_gray(dev.data_ptr(), 4800 * 16, gray.data_ptr(), mx.data_ptr(), _lib.current_stream_handle(gray.device)),
               "hp_ingest_rgbe_to_gray")
    assert np.array_equal(gray.cpu().numpy(), io.bgr2gray_f32(io.rgbe_to_bgr_float(rgbe)))
    counts = ops.add_noise(gray, 10.61, seed=77, poisson=True).cpu().numpy()
    assert np.all(counts >= 0) and np.array_equal(counts, np.round(counts))
    blur = ops.add_noise(gray, 10.61, seed=77, poisson=False).cpu().numpy()
    assert abs(counts.mean() - blur.mean()) < 0.01 * blur.mean()
    m = counts.astype(np.int64)
    m = m / np.max(m)
    m = m.reshape(600, -1, 16)[:512]
    m = (m[::2] + m[1::2]) / 2
    m = io.box_round(m)
    assert np.array_equal(got.cpu().numpy(), m.astype(np.float32))


@pytest.mark.gpu
def test_noise_dataset_getitem(golden, tmp_path):
    """NlosPoseDataset of nlos_pose_dataloader_noise: add_noise=False is the base dataset bit for bit (reference golden);
    add_noise=True applies addnoise_dataset in the reference's position (gray -> noise -> / max) -- checked with the Poisson
    draw off against the oracle, and with it on for shape / range / reproducibility; an all-dark file falls back to sample 0."""
    from scipy.io import savemat

    from hiddenpose_amd.config import make_cfg
    from hiddenpose_amd.nlos_pose_dataloader import PrefetchingLoader
    from hiddenpose_amd.nlos_pose_dataloader_noise import NlosPoseDataset

    g = golden("ingest_getitem.npz")
    base = tmp_path / "pose0" / "train"
    for sub in ("meas", "vol", "joints"):
        (base / sub).mkdir(parents=True)
    rgbe = hpt.synthetic_rgbe(600, 8, 8, seed=7)
    (base / "meas" / "person_3.hdr").write_bytes(io.rgbe_write(rgbe, rle=True))
    (base / "meas" / "dark_9.hdr").write_bytes(io.rgbe_write(np.zeros((4800, 8, 4), np.uint8), rle=True))
    for stem in ("person_3", "dark_9"):
        savemat(str(base / "vol" / f"{stem}.mat"), {"vol": g["vol_in"]})
        np.savetxt(str(base / "joints" / f"{stem}.joints"), g["joints_in"])
    cfg = make_cfg(128, 128)

    def make(**kw):
        ds = NlosPoseDataset(cfg, str(tmp_path), **kw)
        if os.path.basename(ds.measFiles[0]) != "person_3.hdr":
            for lst in (ds.measFiles, ds.volFiles, ds.jointsFiles):
                lst.reverse()
        return ds

    clean = make(add_noise=False)
    meas, vol, joints, pid = clean[0]
    assert pid == "person_3" and np.array_equal(meas.cpu().numpy(), g["meas"]) and np.array_equal(vol.cpu().numpy(), g["vol"])
    blur = make(add_noise=True, poisson=False)
    mb, vb, jb, _ = blur[0]
    want = io.meas_from_bgr_noise(io.rgbe_to_bgr_float(rgbe), None, 600, 512, 1)
    assert mb.shape == meas.shape and np.abs(mb.cpu().numpy()[0] - want).max() <= 2e-6 * want.max()
    assert torch.equal(vb, vol) and np.array_equal(jb, joints)
    assert not torch.equal(mb, meas)                       # the noise path really ran
    noisy = make(add_noise=True, noise_seed=5)
    mn, _, _, _ = noisy[0]
    mn2, _, _, _ = noisy[0]
    assert mn.shape == meas.shape and torch.equal(mn, mn2) and 0.0 <= float(mn.min()) and float(mn.max()) <= 1.0
    md, _, _, pid_d = noisy[1]                             # dark file -> sample 0 (:96-104)
    assert pid_d == "person_3" and noisy.wrongMeasFiles and noisy.wrongMeasFiles[0].endswith("dark_9.hdr")
    # the prefetching loader drives the same two stages
    batches = list(PrefetchingLoader(blur, 2, shuffle=False, workers=2))
    assert len(batches) == 1 and torch.equal(batches[0][0][0], mb)
