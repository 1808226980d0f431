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
