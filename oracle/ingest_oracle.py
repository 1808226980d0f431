"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy) of the reference's per-sample ingest:
utils/nlos_pose_dataloader.py:71-144 (NlosPoseDataset.__getitem__) and utils/loadrealdata.py:6-15.
Imported by tests/ only; the product path (hiddenpose_amd/nlos_pose_dataloader.py -> C ABI -> HIP) never
touches this module.

Pinning status
* box pyramid, time-pair average, 'h w t -> t w h', crop, normalisation order, joint remap:
  **pinned** by goldens captured from the reference itself (tests/golden/ingest_realdata.npz from
  utils.loadrealdata.load_realdata; tests/golden/ingest_getitem.npz from NlosPoseDataset.__getitem__).
* `cv2.imread(path, -1)` on a Radiance .hdr and `cv2.cvtColor(..., COLOR_BGR2GRAY)` on float32
  (nlos_pose_dataloader.py:74,83): **parity unpinned**.  OpenCV (README.md:27 `pip install opencv-python`,
  no version pinned) is absent from this image and from /root/reference, so `rgbe_read`,
  `rgbe_to_bgr_float` and `bgr2gray_f32` restate its published behaviour: the Radiance RGBE container
  (Greg Ward's format: "#?RADIANCE" header, FORMAT=32-bit_rle_rgbe, "-Y H +X W", new-style per-channel
  RLE with a 2,2,hi,lo scanline prefix), OpenCV's rgbe2float (value = mantissa * 2^(e-136), no +0.5,
  e = 0 -> 0, channels returned as B,G,R) and the documented gray weights 0.114 B + 0.587 G + 0.299 R.
  The ingest_getitem golden was captured with these two functions standing in for cv2, so it pins every
  line of __getitem__ except those two calls.
"""
from __future__ import annotations

import numpy as np


# ---------------------------------------------------------------- Radiance .hdr container (cv2.imread stand-in)
def rgbe_read(buf: bytes) -> np.ndarray:
    """File bytes -> (H, W, 4) uint8 R,G,B,E."""
    pos = 0

    def line():
        nonlocal pos
        end = buf.index(b"\n", pos)
        ln = buf[pos:end]
        pos = end + 1
        return ln

    sig = line()
    if not sig.startswith(b"#?"):
        raise ValueError("rgbe: missing '#?' signature line")
    fmt = False
    while True:
        ln = line()
        if not ln:
            break
        fmt |= ln == b"FORMAT=32-bit_rle_rgbe"
    if not fmt:
        raise ValueError("rgbe: FORMAT line not found")
    parts = line().split()
    if len(parts) != 4 or parts[0] != b"-Y" or parts[2] != b"+X":
        raise ValueError("rgbe: only '-Y H +X W' is supported")
    H, W = int(parts[1]), int(parts[3])
    out = np.zeros((H, W, 4), np.uint8)
    for y in range(H):
        b = buf[pos:pos + 4]
        if not (8 <= W < 32768 and b[0] == 2 and b[1] == 2 and not (b[2] & 0x80)):
            rest = np.frombuffer(buf, np.uint8, count=(H - y) * W * 4, offset=pos)
            out[y:] = rest.reshape(H - y, W, 4)
            return out
        assert (b[2] << 8 | b[3]) == W
        pos += 4
        for ch in range(4):
            x = 0
            while x < W:
                cnt = buf[pos]
                pos += 1
                if cnt > 128:
                    cnt -= 128
                    out[y, x:x + cnt, ch] = buf[pos]
                    pos += 1
                else:
                    out[y, x:x + cnt, ch] = np.frombuffer(buf, np.uint8, count=cnt, offset=pos)
                    pos += cnt
                x += cnt
    return out


def rgbe_to_bgr_float(rgbe: np.ndarray) -> np.ndarray:
    """OpenCV rgbe2float: f = (float)ldexp(1.0, e - 136); channel = mantissa * f; e == 0 -> 0.  Returns B,G,R."""
    e = rgbe[..., 3].astype(np.int32)
    f = np.where(e > 0, np.ldexp(np.float64(1.0), e - 136), 0.0).astype(np.float32)
    r = rgbe[..., 0].astype(np.float32) * f
    g = rgbe[..., 1].astype(np.float32) * f
    b = rgbe[..., 2].astype(np.float32) * f
    return np.stack([b, g, r], axis=-1).astype(np.float32)


def bgr2gray_f32(bgr: np.ndarray) -> np.ndarray:
    """cv2.COLOR_BGR2GRAY on float32: 0.114 B + 0.587 G + 0.299 R, evaluated left to right in float32."""
    c = np.float32
    return (c(0.114) * bgr[..., 0] + c(0.587) * bgr[..., 1]) + c(0.299) * bgr[..., 2]


# ---------------------------------------------------------------- nlos_pose_dataloader.py:71-144
def box_round(v: np.ndarray) -> np.ndarray:
    """:114-117 / :119-121: pair averages along axis 0, then 1, then 2."""
    v = (v[::2] + v[1::2]) / 2
    v = (v[:, ::2] + v[:, 1::2]) / 2
    v = (v[:, :, ::2] + v[:, :, 1::2]) / 2
    return v


def meas_from_bgr(bgr: np.ndarray, frames: int = 600, keep: int = 512, downsample_cnt: int = 1) -> np.ndarray:
    """:75-83, :107, :113-117 given what cv2.imread returned.  Raises like :75-81 on an all-zero file."""
    meas = bgr
    if abs(meas.max()) < 1e-10:
        raise ValueError("wrong Meas File!")
    meas = meas / np.max(meas)
    meas = bgr2gray_f32(meas)
    meas = meas / np.max(meas)
    meas = meas.reshape(frames, -1, meas.shape[-1])[:keep]  # rearrange '(t h) w -> t h w', t=600
    meas = (meas[::2] + meas[1::2]) / 2
    for _ in range(downsample_cnt):
        meas = box_round(meas)
    return meas.astype(np.float32)


def addnoise_dataset(meas: np.ndarray, rng=None, sigma: float = 10.61) -> np.ndarray:
    """utils/nlos_pose_dataloader_noise.py:167-172: 'a b -> (a b)', cv2.GaussianBlur(ksize=(0, 0), sigmaX=10.61,
    BORDER_REPLICATE) along that one long column, numpy.random.poisson of the result.  `rng` = a numpy Generator /
    RandomState for the Poisson draw, None = stop after the blur (float32, deterministic).  Returned in the (a, b) shape
    (the reference hands back cv2's (a*b, 1) column and then folds it at :106 as if it were (a, b): see the note in
    hiddenpose_amd/nlos_pose_dataloader_noise.py)."""
    try:
        from nlospose_oracle import blur_flat_replicate
    except ImportError:  # imported as a package member
        from oracle.nlospose_oracle import blur_flat_replicate
    blurred = blur_flat_replicate(meas, sigma)
    if rng is None:
        return blurred
    return rng.poisson(blurred.astype(np.float64))   # int64, as numpy.random.poisson returns


def meas_from_bgr_noise(bgr: np.ndarray, rng=None, frames: int = 600, keep: int = 512, downsample_cnt: int = 1,
                        sigma: float = 10.61) -> np.ndarray:
    """utils/nlos_pose_dataloader_noise.py:86-94, :106, :112-117 given what cv2.imread returned: gray of the RAW image
    (the first '/ max' is commented out at :92), addnoise_dataset, '/ max' (float64 when the image holds int64 Poisson
    counts, float32 for the blur-only image), crop, time pairs, box rounds."""
    meas = bgr
    if abs(meas.max()) < 1e-10:
        raise ValueError("wrong Meas File!")
    meas = bgr2gray_f32(meas)
    meas = addnoise_dataset(meas, rng, sigma)
    meas = meas / np.max(meas)
    meas = meas.reshape(frames, -1, meas.shape[-1])[:keep]
    meas = (meas[::2] + meas[1::2]) / 2
    for _ in range(downsample_cnt):
        meas = box_round(meas)
    return meas.astype(np.float32)


def meas_from_hdr(buf: bytes, frames: int = 600, keep: int = 512, downsample_cnt: int = 1) -> np.ndarray:
    return meas_from_bgr(rgbe_to_bgr_float(rgbe_read(buf)), frames, keep, downsample_cnt)


def vol_pyramid(vol: np.ndarray, downsample_cnt: int = 1) -> np.ndarray:
    vol = vol.astype(np.float32)
    for _ in range(downsample_cnt):
        vol = box_round(vol)
    return vol


def remap_joints(joints: np.ndarray, vol_size: int = 256, heatmap_size: int = 64) -> np.ndarray:
    """:128-139,143: metres -> voxels of the 256^3 volume, (x, y, z) -> (d, h, w), / (VOL_SIZE/HEATMAP_SIZE)."""
    j = np.array(joints, dtype=np.float64, copy=True)
    w = j[:, 0] * 128 + 128
    h = 256 - (j[:, 1] * 128 + 128)
    d = 225 - (j[:, 2] * 128 + 128)
    return np.stack([d, h, w], axis=1) / (vol_size / heatmap_size)


# ---------------------------------------------------------------- loadrealdata.py:6-15
def load_realdata(data_new: np.ndarray, downsample_cnt: int = 1) -> np.ndarray:
    meas = np.transpose(data_new, (2, 1, 0))  # rearrange 'h w t -> t w h'
    meas = (meas[::2] + meas[1::2]) / 2
    for _ in range(downsample_cnt):
        meas = box_round(meas)
    return meas


# ---------------------------------------------------------------- test helper: write a Radiance file
def rgbe_write(rgbe: np.ndarray, rle: bool) -> bytes:
    """(H, W, 4) uint8 -> file bytes (flat or new-style RLE).  Used to build synthetic .hdr inputs."""
    H, W, _ = rgbe.shape
    out = bytearray(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (H, W))
    if not rle:
        out += rgbe.tobytes()
        return bytes(out)
    assert 8 <= W < 32768
    for y in range(H):
        out += bytes([2, 2, W >> 8, W & 255])
        for ch in range(4):
            row = rgbe[y, :, ch]
            x = 0
            while x < W:
                run = 1
                while x + run < W and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 4:
                    out += bytes([128 + run, int(row[x])])
                    x += run
                else:
                    lit = x
                    while lit < W and lit - x < 128:
                        r2 = 1
                        while lit + r2 < W and r2 < 4 and row[lit + r2] == row[lit]:
                            r2 += 1
                        if r2 >= 4:
                            break
                        lit += 1
                    n = max(1, lit - x)
                    out += bytes([n]) + row[x:x + n].tobytes()
                    x += n
    return bytes(out)
