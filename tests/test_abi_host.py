"""CPU-side checks of the C ABI: the library loads, exports every symbol declared in
include/hiddenpose_hip.h, its host-only constant builder is bit-exact with the
reference goldens, and entry points fail with status codes (never crash) when misused."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from hiddenpose_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(hp_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol(hip_lib):
    decl = declared_symbols()
    assert decl, "no declarations found"
    for name in sorted(decl):
        assert hasattr(hip_lib, name), f"{name} declared in include/ but not exported"
    assert set(_lib.SIGNATURES) == decl, "ctypes signature table out of sync with the header"
    assert hip_lib.hp_version() >= 100


@pytest.mark.parametrize("T,N,bin_len", [(32, 16, 0.16), (128, 128, 0.04)])
def test_host_constants_match_reference(hip_lib, golden, T, N, bin_len):
    g = golden("lct_consts.npz")
    tag = f"T{T}_N{N}"
    gridz = np.zeros(T, np.float32)
    mtx = np.zeros((T, T), np.float32)
    z = np.zeros((2 * N, 2 * N), np.int32)
    cnt = np.zeros(1, np.int64)
    re_ = np.zeros((2 * T, 2 * N, 2 * N), np.float32)
    im_ = np.zeros_like(re_)
    rc = hip_lib.hp_lct_host_constants(T, N, bin_len, 2.0, gridz.ctypes.data, mtx.ctypes.data, z.ctypes.data,
                                       cnt.ctypes.data, re_.ctypes.data, im_.ctypes.data)
    assert rc == 0
    assert np.array_equal(gridz, g[tag + "_gridz"])
    r, c = np.nonzero(mtx)
    assert np.array_equal(r, g[tag + "_mtx_rows"]) and np.array_equal(c, g[tag + "_mtx_cols"])
    assert np.array_equal(mtx[r, c], g[tag + "_mtx_vals"])
    assert int(cnt[0]) == int(g[tag + "_psf_nnz"])
    assert np.array_equal(z.astype(np.int16), g[tag + "_psf_zidx"])
    if tag + "_invpsf_re" in g:
        assert np.abs(re_ - g[tag + "_invpsf_re"]).max() < 1e-6 and np.abs(im_ - g[tag + "_invpsf_im"]).max() < 1e-6
    else:
        idx = g[tag + "_invpsf_idx"]
        assert np.abs(re_.reshape(-1)[idx] - g[tag + "_invpsf_re_s"]).max() < 1e-6
        assert np.abs(im_.reshape(-1)[idx] - g[tag + "_invpsf_im_s"]).max() < 1e-6
        l2 = np.sqrt((re_.astype(np.float64) ** 2 + im_.astype(np.float64) ** 2).sum())
        assert abs(l2 / float(g[tag + "_invpsf_l2"]) - 1) < 1e-6


def test_resampler_t512_bit_exact(hip_lib, golden):
    g = golden("lct_consts.npz")
    T = 512
    mtx = np.zeros((T, T), np.float32)
    assert hip_lib.hp_lct_host_constants(T, 1, 0.01, 2.0, None, mtx.ctypes.data, None, None, None, None) == 0
    r, c = np.nonzero(mtx)
    assert np.array_equal(r, g["T512_N128_mtx_rows"]) and np.array_equal(c, g["T512_N128_mtx_cols"])
    assert np.array_equal(mtx[r, c], g["T512_N128_mtx_vals"])


def test_errors_are_status_codes(hip_lib):
    assert hip_lib.hp_lct_host_constants(100, 16, 0.1, 2.0, None, None, None, None, None, None) == -1
    assert b"power of two" in hip_lib.hp_last_error_string()
    h = C.c_void_p()
    rc = hip_lib.hp_lct_plan_create(C.byref(h), 48, 16, 0.1, 2.0, 0, 0)
    assert rc == -2 and not h.value  # unsupported length, reported before any device is touched
    assert hip_lib.hp_lct_workspace_bytes(None, 4) == 0
    assert hip_lib.hp_lct_forward(None, None, None, 1, None, 0, None) == -1
    assert hip_lib.hp_lct_plan_destroy(None) == 0


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    import torch

    from hiddenpose_amd.feature_propagation import LCT

    with pytest.raises(_lib.HiddenPoseHipError):
        LCT(16, 32, 0.16, 2.0)(torch.zeros(1, 1, 32, 16, 16))


def test_config_node():
    from hiddenpose_amd.config import get_cfg_defaults, make_cfg, update_config_t128_128x128

    c = get_cfg_defaults()
    update_config_t128_128x128(c)
    assert c.MODEL.TIME_SIZE == 128 and c.MODEL.IMAGE_SIZE == [128, 128] and c.MODEL.BIN_LEN == 0.04
    with pytest.raises(AttributeError):
        c.MODEL.TIME_SIZE = 1
    c2 = make_cfg(512, 128)
    assert abs(c2.MODEL.BIN_LEN - 0.01) < 1e-12 and c2.TRAIN.LR_STEP == [2, 4, 13]
