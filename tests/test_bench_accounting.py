"""The FLOP / byte accounting that bench.py's roofline object is computed from (CPU only)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_posenet_forward_flops_match_survey():
    # posenet3d_50 forward = 995.4 GFLOP per 128^3 sample (PyTorch FlopCounterMode, 2 x MAC)
    f = bench.posenet_conv_flops(128, 128, 1)
    fwd = sum(f[k] for k in ("conv_igemm_stem", "conv_igemm_k1", "conv_igemm_k3", "conv_igemm_deconv"))
    assert abs(fwd / 1e9 - 995.4) < 0.1
    # every convolution has a weight gradient of the same cost; every one but none is skipped for data gradients
    assert f["conv_wgrad"] == fwd
    assert f["conv_igemm_dgrad"] + f["conv_stem_dgrad"] == fwd
    # 128x128x512 is exactly 4x the voxels
    g = bench.posenet_conv_flops(512, 128, 1)
    assert all(abs(g[k] / f[k] - 4.0) < 1e-12 for k in f)
    assert abs(sum(g[k] for k in ("conv_igemm_stem", "conv_igemm_k1", "conv_igemm_k3", "conv_igemm_deconv")) / 1e9 - 3981.4) < 0.5


def test_lct_algorithmic_bytes():
    V = 128 ** 3
    total = sum(bench.algorithmic_work(k, 128, 128, 2)[1] for k in
                ("lct_axis_fwd_t", "lct_axis_fwd_h", "lct_axis_mid_w", "lct_axis_inv_h", "lct_axis_inv_t"))
    # one packed pair (2 samples): (8V + 16V) + 48V + (64V + 64V) + 48V + (16V + 8V) = 272V bytes = 136V per sample
    assert total == 272 * V
