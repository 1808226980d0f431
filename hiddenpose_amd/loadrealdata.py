"""Drop-in for the reference's utils/loadrealdata.py:6-15: measured `.mat` transient -> network input, with the
rearrangement and the averaging pyramid on the GPU (hp_pair_average_axis0 reads the (h, w, t) array through
the strides of 'h w t -> t w h', so the transposed copy is never materialised; hp_box_downsample_round)."""
from __future__ import annotations

import torch

from . import _lib
from .nlos_pose_dataloader import box_pyramid


def realdata_to_meas(data_new: torch.Tensor, downsample_cnt: int = 1) -> torch.Tensor:
    """(h, w, t) device tensor -> (t / 2^(cnt+1), w / 2^cnt, h / 2^cnt) fp32."""
    if not data_new.is_cuda:
        raise _lib.HiddenPoseHipError("realdata_to_meas needs a HIP device tensor (no CPU path)")
    x = data_new.to(torch.float32).contiguous()
    h, w, t = x.shape
    sh, sw, stt = x.stride()
    out = torch.empty(t // 2, w, h, dtype=torch.float32, device=x.device)
    st = torch.cuda.current_stream(x.device).cuda_stream
    _lib.check(_lib.lib().hp_pair_average_axis0(x.data_ptr(), out.data_ptr(), t, w, h, stt, sw, sh, st), "hp_pair_average_axis0")
    return box_pyramid(out, downsample_cnt)


def load_realdata(datapath="data/lct256_human.mat", downsample_cnt=1, device=None):
    from scipy.io import loadmat

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    data = torch.from_numpy(loadmat(datapath)["data_new"]).to(dev)
    return realdata_to_meas(data, downsample_cnt)
