"""Operator layer between the reference-shaped nn.Modules and libhiddenpose_hip.so.

Every function here is one fused stage of the hot path.  A stage
either calls a hand-written HIP kernel through the C ABI (wrapped in a
torch.autograd.Function so that optimisers and torch.distributed stay stock), or --
for rows whose kernel has not landed yet -- the stock PyTorch-ROCm device operator.
`HIP_STAGES` / `ATEN_STAGES` say which is which; DESIGN.md tracks the same table.
There is no CPU path: tensors must live on a HIP device.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

HIP_STAGES = {"lct_forward", "lct_backward", "posenet3d_50", "feature_extraction.conv", "unet3d.conv3"}
ATEN_STAGES = {"feature_extraction.leaky_add", "normalize_feature", "unet3d.groupnorm_pool_upsample_out", "softmax_integral",
               "bce_dice"}


def _need_cuda(x: torch.Tensor, what: str) -> None:
    if not x.is_cuda:
        from ._lib import HiddenPoseHipError

        raise HiddenPoseHipError(f"{what}: tensor is on {x.device}; this package has no CPU path")


# ---------------------------------------------------------------- thin-channel 3^3 convolutions
import ctypes as _C

from . import _lib


def _stream(t):
    return _lib.current_stream_handle(t.device)


class _DConv3(torch.autograd.Function):
    """y = conv3d(x, w, bias), 3x3x3, stride 1, same size, zero or replicate padding; planar
    (B,C,D,H,W).  csrc/dconv_kernels.hip."""

    @staticmethod
    def forward(ctx, x, w, bias, replicate):
        _need_cuda(x, "dconv3")
        L = _lib.lib()
        x = x.contiguous()
        b, cin, d, h, wd = x.shape
        cout = w.shape[0]
        y = torch.empty(b, cout, d, h, wd, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.hp_dconv3_forward(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), y.data_ptr(), b, cin, cout, d, h, wd,
                                           1 if replicate else 0, _stream(x)), "hp_dconv3_forward")
        ctx.save_for_backward(x, w)
        ctx.cfg = (replicate, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        x, w = ctx.saved_tensors
        replicate, has_bias = ctx.cfg
        gy = gy.contiguous()
        b, cin, d, h, wd = x.shape
        cout = w.shape[0]
        rp = 1 if replicate else 0
        st = _stream(x)
        gx = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                nb = int(L.hp_dconv3_backward_data_workspace_bytes(b, cin, d, h, wd, rp))
                ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device) if nb else None
                _lib.check(L.hp_dconv3_backward_data(gy.data_ptr(), w.data_ptr(), gx.data_ptr(), b, cin, cout, d, h, wd, rp,
                                                     _lib.ptr(ws), st), "hp_dconv3_backward_data")
            dw = torch.empty_like(w)
            db = torch.empty(cout, dtype=torch.float32, device=x.device) if has_bias else None
            _lib.check(L.hp_dconv3_backward_weight(x.data_ptr(), gy.data_ptr(), dw.data_ptr(), _lib.ptr(db), b, cin, cout,
                                                   d, h, wd, rp, st), "hp_dconv3_backward_weight")
        return gx, dw, db, None


# ---------------------------------------------------------------- FeatureExtraction (rows A1, A2)
def conv3d_reppad(x, w, b, stride=1):
    assert stride == 1, "NlosPose uses FeatureExtraction with stride 1 (models/NlosPose.py:20-24)"
    return _DConv3.apply(x, w, b, True)


def feature_extraction_fused(x, fe):
    _need_cuda(x, "feature_extraction")
    a = conv3d_reppad(x, fe.conv1[1].weight, fe.conv1[1].bias)
    a = fe.conv1[3](fe.conv1[2](a))
    return a + _DConv3.apply(x, fe.weights, None, False)


# ---------------------------------------------------------------- normalize_feature (row C8)
def normalize_feature(x):
    """(x - min)/(max(x - min) + 1e-15) * 10 per (b, c); NO ReLU (feature_propagation.py:273-286)."""
    _need_cuda(x, "normalize_feature")
    b, c = x.shape[:2]
    f = x.reshape(b, c, -1)
    z = f - f.min(2, keepdim=True)[0]
    return (z / (z.max(2, keepdim=True)[0] + 1e-15) * 10.0).view_as(x)


# ---------------------------------------------------------------- UNet3d (row U1)
def conv3d(x, w, b=None, stride=1, padding=0):
    """1x1x1 convolution (UNet3d `Out`, unet/unet3d.py:65-71): a per-voxel channel mix."""
    assert w.shape[2:] == (1, 1, 1) and stride == 1 and padding == 0
    y = torch.einsum("bcdhw,oc->bodhw", x, w.reshape(w.shape[0], w.shape[1]))
    return y if b is None else y + b.view(1, -1, 1, 1, 1)


def conv3_gn_relu(x, w, b, gw, gb, groups, eps):
    return F.relu(F.group_norm(_DConv3.apply(x, w, b, False), groups, gw, gb, eps))


def max_pool3d_2(x):
    return F.max_pool3d(x, 2, 2)


def upsample_trilinear_2x(x):
    return F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True)


# ---------------------------------------------------------------- posenet3d_50 (rows P1-P3)
# Channels-last (B, D, H, W, C) fp32 tensors between units; every kernel is in libhiddenpose_hip.so
# (csrc/conv_kernels.hip: exact-fp32 MFMA implicit GEMM; csrc/norm_kernels.hip: BN / pool / layout).
def _desc(x_cl, cout, k, stride, pad, transposed):
    b, d, h, w, cin = x_cl.shape
    return _lib.ConvDesc(b, d, h, w, cin, cout, k, stride, pad, 1 if transposed else 0)


def _out_dims(desc):
    if desc.transposed:
        return 2 * desc.Di, 2 * desc.Hi, 2 * desc.Wi
    f = lambda n: (n + 2 * desc.pad - desc.k) // desc.stride + 1
    return f(desc.Di), f(desc.Hi), f(desc.Wi)


def _pack(desc, w, want_fwd, want_dgrad):
    L = _lib.lib()
    n = int(L.hp_conv3d_packed_weight_elems(_C.byref(desc)))
    wf = torch.empty(n, dtype=torch.float32, device=w.device) if want_fwd else None
    wd = torch.empty(w.numel(), dtype=torch.float32, device=w.device) if want_dgrad else None
    _lib.check(L.hp_conv3d_pack_weight(_C.byref(desc), w.data_ptr(), _lib.ptr(wf), _lib.ptr(wd), _stream(w)),
               "hp_conv3d_pack_weight")
    return wf, wd


def _conv_grads(desc, x, w, dz, need_dx):
    """(dx, dw) of z = conv(x, w) given dz; all channels-last, dw in the torch weight layout."""
    L = _lib.lib()
    st = _stream(x)
    dx = None
    if need_dx:
        _, wd = _pack(desc, w, False, True)
        dx = torch.empty_like(x)
        _lib.check(L.hp_conv3d_backward_data(_C.byref(desc), dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), st),
                   "hp_conv3d_backward_data")
    n = int(L.hp_conv3d_packed_weight_elems(_C.byref(desc)))
    dwp = torch.empty(n, dtype=torch.float32, device=x.device)
    _lib.check(L.hp_conv3d_backward_weight(_C.byref(desc), x.data_ptr(), dz.data_ptr(), dwp.data_ptr(), st),
               "hp_conv3d_backward_weight")
    dw = torch.empty_like(w)
    _lib.check(L.hp_conv3d_unpack_wgrad(_C.byref(desc), dwp.data_ptr(), dw.data_ptr(), st), "hp_conv3d_unpack_wgrad")
    return dx, dw


class _ConvBnAct(torch.autograd.Function):
    """y = act(BN(conv(x)) [+ res]) with the BN batch statistics reduced in the conv epilogue."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, res, bn, k, stride, pad, transposed, relu):
        L = _lib.lib()
        x = x.contiguous()
        cout = w.shape[1] if transposed else w.shape[0]
        desc = _desc(x, cout, k, stride, pad, transposed)
        do, ho, wo = _out_dims(desc)
        st = _stream(x)
        with torch.cuda.device(x.device):
            wf, _ = _pack(desc, w, True, False)
            z = torch.empty(desc.B, do, ho, wo, cout, dtype=torch.float32, device=x.device)
            M = z.numel() // cout
            train = bn.training
            stats = torch.empty(2 * cout, dtype=torch.float64, device=x.device) if train else None
            _lib.check(L.hp_conv3d_forward(_C.byref(desc), x.data_ptr(), wf.data_ptr(), None, z.data_ptr(),
                                           _lib.ptr(stats), st), "hp_conv3d_forward")
            mean = torch.empty(cout, dtype=torch.float32, device=x.device)
            rstd = torch.empty_like(mean)
            if train:
                mom = 0.1 if bn.momentum is None else bn.momentum
                _lib.check(L.hp_bn_train_finalize(stats.data_ptr(), M, cout, bn.eps, mom, mean.data_ptr(), rstd.data_ptr(),
                                                  bn.running_mean.data_ptr(), bn.running_var.data_ptr(), st),
                           "hp_bn_train_finalize")
                bn.num_batches_tracked += 1
            else:
                _lib.check(L.hp_bn_eval_stats(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), cout, bn.eps,
                                              mean.data_ptr(), rstd.data_ptr(), st), "hp_bn_eval_stats")
            y = torch.empty_like(z)
            if res is not None:
                res = res.contiguous()
            _lib.check(L.hp_bn_apply(z.data_ptr(), _lib.ptr(res), y.data_ptr(), M, cout, mean.data_ptr(), rstd.data_ptr(),
                                     gamma.data_ptr(), beta.data_ptr(), 1 if relu else 0, st), "hp_bn_apply")
        ctx.save_for_backward(x, w, gamma, z, y, mean, rstd)
        ctx.cfg = (desc, relu, train, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, gamma, z, y, mean, rstd = ctx.saved_tensors
        desc, relu, train, has_res = ctx.cfg
        cout = z.shape[-1]
        M = z.numel() // cout
        dy = dy.contiguous()
        st = _stream(x)
        with torch.cuda.device(x.device):
            dz = torch.empty_like(z)
            g = torch.empty_like(z) if has_res else None
            dgamma = torch.empty_like(gamma)
            dbeta = torch.empty_like(gamma)
            ws = torch.empty(int(L.hp_bn_backward_workspace_bytes(cout)) // 4 + 2, dtype=torch.float32, device=x.device)
            _lib.check(L.hp_bn_backward(dy.data_ptr(), y.data_ptr(), z.data_ptr(), _lib.ptr(g), dz.data_ptr(), M, cout,
                                        mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), 1 if relu else 0,
                                        1 if train else 0, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), st),
                       "hp_bn_backward")
            dx, dw = _conv_grads(desc, x, w, dz, ctx.needs_input_grad[0])
        return dx, dw, dgamma, dbeta, g, None, None, None, None, None, None


class _ConvBiasToNCDHW(torch.autograd.Function):
    """Final 1^3 conv of the head: channels-last in, (B, C, D, H, W) out."""

    @staticmethod
    def forward(ctx, x, w, bias):
        L = _lib.lib()
        x = x.contiguous()
        cout = w.shape[0]
        desc = _desc(x, cout, 1, 1, 0, False)
        st = _stream(x)
        with torch.cuda.device(x.device):
            wf, _ = _pack(desc, w, True, False)
            b, d, h, wd_, _ = x.shape
            ycl = torch.empty(b, d, h, wd_, cout, dtype=torch.float32, device=x.device)
            _lib.check(L.hp_conv3d_forward(_C.byref(desc), x.data_ptr(), wf.data_ptr(), bias.data_ptr(), ycl.data_ptr(),
                                           None, st), "hp_conv3d_forward")
            y = torch.empty(b, cout, d, h, wd_, dtype=torch.float32, device=x.device)
            _lib.check(L.hp_layout_transpose(ycl.data_ptr(), y.data_ptr(), b, d * h * wd_, cout, 1, st),
                       "hp_layout_transpose")
        ctx.save_for_backward(x, w)
        ctx.desc = desc
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w = ctx.saved_tensors
        desc = ctx.desc
        dy = dy.contiguous()
        b, c, d, h, wd_ = dy.shape
        st = _stream(x)
        with torch.cuda.device(x.device):
            dycl = torch.empty(b, d, h, wd_, c, dtype=torch.float32, device=x.device)
            _lib.check(L.hp_layout_transpose(dy.data_ptr(), dycl.data_ptr(), b, d * h * wd_, c, 0, st),
                       "hp_layout_transpose")
            dx, dw = _conv_grads(desc, x, w, dycl, ctx.needs_input_grad[0])
            dbias = dy.sum(dim=(0, 2, 3, 4))
        return dx, dw, dbias


class _MaxPool3CL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        b, d, h, w, c = x.shape
        y = torch.empty(b, d // 2, h // 2, w // 2, c, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_maxpool3d_k3s2_forward(x.data_ptr(), y.data_ptr(), b, d, h, w, c, _stream(x)),
                       "hp_maxpool3d_k3s2_forward")
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        b, d, h, w, c = x.shape
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_maxpool3d_k3s2_backward(x.data_ptr(), y.data_ptr(), dy.contiguous().data_ptr(),
                                                             dx.data_ptr(), b, d, h, w, c, _stream(x)),
                       "hp_maxpool3d_k3s2_backward")
        return dx


def conv_bn_act(x, conv, bn, relu=True, residual=None):
    """x channels-last (B,D,H,W,C)."""
    return _ConvBnAct.apply(x, conv.weight, bn.weight, bn.bias, residual, bn, conv.kernel_size[0], conv.stride[0],
                            conv.padding[0], False, relu)


def deconv_bn_relu(x, deconv, bn):
    return _ConvBnAct.apply(x, deconv.weight, bn.weight, bn.bias, None, bn, 4, 2, 1, True, True)


def stem_conv_bn_relu_pool(x, conv, bn):
    """x: (B,1,D,H,W) -- identical in memory to channels-last with C = 1."""
    _need_cuda(x, "posenet3d_50")
    b, c, d, h, w = x.shape
    assert c == 1
    y = _ConvBnAct.apply(x.reshape(b, d, h, w, 1), conv.weight, bn.weight, bn.bias, None, bn, 7, 1, 3, False, True)
    return _MaxPool3CL.apply(y)


def head_conv_to_ncdhw(x, conv):
    return _ConvBiasToNCDHW.apply(x, conv.weight, conv.bias)


# ---------------------------------------------------------------- decode + losses (rows L1-L3)
def softmax_integral(preds, num_joints, W, H, D):
    """soft-argmax in voxel units, order (x,y,z) = (W,H,D axis) (utils/criterion.py:96-153)."""
    _need_cuda(preds, "softmax_integral")
    B = preds.shape[0]
    p = F.softmax(preds.reshape(B, num_joints, -1), 2).reshape(B, num_joints, D, H, W)
    ar = lambda n: torch.arange(n, dtype=p.dtype, device=p.device)
    ax = (p.sum(dim=(2, 3)) * ar(W)).sum(2, keepdim=True)
    ay = (p.sum(dim=(2, 4)) * ar(H)).sum(2, keepdim=True)
    az = (p.sum(dim=(3, 4)) * ar(D)).sum(2, keepdim=True)
    return torch.cat((ax, ay, az), dim=2).reshape(B, num_joints * 3)


def bce_dice(logits, targets, eps=1e-9):
    _need_cuda(logits, "bce_dice")
    prob = torch.sigmoid(logits)
    dice = (2.0 * (prob * targets).sum() + eps) / (prob.sum() + targets.sum())
    return F.binary_cross_entropy_with_logits(logits, targets) + (1.0 - dice)
