"""Operator layer between the reference-shaped nn.Modules and libhiddenpose_hip.so.

Every function here is one fused stage of the hot path: a hand-written
HIP kernel called through the C ABI, wrapped in a torch.autograd.Function so that optimisers
and torch.distributed stay stock.  `HIP_STAGES` lists the rows; `ATEN_STAGES` names the only
stock device operator left on the path (Adam).  There is no CPU path: tensors must live on a
HIP device.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

HIP_STAGES = {"feature_extraction", "lct_forward", "lct_backward", "normalize_feature", "unet3d", "posenet3d_50",
              "softmax_integral", "weighted_mse", "bce_dice"}
ATEN_STAGES = {"Adam"}


def _need_cuda(x: torch.Tensor, what: str) -> None:
    if not x.is_cuda:
        from ._lib import HiddenPoseHipError

        raise HiddenPoseHipError(f"{what}: tensor is on {x.device}; this package has no CPU path")


# ---------------------------------------------------------------- thin-channel 3^3 convolutions
import ctypes as _C

from . import _lib


def _stream(t):
    return _lib.current_stream_handle(t.device)


_DCONV_PRECISION = 0  # HP_PRECISION_FP32; 1 = HP_PRECISION_BF16 (forward and data gradient on v_mfma_f32_4x4x4_16b_bf16)


def set_dconv_precision(name: str) -> str:
    """Arithmetic of the thin-channel 3^3 convolutions (U-Net / FeatureExtraction layers with more than one channel):
    "fp32" (exact, default) or "bf16" (operands rounded to bf16, fp32 accumulation: BASELINE configs[2], "bf16 with fp32
    LCT").  Returns the previous setting; a node's backward uses the precision its forward ran with."""
    global _DCONV_PRECISION
    prev = _DCONV_NAMES[_DCONV_PRECISION]
    _DCONV_PRECISION = _DCONV_NAMES.index(name)
    return prev


# "bf16emu" is a checker, not a mode to run: the EXACT kernels on operands rounded to bf16 beforehand (three extra passes per
# convolution).  Products and accumulation type are those of the bf16 kernels, so the two agree up to fp32 summation order
# through a whole network (tests/test_stages_gpu.py) -- which pins the bf16 path in situ, independent of how strongly a
# randomly filled network amplifies the operand rounding itself.
_DCONV_NAMES = ("fp32", "bf16", "bf16emu")
# A/B switch: HP_DCONV_WGRAD_BF16=0 keeps the weight gradients of the bf16 mode on the exact kernel
_DCONV_WGRAD_STREAM = bool(int(__import__("os").environ.get("HP_DCONV_WGRAD_STREAM", "1")))   # thin-channel weight gradients on the second stream
_DCONV_WGRAD_BF16 = os.environ.get("HP_DCONV_WGRAD_BF16", "1") != "0"


def _emu(t, prec, cin, cout):
    return t.bfloat16().float() if prec == 2 and not (cin == 1 and cout == 1) else t


_side_dconv_calls = [0]    # thin-channel weight gradients that took the second stream (tests)
_side_conv_calls = [0]     # the regressor's


def _note_use(w):
    """Forward half of the shared-weight test of the side-stream weight gradients: a weight used twice in one graph has its two
    gradients SUMMED by autograd on the main stream, so neither may be written on the side stream.  Called by the Python
    wrappers right before `Function.apply` -- inside `forward` grad mode is always off and says nothing about the caller's
    (rounds 3-4 counted there, i.e. never: no model of this package shares a convolution weight, so nothing ever raced).
    `w._hp_uses`: {forward epoch: uses not yet consumed by a backward}; entries older than 8 epochs are dropped (a graph that was
    built and never differentiated must not keep a weight off the side stream for the rest of the process)."""
    if torch.is_grad_enabled() and w is not None and w.requires_grad:
        ep = _use_epoch[0]
        d = getattr(w, "_hp_uses", None)
        if not isinstance(d, dict):
            d = {}
        for k in [k for k in d if k < ep - 8]:
            del d[k]
        d[ep] = d.get(ep, 0) + 1
        w._hp_uses = d


def _shared_use(ctx, w):
    """Backward half: True if this weight's gradient must stay on the main stream -- it has more than one outstanding use (in
    this forward or in another one whose graph may be part of the same backward), or this use was never counted (a caller
    that bypassed the wrappers: nothing is known).  Releases this use; the flag `_hp_shared` keeps the remaining uses of the
    pass on the main stream as well."""
    d = getattr(w, "_hp_uses", None)
    if not isinstance(d, dict):
        d = {}
    total, mine = sum(d.values()), d.get(ctx.use_epoch, 0)
    shared = total > 1 or mine == 0 or getattr(w, "_hp_shared", False)
    if total > 1:
        w._hp_shared = True
    if mine > 0:
        if mine == 1:
            del d[ctx.use_epoch]
        else:
            d[ctx.use_epoch] = mine - 1
        if not d:
            w._hp_shared = False
    return shared


def _dconv3_grads(x, w, g, replicate, has_bias, need_dx, prec=0, w_param=None, b_param=None):
    """(dx, dw, db) of y = conv3(x, w) + b given g = dL/dy; planar tensors."""
    L = _lib.lib()
    b, cin, d, h, wd = x.shape
    cout = w.shape[0]
    rp = 1 if replicate else 0
    st = _stream(x)
    gx = None
    if need_dx:
        gx = torch.empty_like(x)
        # the 1 -> 1 layers (FeatureExtraction) fold the replicate padding inside the stencil kernel: no halo workspace
        # (270 MB at 1024 x 256 x 256) unless the A/B switch puts them back on the matrix-core kernel
        c1 = cin == 1 and cout == 1 and not int(__import__("os").environ.get("HP_DCONV_C1_MFMA", "0") or 0)
        nb = 0 if c1 else int(L.hp_dconv3_backward_data_workspace_bytes(b, cin, d, h, wd, rp))
        ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device) if nb else None
        ge, we = _emu(g, prec, cin, cout), _emu(w, prec, cin, cout)
        _lib.check(L.hp_dconv3_backward_data_p(ge.data_ptr(), we.data_ptr(), gx.data_ptr(), b, cin, cout, d, h, wd, rp, prec & 1,
                                               _lib.ptr(ws), st), "hp_dconv3_backward_data_p")
    dw = torch.empty_like(w)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if has_bias else None
    nbw = int(L.hp_dconv3_backward_weight_workspace_bytes(b, cin, cout, d, h, wd))
    wsw = torch.empty(nbw // 4, dtype=torch.float32, device=x.device)
    wprec = prec if _DCONV_WGRAD_BF16 else 0
    emu_w = wprec == 2 and cin > 1 and wd % 4 == 0   # the layers the bf16 weight-gradient kernel takes (others run exact)
    xe, ge = (x.bfloat16().float(), g.bfloat16().float()) if emu_w else (x, g)
    # The weight gradient has no reader before the optimizer: second stream, as the regressor's (see _conv_grads for the
    # allocation / lifetime rules: outputs allocated here on the main stream and not held, operands held until the main stream has
    # waited for the kernel).  Only the plain case: leaf parameters without hooks whose .grad autograd will simply adopt.
    side = _wgrad_side_stream(x.device) if (_DCONV_WGRAD_STREAM and w_param is not None) else None
    if side is not None:
        ps = [w_param] + ([b_param] if b_param is not None else [])
        if any((not q.is_leaf) or q.grad is not None or getattr(q, "_backward_hooks", None) or
               getattr(q, "_post_accumulate_grad_hooks", None) for q in ps):
            side = None
    if side is None:
        _lib.check(L.hp_dconv3_backward_weight_p(xe.data_ptr(), ge.data_ptr(), dw.data_ptr(), _lib.ptr(db), b, cin, cout,
                                                 d, h, wd, rp, wprec & 1, wsw.data_ptr(), st), "hp_dconv3_backward_weight_p")
        return gx, dw, db
    main = torch.cuda.current_stream(x.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        _lib.check(L.hp_dconv3_backward_weight_p(xe.data_ptr(), ge.data_ptr(), dw.data_ptr(), _lib.ptr(db), b, cin, cout,
                                                 d, h, wd, rp, wprec & 1, wsw.data_ptr(), _stream(x)), "hp_dconv3_backward_weight_p")
    _hold(x.device, side, main, (xe, ge, wsw))
    _side_dconv_calls[0] += 1
    return gx, dw, db


class _DConv3(torch.autograd.Function):
    """y = leaky(conv3d(x, w, bias) [+ res], slope), 3x3x3, stride 1, same size, zero or replicate padding; planar
    (B,C,D,H,W); slope 1 = no activation.  The residual add and the activation ride in the convolution's epilogue
    (csrc/dconv_kernels.hip, hp_dconv3_forward_fused)."""

    @staticmethod
    def forward(ctx, x, w, bias, replicate, res=None, slope=1.0):
        _need_cuda(x, "dconv3")
        L = _lib.lib()
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        b, cin, d, h, wd = x.shape
        cout = w.shape[0]
        y = torch.empty(b, cout, d, h, wd, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            xe, we = _emu(x, _DCONV_PRECISION, cin, cout), _emu(w, _DCONV_PRECISION, cin, cout)
            _lib.check(L.hp_dconv3_forward_fused_p(xe.data_ptr(), we.data_ptr(), _lib.ptr(bias), _lib.ptr(res), y.data_ptr(), None,
                                                   b, cin, cout, d, h, wd, 1 if replicate else 0, float(slope),
                                                   _DCONV_PRECISION & 1, _stream(x)), "hp_dconv3_forward_fused_p")
        if slope != 1.0:
            ctx.save_for_backward(x, w, y)
        else:
            ctx.save_for_backward(x, w)
        ctx.cfg = (replicate, bias is not None, res is not None, float(slope))
        ctx.prec = _DCONV_PRECISION
        ctx.bias_ref = bias     # (the parameter itself, for the side-stream test of its backward; not a saved tensor: no version check)
        ctx.use_epoch = _use_epoch[0]
        return y

    @staticmethod
    def backward(ctx, gy):
        replicate, has_bias, has_res, slope = ctx.cfg
        gy = gy.contiguous()
        if slope != 1.0:
            x, w, y = ctx.saved_tensors
            g = torch.empty_like(gy)
            with torch.cuda.device(gy.device):
                _lib.check(_lib.lib().hp_leaky_backward(gy.data_ptr(), y.data_ptr(), g.data_ptr(), gy.numel(), slope,
                                                        _stream(gy)), "hp_leaky_backward")
        else:
            x, w = ctx.saved_tensors
            g = gy
        with torch.cuda.device(x.device):
            gx, dw, db = _dconv3_grads(x, w, g, replicate, has_bias, ctx.needs_input_grad[0], ctx.prec,
                                       None if _shared_use(ctx, w) else w, ctx.bias_ref)
        return gx, dw, db, None, (g if has_res else None), None


class _ConvGnRelu(torch.autograd.Function):
    """y = relu(GroupNorm(conv3d(x, w, bias)))  (DoubleConv half, unet/unet3d.py:14-24) as ONE autograd node: the
    convolution's epilogue leaves the per-channel sums GroupNorm needs (no statistics pass over z), the backward rebuilds
    the ReLU mask from z (the normalised tensor is not read back)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, groups, eps, out=None):
        _need_cuda(x, "conv3_gn_relu")
        L = _lib.lib()
        x = x.contiguous()
        b, cin, d, h, wd = x.shape
        cout = w.shape[0]
        V = d * h * wd
        dev = x.device
        z = torch.empty(b, cout, d, h, wd, dtype=torch.float32, device=dev)
        # `out`: a contiguous (b, cout, d, h, wd) view to write y into (the U-Net hands the first half of a decoder's
        # concatenation buffer here, so that the skip tensor never has to be copied into it)
        y = out if out is not None and out.shape == z.shape and out.is_contiguous() and out.dtype == z.dtype else torch.empty_like(z)
        stats = torch.empty(2 * b * cout, dtype=torch.float64, device=dev)
        mean = torch.empty(b * groups, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        aff = torch.empty(2, b * cout, dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.hp_groupnorm_workspace_bytes(b, cout)) // 4 + 2, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _stream(x)
            xe, we = _emu(x, _DCONV_PRECISION, cin, cout), _emu(w, _DCONV_PRECISION, cin, cout)
            _lib.check(L.hp_dconv3_forward_fused_p(xe.data_ptr(), we.data_ptr(), _lib.ptr(bias), None, z.data_ptr(), stats.data_ptr(),
                                                   b, cin, cout, d, h, wd, 0, 1.0, _DCONV_PRECISION & 1, st), "hp_dconv3_forward_fused_p")
            _lib.check(L.hp_groupnorm_relu_forward_v2(z.data_ptr(), y.data_ptr(), b, cout, groups, V, gamma.data_ptr(),
                                                      beta.data_ptr(), eps, stats.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                      aff[0].data_ptr(), aff[1].data_ptr(), ws.data_ptr(), st),
                       "hp_groupnorm_relu_forward_v2")
        ctx.save_for_backward(x, w, z, gamma, mean, rstd, aff)
        ctx.cfg = (groups, bias is not None)
        ctx.prec = _DCONV_PRECISION
        ctx.bias_ref = bias
        ctx.use_epoch = _use_epoch[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, z, gamma, mean, rstd, aff = ctx.saved_tensors
        groups, has_bias = ctx.cfg
        dy = dy.contiguous()
        b, cout = z.shape[:2]
        V = z.numel() // (b * cout)
        dz = torch.empty_like(z)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        ws = torch.empty(int(L.hp_groupnorm_workspace_bytes(b, cout)) // 4 + 2, dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            _lib.check(L.hp_groupnorm_relu_backward_v2(dy.data_ptr(), z.data_ptr(), dz.data_ptr(), b, cout, groups, V,
                                                       gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), aff[0].data_ptr(),
                                                       aff[1].data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
                                                       _stream(z)), "hp_groupnorm_relu_backward_v2")
            gx, dw, db = _dconv3_grads(x, w, dz, False, has_bias, ctx.needs_input_grad[0], ctx.prec,
                                       None if _shared_use(ctx, w) else w, ctx.bias_ref)
        return gx, dw, db, dgamma, dbeta, None, None, None


# ---------------------------------------------------------------- FeatureExtraction (rows A1, A2)
def conv3d_reppad(x, w, b, stride=1, residual=None, slope=1.0):
    """ReplicationPad3d(1) + Conv3d(3^3) [+ residual] [+ LeakyReLU(slope)] in one kernel."""
    assert stride == 1, "NlosPose uses FeatureExtraction with stride 1 (models/NlosPose.py:20-24)"
    _note_use(w)
    return _DConv3.apply(x, w, b, True, residual, slope)


class _LeakyAdd(torch.autograd.Function):
    """y = leaky_relu(a [+ b], slope); slope = 1 gives a plain add.  csrc/misc_kernels.hip."""

    @staticmethod
    def forward(ctx, a, b, slope):
        _need_cuda(a, "leaky_add")
        a = a.contiguous()
        b = b.contiguous() if b is not None else None
        y = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _lib.check(_lib.lib().hp_leaky_add_forward(a.data_ptr(), _lib.ptr(b), y.data_ptr(), a.numel(), slope,
                                                       _stream(a)), "hp_leaky_add_forward")
        ctx.slope, ctx.has_b = slope, b is not None
        if slope != 1.0:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.slope == 1.0:
            return dy, (dy if ctx.has_b else None), None
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        g = torch.empty_like(dy)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.lib().hp_leaky_backward(dy.data_ptr(), y.data_ptr(), g.data_ptr(), dy.numel(), ctx.slope,
                                                    _stream(dy)), "hp_leaky_backward")
        return g, (g if ctx.has_b else None), None


def leaky_add(a, b=None, slope=0.2):
    return _LeakyAdd.apply(a, b, slope)


def add(a, b):
    return _LeakyAdd.apply(a, b, 1.0)


def feature_extraction_fused(x, fe):
    _need_cuda(x, "feature_extraction")
    a = conv3d_reppad(x, fe.conv1[1].weight, fe.conv1[1].bias)
    a = fe.conv1[3](fe.conv1[2](a))
    _note_use(fe.weights)
    return _DConv3.apply(x, fe.weights, None, False, a, 1.0)   # box filter branch + learned branch, added in the epilogue


# ---------------------------------------------------------------- normalize_feature (row C8)
class _NormalizeFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x, "normalize_feature")
        x = x.contiguous()
        nvol = x.shape[0] * x.shape[1]
        V = x.numel() // nvol
        y = torch.empty_like(x)
        keys = torch.empty(2 * nvol, dtype=torch.int64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_normalize_feature_forward(x.data_ptr(), y.data_ptr(), nvol, V, 10.0, keys.data_ptr(),
                                                               _stream(x)), "hp_normalize_feature_forward")
        ctx.save_for_backward(x, keys)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, keys = ctx.saved_tensors
        dy = dy.contiguous()
        nvol = x.shape[0] * x.shape[1]
        V = x.numel() // nvol
        dx = torch.empty_like(x)
        ws = torch.empty(2 * nvol, dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_normalize_feature_backward(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), nvol, V, 10.0,
                                                                keys.data_ptr(), ws.data_ptr(), _stream(x)),
                       "hp_normalize_feature_backward")
        return dx


def normalize_feature(x):
    """(x - min)/(max(x - min) + 1e-15) * 10 per (b, c); NO ReLU (feature_propagation.py:273-286)."""
    return _NormalizeFeature.apply(x)


# ---------------------------------------------------------------- UNet3d (row U1)
class _GroupNormRelu(torch.autograd.Function):
    """y = relu(GroupNorm(z)) on its own (the U-Net itself uses _ConvGnRelu)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, groups, eps):
        L = _lib.lib()
        z = z.contiguous()
        b, c = z.shape[:2]
        V = z.numel() // (b * c)
        y = torch.empty_like(z)
        mean = torch.empty(b * groups, dtype=torch.float32, device=z.device)
        rstd = torch.empty_like(mean)
        aff = torch.empty(2, b * c, dtype=torch.float32, device=z.device)
        ws = torch.empty(int(L.hp_groupnorm_workspace_bytes(b, c)) // 4 + 2, dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            _lib.check(L.hp_groupnorm_relu_forward_v2(z.data_ptr(), y.data_ptr(), b, c, groups, V, gamma.data_ptr(),
                                                      beta.data_ptr(), eps, None, mean.data_ptr(), rstd.data_ptr(),
                                                      aff[0].data_ptr(), aff[1].data_ptr(), ws.data_ptr(), _stream(z)),
                       "hp_groupnorm_relu_forward_v2")
        ctx.save_for_backward(z, gamma, mean, rstd, aff)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        z, gamma, mean, rstd, aff = ctx.saved_tensors
        dy = dy.contiguous()
        b, c = z.shape[:2]
        V = z.numel() // (b * c)
        dz = torch.empty_like(z)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        ws = torch.empty(int(L.hp_groupnorm_workspace_bytes(b, c)) // 4 + 2, dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            _lib.check(L.hp_groupnorm_relu_backward_v2(dy.data_ptr(), z.data_ptr(), dz.data_ptr(), b, c, ctx.groups, V,
                                                       gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), aff[0].data_ptr(),
                                                       aff[1].data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
                                                       _stream(z)), "hp_groupnorm_relu_backward_v2")
        return dz, dgamma, dbeta, None, None


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        b, c, d, h, w = x.shape
        y = torch.empty(b, c, d // 2, h // 2, w // 2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_maxpool3d_k2_forward(x.data_ptr(), y.data_ptr(), b * c, d, h, w, _stream(x)),
                       "hp_maxpool3d_k2_forward")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        b, c, d, h, w = x.shape
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_maxpool3d_k2_backward(x.data_ptr(), dy.contiguous().data_ptr(), dx.data_ptr(), b * c, d,
                                                           h, w, _stream(x)), "hp_maxpool3d_k2_backward")
        return dx


class _PoolSkip(torch.autograd.Function):
    """(skip, pooled) = (x, max_pool3d(x, 2)) as ONE node: a U-Net level's output feeds the next level's pool AND the
    decoder's concatenation (unet/unet3d.py:31-39, 42-62), and with one node for both the two gradients meet inside the
    pool's backward kernel (hp_maxpool3d_k2_backward_add) -- the skip gradient is read in place from the concatenation's
    gradient -- instead of being copied out and summed by a separate accumulation pass."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        b, c, d, h, w = x.shape
        y = torch.empty(b, c, d // 2, h // 2, w // 2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_maxpool3d_k2_forward(x.data_ptr(), y.data_ptr(), b * c, d, h, w, _stream(x)),
                       "hp_maxpool3d_k2_forward")
        ctx.save_for_backward(x)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, g_skip, g_pool):
        (x,) = ctx.saved_tensors
        if g_pool is None:
            return g_skip
        b, c, d, h, w = x.shape
        L = _lib.lib()
        dx = torch.empty_like(x)
        V = d * h * w
        with torch.cuda.device(x.device):
            if g_skip is None:
                _lib.check(L.hp_maxpool3d_k2_backward(x.data_ptr(), g_pool.contiguous().data_ptr(), dx.data_ptr(), b * c, d, h, w,
                                                      _stream(x)), "hp_maxpool3d_k2_backward")
            else:
                st = g_skip.stride()
                if not (st[4] == 1 and st[3] == w and st[2] == h * w and st[1] == V and st[0] >= c * V and st[0] % 2 == 0
                        and g_skip.data_ptr() % 8 == 0):
                    g_skip = g_skip.contiguous()
                    st = g_skip.stride()
                _lib.check(L.hp_maxpool3d_k2_backward_add(x.data_ptr(), g_pool.contiguous().data_ptr(), g_skip.data_ptr(), st[0],
                                                          dx.data_ptr(), b, c, d, h, w, _stream(x)), "hp_maxpool3d_k2_backward_add")
        return dx


class _UpsampleCat(torch.autograd.Function):
    """cat([skip, upsample2x_trilinear_align_corners(x1)], dim=1) without materialising the upsampled tensor."""

    @staticmethod
    def forward(ctx, x1, skip, buf=None):
        L = _lib.lib()
        x1, skip = x1.contiguous(), skip.contiguous()
        b, c1, d, h, w = x1.shape
        c2 = skip.shape[1]
        assert skip.shape[2:] == (2 * d, 2 * h, 2 * w), "UNet3d skip and upsampled sizes must match (even input sizes)"
        # `buf`: the concatenation buffer whose first c2 channels ALREADY hold the skip tensor (it was produced there): no copy
        in_place = (buf is not None and buf.shape == (b, c1 + c2, 2 * d, 2 * h, 2 * w) and buf.is_contiguous()
                    and buf.dtype == torch.float32 and b == 1 and skip.data_ptr() == buf.data_ptr())
        out = buf if in_place else torch.empty(b, c1 + c2, 2 * d, 2 * h, 2 * w, dtype=torch.float32, device=x1.device)
        st = _stream(x1)
        with torch.cuda.device(x1.device):
            if not in_place:
                _lib.check(L.hp_channel_slice_copy(skip.data_ptr(), out.data_ptr(), b, c2, 8 * d * h * w, c1 + c2, 0, 0, st),
                           "hp_channel_slice_copy")
            ws = torch.empty(int(L.hp_upsample_trilinear2x_forward_workspace_bytes(b, c1, d, h, w)) // 4, dtype=torch.float32,
                             device=x1.device)
            _lib.check(L.hp_upsample_trilinear2x_forward_ws(x1.data_ptr(), out.data_ptr(), b, c1, d, h, w, c1 + c2, c2,
                                                            ws.data_ptr(), st), "hp_upsample_trilinear2x_forward_ws")
        ctx.dims = (b, c1, c2, d, h, w)
        return out

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        b, c1, c2, d, h, w = ctx.dims
        dy = dy.contiguous()
        dx1 = torch.empty(b, c1, d, h, w, dtype=torch.float32, device=dy.device)
        # the skip's gradient = the first c2 channels of dy, handed on as a VIEW (no copy): its consumer is _PoolSkip's backward,
        # which reads it in place; any other consumer takes a strided gradient like any other
        dskip = dy[:, :c2]
        st = _stream(dy)
        with torch.cuda.device(dy.device):
            ws = torch.empty(int(L.hp_upsample_trilinear2x_backward_workspace_bytes(b, c1, d, h, w)) // 4, dtype=torch.float32,
                             device=dy.device)
            _lib.check(L.hp_upsample_trilinear2x_backward_ws(dy.data_ptr(), dx1.data_ptr(), b, c1, d, h, w, c1 + c2, c2,
                                                             ws.data_ptr(), st), "hp_upsample_trilinear2x_backward_ws")
        return dx1, dskip, None


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        x = x.contiguous()
        b, cin = x.shape[:2]
        cout = w.shape[0]
        V = x.numel() // (b * cin)
        y = torch.empty((b, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_conv1x1_forward(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), y.data_ptr(), b, cin, cout,
                                                     V, _stream(x)), "hp_conv1x1_forward")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        b, cin = x.shape[:2]
        cout = w.shape[0]
        V = x.numel() // (b * cin)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_conv1x1_backward(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                      _lib.ptr(db), b, cin, cout, V, _stream(x)), "hp_conv1x1_backward")
        return dx, dw, db


class _Conv1x1Sum(torch.autograd.Function):
    """(conv1x1(x), conv1x1(x) + addend) in one pass; the backward adds the two incoming gradients of the convolution's
    output on load (row U2: `feature + refine` has no launch of its own in either direction)."""

    @staticmethod
    def forward(ctx, x, w, bias, addend):
        x, addend = x.contiguous(), addend.contiguous()
        b, cin = x.shape[:2]
        cout = w.shape[0]
        V = x.numel() // (b * cin)
        y = torch.empty((b, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        assert addend.shape == y.shape
        s = torch.empty_like(y)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_conv1x1_forward_sum(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), addend.data_ptr(), y.data_ptr(),
                                                         s.data_ptr(), b, cin, cout, V, _stream(x)), "hp_conv1x1_forward_sum")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y, s

    @staticmethod
    def backward(ctx, dy, dsum):
        x, w = ctx.saved_tensors
        if dy is None and dsum is None:
            return None, None, None, None
        g1, g2 = (dy, dsum) if dy is not None else (dsum, None)
        g1 = g1.contiguous()
        g2 = g2.contiguous() if g2 is not None else None
        b, cin = x.shape[:2]
        cout = w.shape[0]
        V = x.numel() // (b * cin)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hp_conv1x1_backward_sum(x.data_ptr(), w.data_ptr(), g1.data_ptr(), _lib.ptr(g2), dx.data_ptr(),
                                                          dw.data_ptr(), _lib.ptr(db), b, cin, cout, V, _stream(x)),
                       "hp_conv1x1_backward_sum")
        return dx, dw, db, dsum


def conv3d_sum(x, w, b, addend):
    """(conv(x), conv(x) + addend) for the 1x1x1 output convolution of UNet3d."""
    _need_cuda(x, "unet3d")
    assert tuple(w.shape[2:]) == (1, 1, 1)
    return _Conv1x1Sum.apply(x, w, b, addend)


def conv3d(x, w, b=None, stride=1, padding=0):
    """1x1x1 convolution (UNet3d `Out`, unet/unet3d.py:65-71)."""
    assert tuple(w.shape[2:]) == (1, 1, 1) and stride == 1 and padding == 0
    return _Conv1x1.apply(x, w, b)


def conv3_gn_relu(x, w, b, gw, gb, groups, eps, out=None):
    _note_use(w)
    return _ConvGnRelu.apply(x, w, b, gw, gb, groups, eps, out)


def max_pool3d_2(x):
    return _MaxPool2.apply(x)


def upsample_cat(x1, skip, buf=None):
    return _UpsampleCat.apply(x1, skip, buf)


def pool_and_skip(x):
    """(x, max_pool3d(x, 2)) with the two gradients of x summed inside the pool's backward pass."""
    return _PoolSkip.apply(x)


# ---------------------------------------------------------------- posenet3d_50 (rows P1-P3)
# Channels-last (B, D, H, W, C) fp32 tensors between units; every kernel is in libhiddenpose_hip.so
# (csrc/conv_kernels.hip: exact-fp32 MFMA implicit GEMM; csrc/norm_kernels.hip: BN / pool / layout).
_PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3, "bf16s": 1}  # HP_PRECISION_* of include/hiddenpose_hip.h
_conv_precision = 0
_conv_precision_name = "fp32"
# "bf16s" = BASELINE configs[2] in full: bf16 matrix cores AND bf16 activation storage -- every tensor that crosses an
# autograd-node boundary inside the regressor (unit outputs y, their gradients) and the gradients handed from BatchNorm
# to the convolution gradients (dz) are bf16 in HBM; raw convolution outputs z, statistics, weights, weight
# gradients and everything outside the regressor stay fp32.  ("bf16": bf16 matrix cores over fp32 tensors.)
_act_bf16 = False
_BF = torch.bfloat16
HP_IO_X, HP_IO_Y, HP_IO_DY, HP_IO_DX = 1, 2, 4, 8         # hp_conv_desc.io
HP_BN_ACT, HP_BN_DY, HP_BN_DZ, HP_BN_Z = 1, 2, 4, 8        # `io` of the hp_bn_* calls


def _h(t) -> bool:
    return t is not None and t.dtype == _BF


def set_conv_precision(name: str) -> str:
    """Arithmetic of the regressor's convolution GEMMs: 'fp32' (exact-fp32 MFMA, default), 'bf16'
    (bf16 operands, fp32 accumulation; tensors stay fp32), or the split modes 'bf16x3' / 'bf16x6' (each fp32
    operand as 2 / 3 bf16 planes, 3 / 6 plane products: ~2^-16 / ~2^-24 relative error per product).  Returns the previous setting.  The choice is
    recorded in the descriptor each forward call builds, so a backward pass uses what its forward used."""
    global _conv_precision
    if name not in _PRECISIONS:
        raise ValueError(f"conv precision must be one of {sorted(_PRECISIONS)}, got {name!r}")
    global _conv_precision_name, _act_bf16
    prev = get_conv_precision()
    _conv_precision = _PRECISIONS[name]
    _conv_precision_name = name
    _act_bf16 = name == "bf16s"
    return prev


def get_conv_precision() -> str:
    return _conv_precision_name


def _desc(x_cl, cout, k, stride, pad, transposed):
    b, d, h, w, cin = x_cl.shape
    return _lib.ConvDesc(b, d, h, w, cin, cout, k, stride, pad, 1 if transposed else 0, _conv_precision)


def _out_dims(desc):
    if desc.transposed:
        return 2 * desc.Di, 2 * desc.Hi, 2 * desc.Wi
    f = lambda n: (n + 2 * desc.pad - desc.k) // desc.stride + 1
    return f(desc.Di), f(desc.Hi), f(desc.Wi)


def _same_as_packed(desc):
    """1^3 convolutions: the packed forward / weight-gradient layout [tap][Cout][Cin] IS the torch layout
    (Cout, Cin, 1, 1, 1), so neither a pack nor an unpack launch is needed."""
    return desc.k == 1 and not desc.transposed


HP_IO_W = 16


def _w_half(desc, gathered_is_bf16: bool, gathered_channels: int) -> bool:
    """bf16 packed weights: the bf16-storage tiles of the implicit GEMM (bf16 gathered tensor, single-plane bf16
    arithmetic, its channel count a multiple of 64, not the stem)."""
    return gathered_is_bf16 and desc.precision == 1 and gathered_channels % 64 == 0 and not (desc.Cin == 1 and desc.k == 7)


def _pack(desc, w, want_fwd, want_dgrad, half=False):
    """Packed weight images of hp_conv3d_pack_weight: (w_fwd, w_dgrad); `half`: as bf16 (see _w_half)."""
    L = _lib.lib()
    n = int(L.hp_conv3d_packed_weight_elems(_C.byref(desc)))
    io0 = desc.io
    desc.io = HP_IO_W if half else 0
    try:
        if want_fwd and not half and _same_as_packed(desc) and w.is_contiguous():
            wf, want_fwd = w.detach().reshape(-1), False
            if not want_dgrad:
                return wf, None
            wd = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
            _lib.check(L.hp_conv3d_pack_weight(_C.byref(desc), w.data_ptr(), None, wd.data_ptr(), _stream(w)), "hp_conv3d_pack_weight")
            return wf, wd
        dt = _BF if half else torch.float32
        wf = torch.empty(n, dtype=dt, device=w.device) if want_fwd else None
        wd = torch.empty(w.numel(), dtype=dt, device=w.device) if want_dgrad else None
        _lib.check(L.hp_conv3d_pack_weight(_C.byref(desc), w.data_ptr(), _lib.ptr(wf), _lib.ptr(wd), _stream(w)),
                   "hp_conv3d_pack_weight")
        return wf, wd
    finally:
        desc.io = io0


class GradLink:
    """Carries one gradient contribution to a tensor that is used twice inside a Bottleneck (block input:
    conv1 + identity shortcut, or conv1 + shortcut convolution) from the autograd node that produces it
    first to the data-gradient GEMM that runs last, which adds it in its epilogue.  Replaces the
    separate accumulation pass autograd would issue."""
    __slots__ = ("g", "arrivals", "mask")

    def __init__(self):
        self.g = None
        self.arrivals = 0
        self.mask = None   # byte mask gating `g` (identity shortcut: g is the block's raw output gradient)

    def take(self):
        """One consumer of the block input runs its backward: returns (addend, mask, last).  The link is consumed
        by ONE backward pass over the graph; a second pass (retain_graph=True, torch.autograd.grad twice) would
        silently drop the parked partial sums, so it raises instead."""
        if self.arrivals <= 0:
            raise RuntimeError("hiddenpose_amd: a Bottleneck's fused gradient link was already consumed -- a second "
                               "backward pass over the same graph (retain_graph=True) is not supported; run the "
                               "forward again")
        self.arrivals -= 1
        addend, self.g = self.g, None
        mask, self.mask = self.mask, None
        return addend, mask, self.arrivals == 0


class ResLink:
    """Joins the shortcut unit (downsample conv + BN) of a Bottleneck to the unit that adds its output as the
    residual.  The gradient of the residual branch is dy (.) [block output > 0]; instead of writing that product
    the residual unit hands back dy itself and leaves the sign mask of its output here, and the shortcut unit's
    BatchNorm backward applies the mask while it reads dy."""
    __slots__ = ("mask", "affine", "short", "done")

    def __init__(self):
        self.mask = None
        self.affine = None   # (mean, rstd, gamma, beta) of the shortcut's BatchNorm while its output is still raw
        self.short = None    # (z, mean, rstd, gamma, train) of the shortcut unit: the residual unit's backward runs
        self.done = None     # both BatchNorm backwards in one pair of passes and leaves (dz, dgamma, dbeta) here


class BnLink:
    """Joins a conv + BatchNorm (+ ReLU) unit WITHOUT residual (the producer) to the one convolution that consumes its output
    (the consumer).  The consumer's data gradient IS the producer's incoming gradient, so the consumer's data-gradient kernel
    also takes the producer's two BatchNorm-backward sums (hp_conv3d_backward_data_bnsums, from the tile in hand and one read
    of the producer's raw output) and leaves them here; the producer's backward then skips its reduction pass over dy and z
    (hp_bn_backward_presummed).  Exact-fp32 tensors and whole-tile geometries only; otherwise `sums` stays None and nothing
    changes.  HP_BN_FUSE=0 switches the hand-over off (A/B runs)."""
    __slots__ = ("src", "sums")

    def __init__(self):
        self.src = None    # (z, mean, rstd, gamma, beta, relu, byte mask or None) of the producer, set by its forward
        self.sums = None   # HP_STATS_SLOTS x 2C doubles, set by the consumer's backward


# A Bottleneck's OUTPUT unit (bn3 + identity shortcut + ReLU) is consumed by the next block: by its conv1 and by its identity
# shortcut.  The block-output gradient is complete in the epilogue of that conv1's data gradient (it adds the parked shortcut
# gradient), so conv1 of the NEXT block is the consumer of this unit's BnLink.  The two forwards do not see each other (blocks
# sit in an nn.Sequential): the producer offers its link together with a weak reference to its output tensor, and the next
# block takes it if -- and only if -- its input IS that tensor.
_res_bn_pending = [None]


# OFF by default (HP_BN_FUSE_RES=1 turns it on): measured at the headline shape, same box -- the reduction passes it removes
# are worth -8.1 ms/step, but its consumers are the 1^3 data gradients with 4 x planes output channels, which are HBM-bound on
# exactly that output, and the extra read of z_a (as large as the output) costs them +8.3 ms: 472.3 ms with it, 468.4 without
# (bn1 / bn2 hand-overs alone; 474.0 with neither).
_BN_FUSE_RES = __import__("os").environ.get("HP_BN_FUSE_RES", "0") != "0"


def offer_res_bn(y, link) -> None:
    _res_bn_pending[0] = None if (link is None or not _BN_FUSE_RES) else (__import__("weakref").ref(y), link)


def take_res_bn(x):
    p, _res_bn_pending[0] = _res_bn_pending[0], None
    return p[1] if (p is not None and p[0]() is x) else None


_BN_FUSE = __import__("os").environ.get("HP_BN_FUSE", "1") != "0"
_bn_fused_calls = [0]   # data gradients that took a BatchNorm unit's sums (tests assert the path was taken where it should be)


def bn_link():
    """A BnLink when a backward pass may follow and the hand-over is enabled, else None."""
    return BnLink() if (_BN_FUSE and torch.is_grad_enabled()) else None


# Weight gradients on a second HIP stream (default since round 4; HP_WGRAD_STREAM=0 or set_wgrad_async(False) keeps
# everything on one stream).  A weight gradient has no consumer before the optimizer (or the gradient all-reduce), so it
# runs concurrently with the rest of backward: tails of one GEMM are filled by the other's blocks and the memory-bound
# BatchNorm passes of backward's critical path overlap matrix-core work.  With two kernels in flight a kernel's duration
# describes a contended launch, so bench.py times the step with the overlap and takes its per-kernel table / roofline from
# appended steps with the mode switched off.
_WGRAD_ASYNC = bool(int(__import__("os").environ.get("HP_WGRAD_STREAM", "1")))
_WGRAD_SIDE_MODE = int(__import__("os").environ.get("HP_WGRAD_SIDE_MODE", "3"))
_WGRAD_SIDE_MB = int(__import__("os").environ.get("HP_WGRAD_SIDE_MB", "512"))
_side_streams = {}
_joined_task = [-1]
# Which top-level forward a weight's use count belongs to: `begin_forward()` (called by the regressor's forward) opens a new
# epoch, so a count left behind by a forward whose backward never ran (validation without no_grad, an exception, a discarded
# graph) cannot keep a weight off the side stream for the rest of the process; a backward that finds a count from another
# epoch takes the main stream (always safe).
_use_epoch = [0]


def begin_forward() -> None:
    _use_epoch[0] += 1


def set_wgrad_async(on: bool) -> bool:
    global _WGRAD_ASYNC
    prev, _WGRAD_ASYNC = _WGRAD_ASYNC, bool(on)
    return prev


def _wgrad_side_stream(device):
    # only inside a backward pass: the join below is queued as an end-of-backward callback of that pass
    if not _WGRAD_ASYNC or torch._C._current_graph_task_id() == -1:
        return None
    s = _side_streams.get(device)
    if s is None:
        # Same priority as the training stream.  HIP offers two levels on this device (priority_range() = (0, -1)); giving the
        # training stream -- backward's critical path -- the high one, or the weight gradients the high one, measured 455.5 /
        # 455.2 ms against 456.6 (HP_MAIN_PRIO in bench.py, HP_WGRAD_PRIO here: A/B hooks): inside the noise
        s = _side_streams[device] = torch.cuda.Stream(device, priority=int(__import__("os").environ.get("HP_WGRAD_PRIO", "0")))
    task = torch._C._current_graph_task_id()
    if _joined_task[0] != task:
        _joined_task[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
    return s


# Operands of weight gradients still queued on a side stream: (event recorded behind the kernel, tensors).  The tensors stay
# referenced here -- NOT handed to the allocator with record_stream(): a record_stream'ed block is reusable only once the
# HOST has seen its event complete, and in a training loop the host runs a whole step ahead of the device, so those blocks
# were never reusable in time, the pool grew until allocation failed and every step paid for the allocator's
# free-everything-and-synchronize path (measured: 1.2 s instead of 0.47 s per headline step with no host synchronisation
# between steps).  Instead the MAIN stream waits for the weight gradient of `_HOLD_DEPTH` convolutions ago (a device-side
# wait, normally already satisfied: the side stream runs one or two kernels behind) and only then is the reference dropped,
# so the block returns to the allocator as an ordinary main-stream block, in stream order, and at most `_HOLD_DEPTH`
# convolutions' operands are held beyond their autograd lifetime.
_held = {}
_HOLD_DEPTH = int(__import__("os").environ.get("HP_WGRAD_HOLD", "4"))


def _hold(device, side, main, tensors):
    q = _held.get(device)
    if q is None:
        q = _held[device] = __import__("collections").deque()
    ev = torch.cuda.Event()
    ev.record(side)
    q.append((ev, tensors))
    while len(q) > _HOLD_DEPTH:
        ev0, _ = q.popleft()
        main.wait_event(ev0)     # whatever reuses those blocks on the main stream is ordered behind the kernel that read them


def join_side_streams():
    """The current stream of every device waits for the weight gradients queued on its side stream.  Runs by itself
    when the backward pass that queued them ends, i.e. before `backward()` returns to the caller."""
    for dev, s in _side_streams.items():
        torch.cuda.current_stream(dev).wait_stream(s)
        q = _held.get(dev)
        if q:
            q.clear()            # everything queued so far is ordered before whatever the main stream does next


def _conv_grads(desc, x, w, dz, need_dx, addend=None, addend_mask=None, bn_in=None, force_main=False):
    """(dx, dw) of z = conv(x, w) given dz; all channels-last, dw in the torch weight layout.  bn_in: the BnLink of the unit that
    produced x (see there): its BatchNorm-backward sums are taken by the data-gradient kernel where the geometry allows."""
    L = _lib.lib()
    st = _stream(x)
    dx = None
    # element types of this call's activation tensors: x / dx follow x, dy follows dz (hp_conv_desc.io)
    desc.io = (HP_IO_X | HP_IO_DX if _h(x) else 0) | (HP_IO_DY if _h(dz) else 0)
    assert addend is None or addend.dtype == x.dtype
    if need_dx:
        wh = _w_half(desc, _h(dz), desc.Cout)   # the data gradient gathers dz: Cout channels
        _, wd = _pack(desc, w, False, True, wh)
        desc.io |= HP_IO_W if wh else 0
        # strided 1^3 convolution: its gradient reaches every second voxel per axis only, so it is added into the
        # addend's own buffer (no zero-filled tensor, no copy)
        inplace = addend is not None and addend_mask is None and desc.k == 1 and desc.stride == 2 and not desc.transposed
        dx = addend if inplace else torch.empty_like(x)
        if bn_in is not None and bn_in.src is not None and not inplace and desc.io == 0 and desc.precision == 0 and x.dtype == torch.float32:
            zs, ms, rs, gs, bs, relu_s, mask_s = bn_in.src
            sums = torch.empty(_lib.STATS_SLOTS * 2 * x.shape[-1], dtype=torch.float64, device=x.device)
            fused = _C.c_int(0)
            _lib.check(L.hp_conv3d_backward_data_bnsums(_C.byref(desc), dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), _lib.ptr(addend),
                                                        _lib.ptr(addend_mask), zs.data_ptr(), ms.data_ptr(), rs.data_ptr(),
                                                        gs.data_ptr(), bs.data_ptr(), 1 if relu_s else 0, _lib.ptr(mask_s),
                                                        sums.data_ptr(), _C.byref(fused), st), "hp_conv3d_backward_data_bnsums")
            if fused.value:
                bn_in.sums = sums
                _bn_fused_calls[0] += 1
        elif addend_mask is not None:
            _lib.check(L.hp_conv3d_backward_data_masked(_C.byref(desc), dz.data_ptr(), wd.data_ptr(), dx.data_ptr(),
                                                        addend.data_ptr(), addend_mask.data_ptr(), st),
                       "hp_conv3d_backward_data_masked")
        else:
            _lib.check(L.hp_conv3d_backward_data(_C.byref(desc), dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), _lib.ptr(addend), st),
                       "hp_conv3d_backward_data")
    desc.io &= ~HP_IO_W
    n = int(L.hp_conv3d_packed_weight_elems(_C.byref(desc)))
    side = _wgrad_side_stream(x.device)
    if side is not None and _WGRAD_SIDE_MODE != 1:
        # Which weight gradients take the second stream (HP_WGRAD_SIDE_MODE; 1 = all of them: rounds 3-4): the convolutions WITH
        # taps (k > 1) -- their weight gradients are matrix-core-bound and re-read their operands through L2, so the memory-bound
        # passes of backward's critical path really run in their shadow -- and (mode 3, the default) the 1^3 layers whose operands
        # are small (x + dz below HP_WGRAD_SIDE_MB).  A LARGE 1^3 weight gradient reads x and dz exactly once at 2+ TB/s: next to
        # a BatchNorm pass that wants all of HBM the two only slow each other down.  Same box, 10 steps, twice each, headline
        # shape: all 457.7 / 457.9, mode 2 455.6 / 454.2, mode 3 454.9 / 455.0, one stream 455.7 / 455.6 ms/step; bf16s and the
        # 128^3 shape: 148.6-148.7 and 125.8-126.1 in every mode (one stream: 151.7 / 127.7).
        big = (x.numel() * x.element_size() + dz.numel() * dz.element_size()) >= (_WGRAD_SIDE_MB << 20)
        if desc.k == 1 and (_WGRAD_SIDE_MODE == 2 or big):
            side = None
    if side is not None:
        # The side-stream hand-over is safe only where nothing reads `dw` on the main stream before the join: a leaf whose
        # gradient autograd merely adopts (or that we accumulate ourselves, below).  A tensor hook receives the gradient, a
        # post-accumulate hook may read it (unless the gradient is pre-allocated, which is the accumulate path below), and a
        # weight used twice in one graph has its two gradients SUMMED by autograd on the main stream: main-stream path.
        hooked = bool(getattr(w, "_backward_hooks", None)) or \
            (bool(getattr(w, "_post_accumulate_grad_hooks", None)) and not (w.is_leaf and w.grad is not None))
        if (not w.is_leaf) or hooked or force_main:
            side = None
    if side is None:
        dwp = torch.empty(n, dtype=torch.float32, device=x.device)
        _lib.check(L.hp_conv3d_backward_weight(_C.byref(desc), x.data_ptr(), dz.data_ptr(), dwp.data_ptr(), st),
                   "hp_conv3d_backward_weight")
        if _same_as_packed(desc):
            return dx, dwp.view_as(w)
        dw = torch.empty_like(w)
        _lib.check(L.hp_conv3d_unpack_wgrad(_C.byref(desc), dwp.data_ptr(), dw.data_ptr(), st), "hp_conv3d_unpack_wgrad")
        return dx, dw
    # The weight gradient has no consumer before the optimizer: it runs on a second stream, concurrently with the
    # rest of backward on the main one (join_side_streams() before the first reader of the gradients).  Its output is
    # allocated HERE, on the main stream (written on the side stream, read and freed on the main one after the join), and its
    # operands are kept alive by _hold(): no block of this exchange is owned by, or recorded on, the side stream.
    main = torch.cuda.current_stream(x.device)
    accumulate = w.is_leaf and w.grad is not None
    dwp = torch.empty(n, dtype=torch.float32, device=x.device)
    dw = dwp.view_as(w) if _same_as_packed(desc) else torch.empty_like(w)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        sst = _stream(x)
        _lib.check(L.hp_conv3d_backward_weight(_C.byref(desc), x.data_ptr(), dz.data_ptr(), dwp.data_ptr(), sst),
                   "hp_conv3d_backward_weight")
        if not _same_as_packed(desc):
            _lib.check(L.hp_conv3d_unpack_wgrad(_C.byref(desc), dwp.data_ptr(), dw.data_ptr(), sst), "hp_conv3d_unpack_wgrad")
        if accumulate:
            # `w.grad` already holds something (gradient accumulation over micro-batches, zero_grad(set_to_none=False),
            # or a view into a GradBucketReducer bucket): autograd's AccumulateGrad would read `dw` on the MAIN stream
            # before the side stream has written it.  The sum is therefore taken here, on the side stream, and autograd
            # gets no gradient for this weight; whoever listens for gradients (the bucket reducer) is told directly.
            with torch.no_grad():
                w.grad.add_(dw)
    # (a dw that goes to autograd must NOT be held: autograd adopts a gradient tensor as `w.grad` only while nobody else
    # references it, and would otherwise clone it -- on the main stream, before the side stream has written it.  A dw that was
    # summed into w.grad above dies with this call and MUST be held: its block was allocated on the main stream.)
    keep = (x, dz, w, dwp, dw) if accumulate else ((x, dz, w) if dw.data_ptr() == dwp.data_ptr() else (x, dz, w, dwp))
    _hold(x.device, side, main, keep)
    _side_conv_calls[0] += 1
    del dwp, keep
    if accumulate:
        for fn in _side_grad_listeners:   # called with the MAIN stream current: a listener orders itself behind both
            fn(w, side)
        return dx, None
    return dx, dw


# callables (parameter, side_stream) run after a weight gradient was ACCUMULATED on the side stream (see above)
_side_grad_listeners = []


class _ConvBnAct(torch.autograd.Function):
    """y = act(BN(conv(x)) [+ res]) with the BN batch statistics reduced in the conv epilogue."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, res, bn, k, stride, pad, transposed, relu, link_in=None, link_out=None, res_link=None,
                bn_in=None, bn_out=None):
        L = _lib.lib()
        x = x.contiguous()
        ctx.use_epoch = _use_epoch[0]   # (the use count of w itself: _note_use() in the wrappers conv_bn_act / deconv_bn_relu)
        cout = w.shape[1] if transposed else w.shape[0]
        desc = _desc(x, cout, k, stride, pad, transposed)
        do, ho, wo = _out_dims(desc)
        st = _stream(x)
        act = _act_bf16 or _h(x)   # bf16 activation storage: z, y (and dz, dx in the backward) are bf16; statistics fp32
        with torch.cuda.device(x.device):
            wh = _w_half(desc, _h(x), desc.Cin)
            wf, _ = _pack(desc, w, True, False, wh)
            z = torch.empty(desc.B, do, ho, wo, cout, dtype=_BF if act else torch.float32, device=x.device)
            M = z.numel() // cout
            train = bn.training
            stats = torch.empty(_lib.STATS_SLOTS * 2 * cout, dtype=torch.float64, device=x.device) if train else None
            desc.io = (HP_IO_X if _h(x) else 0) | (HP_IO_W if wh else 0) | (HP_IO_Y if act else 0)
            _lib.check(L.hp_conv3d_forward(_C.byref(desc), x.data_ptr(), wf.data_ptr(), None, z.data_ptr(),
                                           _lib.ptr(stats), st), "hp_conv3d_forward")
            mean = torch.empty(cout, dtype=torch.float32, device=x.device)
            rstd = torch.empty_like(mean)
            if train:
                mom = 0.1 if bn.momentum is None else bn.momentum
                _lib.check(L.hp_bn_train_finalize_counted(stats.data_ptr(), M, cout, bn.eps, mom, mean.data_ptr(), rstd.data_ptr(),
                                                          bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                          bn.num_batches_tracked.data_ptr(), st), "hp_bn_train_finalize")
            else:
                _lib.check(L.hp_bn_eval_stats(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), cout, bn.eps,
                                              mean.data_ptr(), rstd.data_ptr(), st), "hp_bn_eval_stats")
            if res_link is not None and res is None and not relu:
                # Shortcut unit of a Bottleneck: its BatchNorm is applied by the unit that adds it as the residual
                # (hp_bn_apply_res_bn); the raw convolution output travels instead of a normalised copy.
                res_link.affine = (mean, rstd, gamma, beta)
                res_link.short = (z.detach(), mean, rstd, gamma, train)   # an alias, not the output object: no ctx cycle
                ctx.save_for_backward(x, w, gamma, beta, z, None, mean, rstd)
                ctx.cfg = (desc, relu, train, False)
                ctx.act = _act_bf16 or _h(x)
                ctx.links = (link_in, link_out, res_link)
                ctx.bn_links = (bn_in, None)
                return z
            y = torch.empty_like(z, dtype=_BF if act else torch.float32)
            bio = (HP_BN_ACT | HP_BN_Z) if act else 0
            if res is not None:
                res = res.contiguous()
                assert (res_link is not None and res_link.affine is not None) or _h(res) == act, "residual / output element types differ"
            # with a residual the backward needs the sign pattern of the output: a byte per channel quad, written here
            mask = (torch.empty(M * cout // 4, dtype=torch.uint8, device=x.device)
                    if (res is not None and relu and any(ctx.needs_input_grad)) else None)
            raff = res_link.affine if (res_link is not None and res is not None) else None
            if raff is not None:
                res_link.affine = None
                _lib.check(L.hp_bn_apply_res_bn(z.data_ptr(), res.data_ptr(), y.data_ptr(), M, cout, mean.data_ptr(),
                                                rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1 if relu else 0,
                                                _lib.ptr(mask), raff[0].data_ptr(), raff[1].data_ptr(), raff[2].data_ptr(),
                                                raff[3].data_ptr(), bio, st), "hp_bn_apply_res_bn")
            else:
                _lib.check(L.hp_bn_apply(z.data_ptr(), _lib.ptr(res), y.data_ptr(), M, cout, mean.data_ptr(), rstd.data_ptr(),
                                         gamma.data_ptr(), beta.data_ptr(), 1 if relu else 0, _lib.ptr(mask), bio, st), "hp_bn_apply")
        # without a residual the mask is rebuilt from z; y itself is kept only by the consumer that reads it as input
        ctx.save_for_backward(x, w, gamma, beta, z, mask, mean, rstd)
        ctx.cfg = (desc, relu, train, res is not None)
        ctx.act = act
        ctx.links = (link_in, link_out, res_link)
        ctx.bn_links = (bn_in, None)
        if bn_out is not None and not act and train and res_link is None:
            if res is None and link_out is None:
                # producer of a BnLink: a plain fp32 unit in training mode whose output gradient arrives from ONE data gradient
                bn_out.src = (z.detach(), mean, rstd, gamma.detach(), beta.detach(), relu, None)
                ctx.bn_links = (bn_in, bn_out)
            elif res is not None and relu and mask is not None and raff is None:
                # a block's output unit with an identity shortcut: the mask is the byte mask of the output (see offer_res_bn)
                bn_out.src = (z.detach(), mean, rstd, gamma.detach(), beta.detach(), True, mask)
                ctx.bn_links = (bn_in, bn_out)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, gamma, beta, z, mask, mean, rstd = ctx.saved_tensors
        desc, relu, train, has_res = ctx.cfg
        cout = z.shape[-1]
        M = z.numel() // cout
        dy = dy.contiguous()
        st = _stream(x)
        link_in, link_out, res_link = ctx.links
        with torch.cuda.device(x.device):
            # Residual branch gradient g = dy (.) [y > 0].  When its consumer is one of our own nodes (the block's conv1
            # data gradient through link_out, or the shortcut unit through res_link) it takes dy and the byte mask
            # instead, and g is never written.
            deferred = has_res and mask is not None and (link_out is not None or res_link is not None)
            hdt = _BF if ctx.act else torch.float32     # dz (to the convolution gradients) and g (to the residual input)
            bio = (HP_BN_DY if _h(dy) else 0) | (HP_BN_DZ if ctx.act else 0) | (HP_BN_Z if _h(z) else 0)
            g = torch.empty_like(z, dtype=hdt) if (has_res and not deferred) else None
            in_mask, relu_flag = mask, relu
            if res_link is not None and not has_res and res_link.mask is not None:
                # shortcut unit (no ReLU of its own): the incoming dy still lacks the residual unit's output mask
                in_mask, relu_flag, res_link.mask = res_link.mask, True, None
            dgamma = torch.empty_like(gamma)
            dbeta = torch.empty_like(gamma)
            nws = (int(L.hp_bn_backward_workspace_bytes(cout)) + 15) // 16 * 16
            if res_link is not None and not has_res and res_link.done is not None:
                # shortcut unit whose BatchNorm backward was already run by the residual unit (dual pass below)
                (dz, dgamma, dbeta), res_link.done = res_link.done, None
            elif deferred and res_link is not None and res_link.short is not None and cout <= 1024:
                (zb, mb, rb, gb, trb), res_link.short = res_link.short, None
                dz = torch.empty_like(z, dtype=hdt)
                dzb, dgb, dbb = torch.empty_like(zb, dtype=hdt), torch.empty_like(gb), torch.empty_like(gb)
                ws = torch.empty(2 * nws // 4 + 8, dtype=torch.float32, device=x.device)
                _lib.check(L.hp_bn_backward_dual(dy.data_ptr(), mask.data_ptr(), M, cout, z.data_ptr(), dz.data_ptr(),
                                                 mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), 1 if train else 0,
                                                 dgamma.data_ptr(), dbeta.data_ptr(), zb.data_ptr(), dzb.data_ptr(), mb.data_ptr(),
                                                 rb.data_ptr(), gb.data_ptr(), 1 if trb else 0, dgb.data_ptr(), dbb.data_ptr(),
                                                 ws.data_ptr(), bio, st), "hp_bn_backward_dual")
                res_link.done = (dzb, dgb, dbb)
            elif (ctx.bn_links[1] is not None and ctx.bn_links[1].sums is not None and g is None and dy.dtype == torch.float32
                  and ((not has_res and in_mask is None) or (has_res and deferred and in_mask is mask))):
                # the consumer's data gradient already took this unit's two sums (BnLink): coefficient kernel + apply pass only
                # (a block's output unit: the apply pass gates dy with the same byte mask the sums were taken with)
                sums, ctx.bn_links[1].sums = ctx.bn_links[1].sums, None
                dz = torch.empty_like(z, dtype=hdt)
                ws = torch.empty(nws // 4 + 2, dtype=torch.float32, device=x.device)
                _lib.check(L.hp_bn_backward_presummed(dy.data_ptr(), z.data_ptr(), dz.data_ptr(), M, cout, mean.data_ptr(),
                                                      rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1 if relu_flag else 0,
                                                      1 if train else 0, dgamma.data_ptr(), dbeta.data_ptr(), _lib.ptr(in_mask),
                                                      sums.data_ptr(), ws.data_ptr(), bio, st), "hp_bn_backward_presummed")
            else:
                dz = torch.empty_like(z, dtype=hdt)
                ws = torch.empty(nws // 4 + 2, dtype=torch.float32, device=x.device)
                _lib.check(L.hp_bn_backward(dy.data_ptr(), None, z.data_ptr(), _lib.ptr(g), dz.data_ptr(), M, cout,
                                            mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                            1 if relu_flag else 0, 1 if train else 0, dgamma.data_ptr(), dbeta.data_ptr(),
                                            _lib.ptr(in_mask), ws.data_ptr(), bio, st), "hp_bn_backward")
            addend = addend_mask = None
            last = True
            if link_in is not None and ctx.needs_input_grad[0]:
                addend, addend_mask, last = link_in.take()
            # (a block input reaches two convolutions: its gradient is complete only in the one that runs last)
            dx, dw = _conv_grads(desc, x, w, dz, ctx.needs_input_grad[0], addend, addend_mask, bn_in=ctx.bn_links[0] if last else None,
                                 force_main=_WGRAD_ASYNC and _shared_use(ctx, w))
            if not last:           # first of two convolutions reading the block input: park the partial sum
                link_in.g, dx = dx, None
            gres = g
            if link_out is not None and has_res:   # identity shortcut: hand the gradient to the block's conv1 data gradient
                if deferred:
                    link_out.g, link_out.mask, gres = dy, mask, None
                else:
                    link_out.g, gres = g, None
            elif deferred:                         # shortcut unit: it receives dy itself and masks it on the fly
                res_link.mask, gres = mask, dy
        return dx, dw, dgamma, dbeta, gres, None, None, None, None, None, None, None, None, None, None, None


class _ConvBiasToNCDHW(torch.autograd.Function):
    """Final 1^3 conv of the head: channels-last in, (B, C, D, H, W) out."""

    @staticmethod
    def forward(ctx, x, w, bias, bn_in=None):
        L = _lib.lib()
        x = x.contiguous()
        ctx.bn_in = bn_in
        cout = w.shape[0]
        desc = _desc(x, cout, 1, 1, 0, False)
        st = _stream(x)
        with torch.cuda.device(x.device):
            wh = _w_half(desc, _h(x), desc.Cin)
            wf, _ = _pack(desc, w, True, False, wh)
            b, d, h, wd_, _ = x.shape
            ycl = torch.empty(b, d, h, wd_, cout, dtype=torch.float32, device=x.device)
            desc.io = (HP_IO_X if _h(x) else 0) | (HP_IO_W if wh else 0)
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
            dx, dw = _conv_grads(desc, x, w, dycl, ctx.needs_input_grad[0], bn_in=ctx.bn_in)
            dbias = dy.sum(dim=(0, 2, 3, 4))
        return dx, dw, dbias, None


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


def conv_bn_act(x, conv, bn, relu=True, residual=None, link_in=None, link_out=None, res_link=None, bn_in=None, bn_out=None):
    """x channels-last (B,D,H,W,C).  bn_in / bn_out: BnLinks to the unit that produced x / to the one consumer of the output."""
    if _WGRAD_ASYNC:
        _note_use(conv.weight)
    return _ConvBnAct.apply(x, conv.weight, bn.weight, bn.bias, residual, bn, conv.kernel_size[0], conv.stride[0],
                            conv.padding[0], False, relu, link_in, link_out, res_link, bn_in, bn_out)


def deconv_bn_relu(x, deconv, bn, bn_in=None, bn_out=None):
    if _WGRAD_ASYNC:
        _note_use(deconv.weight)
    return _ConvBnAct.apply(x, deconv.weight, bn.weight, bn.bias, None, bn, 4, 2, 1, True, True, None, None, None, bn_in, bn_out)


class _StemConvBnReluPool(torch.autograd.Function):
    """maxpool3(relu(BN(conv7(x)))) -- the normalised 64-channel full-resolution volume is never materialised:
    the pooling kernel applies BN + ReLU on the fly and the backward recomputes them from the raw conv output."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, bn):
        L = _lib.lib()
        x = x.contiguous()
        cout = w.shape[0]
        desc = _desc(x, cout, 7, 1, 3, False)
        b, d, h, wd_, _ = x.shape
        st = _stream(x)
        with torch.cuda.device(x.device):
            wf, _ = _pack(desc, w, True, False)
            z = torch.empty(b, d, h, wd_, cout, dtype=torch.float32, device=x.device)
            M = z.numel() // cout
            train = bn.training
            stats = torch.empty(_lib.STATS_SLOTS * 2 * cout, dtype=torch.float64, device=x.device) if train else None
            _lib.check(L.hp_conv3d_forward(_C.byref(desc), x.data_ptr(), wf.data_ptr(), None, z.data_ptr(), _lib.ptr(stats), st),
                       "hp_conv3d_forward")
            mean = torch.empty(cout, dtype=torch.float32, device=x.device)
            rstd = torch.empty_like(mean)
            if train:
                mom = 0.1 if bn.momentum is None else bn.momentum
                _lib.check(L.hp_bn_train_finalize_counted(stats.data_ptr(), M, cout, bn.eps, mom, mean.data_ptr(), rstd.data_ptr(),
                                                          bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                                          bn.num_batches_tracked.data_ptr(), st), "hp_bn_train_finalize")
            else:
                _lib.check(L.hp_bn_eval_stats(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), cout, bn.eps,
                                              mean.data_ptr(), rstd.data_ptr(), st), "hp_bn_eval_stats")
            pooled = torch.empty(b, d // 2, h // 2, wd_ // 2, cout, dtype=torch.float32, device=x.device)
            ws = torch.empty(int(L.hp_stem_bn_pool_workspace_bytes(cout)) // 4 + 4, dtype=torch.float32, device=x.device)
            _lib.check(L.hp_stem_bn_relu_pool_forward(z.data_ptr(), pooled.data_ptr(), b, d, h, wd_, cout, mean.data_ptr(),
                                                      rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), st),
                       "hp_stem_bn_relu_pool_forward")
            out = pooled
            if _act_bf16:   # the fused stem kernels stay fp32 (its backward compares pooled values); the consumers get a bf16 copy
                out = torch.empty_like(pooled, dtype=_BF)
                _lib.check(L.hp_cast_f32_to_bf16(pooled.data_ptr(), out.data_ptr(), pooled.numel(), st), "hp_cast_f32_to_bf16")
        ctx.save_for_backward(x, w, gamma, beta, z, pooled, mean, rstd)
        ctx.cfg = (desc, train)
        return out

    @staticmethod
    def backward(ctx, dp):
        L = _lib.lib()
        x, w, gamma, beta, z, pooled, mean, rstd = ctx.saved_tensors
        desc, train = ctx.cfg
        b, d, h, wd_, cout = z.shape
        dp = dp.contiguous()
        st = _stream(x)
        with torch.cuda.device(x.device):
            if _h(dp):
                dpf = torch.empty_like(dp, dtype=torch.float32)
                _lib.check(L.hp_cast_bf16_to_f32(dp.data_ptr(), dpf.data_ptr(), dp.numel(), st), "hp_cast_bf16_to_f32")
                dp = dpf
            dz = torch.empty_like(z)
            dgamma = torch.empty_like(gamma)
            dbeta = torch.empty_like(gamma)
            ws = torch.empty(int(L.hp_stem_bn_pool_workspace_bytes(cout)) // 4 + 4, dtype=torch.float32, device=x.device)
            _lib.check(L.hp_stem_bn_relu_pool_backward(z.data_ptr(), pooled.data_ptr(), dp.data_ptr(), dz.data_ptr(), b, d, h, wd_,
                                                       cout, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                       1 if train else 0, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), st),
                       "hp_stem_bn_relu_pool_backward")
            dx, dw = _conv_grads(desc, x, w, dz, ctx.needs_input_grad[0])
        return dx, dw, dgamma, dbeta, None


def stem_conv_bn_relu_pool(x, conv, bn):
    """x: (B,1,D,H,W) -- identical in memory to channels-last with C = 1."""
    _need_cuda(x, "posenet3d_50")
    b, c, d, h, w = x.shape
    assert c == 1
    return _StemConvBnReluPool.apply(x.reshape(b, d, h, w, 1), conv.weight, bn.weight, bn.bias, bn)


def head_conv_to_ncdhw(x, conv, bn_in=None):
    return _ConvBiasToNCDHW.apply(x, conv.weight, conv.bias, bn_in)


# ---------------------------------------------------------------- decode + losses (rows L1-L3)
class _SoftArgmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, num_joints, W, H, D):
        _need_cuda(preds, "softmax_integral")
        heat = preds.contiguous()
        B = heat.shape[0]
        bj = B * num_joints
        assert heat.numel() == bj * D * H * W
        joints = torch.empty(B, num_joints * 3, dtype=torch.float32, device=heat.device)
        stat = torch.empty(bj * 2, dtype=torch.float32, device=heat.device)
        with torch.cuda.device(heat.device):
            _lib.check(_lib.lib().hp_softargmax_forward(heat.data_ptr(), joints.data_ptr(), stat.data_ptr(), bj, D, H, W,
                                                        _stream(heat)), "hp_softargmax_forward")
        ctx.save_for_backward(heat, joints, stat)
        ctx.dims = (bj, D, H, W)
        return joints

    @staticmethod
    def backward(ctx, gj):
        heat, joints, stat = ctx.saved_tensors
        bj, D, H, W = ctx.dims
        dheat = torch.empty_like(heat)
        with torch.cuda.device(heat.device):
            _lib.check(_lib.lib().hp_softargmax_backward(heat.data_ptr(), joints.data_ptr(), stat.data_ptr(),
                                                         gj.contiguous().data_ptr(), dheat.data_ptr(), bj, D, H, W,
                                                         _stream(heat)), "hp_softargmax_backward")
        return dheat, None, None, None, None


def softmax_integral(preds, num_joints, W, H, D):
    """soft-argmax in voxel units, order (x,y,z) = (W,H,D axis) (utils/criterion.py:96-153) -> (B, 3J)."""
    return _SoftArgmax.apply(preds, num_joints, W, H, D)


class _WeightedMse(torch.autograd.Function):
    """sum((pred - gt)^2 * w) * scale  (utils/criterion.py:156-162; scale = 1/B when size_average)."""

    @staticmethod
    def forward(ctx, pred, gt, w, scale):
        _need_cuda(pred, "weighted_mse")
        pred, gt, w = pred.contiguous().float(), gt.contiguous().float(), w.contiguous().float()
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        with torch.cuda.device(pred.device):
            _lib.check(_lib.lib().hp_weighted_mse_forward(pred.data_ptr(), gt.data_ptr(), w.data_ptr(), pred.numel(), scale,
                                                          loss.data_ptr(), _stream(pred)), "hp_weighted_mse_forward")
        ctx.save_for_backward(pred, gt, w)
        ctx.scale = scale
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        pred, gt, w = ctx.saved_tensors
        d = torch.empty_like(pred)
        gl = gl.reshape(1).contiguous().float()
        with torch.cuda.device(pred.device):
            _lib.check(_lib.lib().hp_weighted_mse_backward(pred.data_ptr(), gt.data_ptr(), w.data_ptr(), gl.data_ptr(),
                                                           pred.numel(), ctx.scale, d.data_ptr(), _stream(pred)),
                       "hp_weighted_mse_backward")
        return d, None, None, None


def weighted_mse(pred, gt, weights, size_average=True):
    return _WeightedMse.apply(pred, gt, weights, (1.0 / len(pred)) if size_average else 1.0)


def visible_projection(x):
    """VisibleNet.forward (models/feature_propagation.py:296-312): relu -> per-volume min-max normalisation -> x 1e5 ->
    the 4 largest values along depth and their depth coordinates, concatenated on the channel axis:
    (B, C, D, H, W) -> (B, 2C, 4, H, W).  Forward only (the reference builds the module for its unused 2-D backbone)."""
    _need_cuda(x, "visible_projection")
    L = _lib.lib()
    x = x.contiguous().float()
    b, c, d, h, w = x.shape
    st = _stream(x)
    with torch.cuda.device(x.device):
        r = torch.empty_like(x)
        _lib.check(L.hp_leaky_add_forward(x.data_ptr(), None, r.data_ptr(), x.numel(), 0.0, st), "hp_leaky_add_forward")  # ReLU
        nrm = torch.empty_like(x)
        keys = torch.empty(2 * b * c, dtype=torch.int64, device=x.device)
        _lib.check(L.hp_normalize_feature_forward(r.data_ptr(), nrm.data_ptr(), b * c, d * h * w, 1.0e5, keys.data_ptr(), st),
                   "hp_normalize_feature_forward")
        out = torch.empty(b, 2 * c, 4, h, w, dtype=torch.float32, device=x.device)
        for i in range(b):   # values / depths are the two channel halves of sample i
            _lib.check(L.hp_depth_top4(nrm[i].data_ptr(), out[i, :c].data_ptr(), out[i, c:].data_ptr(), c, d, h * w, st),
                       "hp_depth_top4")
    return out


def gaussian_taps(sigma: float):
    """Normalised taps OpenCV's GaussianBlur uses for float images with ksize = (0, 0): 2 * round(4 sigma) + 1 taps
    (cv2.getGaussianKernel: exp(-x^2 / (2 sigma^2)) / sum)."""
    import numpy as np

    k = int(round(sigma * 4 * 2 + 1)) | 1
    r = k // 2
    xs = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-(xs * xs) / (2.0 * sigma * sigma))
    return (g / g.sum()).astype(np.float32), r


def add_noise(meas, sigma=10.61, seed=0, poisson=True):
    """addnoise_dataset (utils/nlos_pose_dataloader_noise.py:167-172) on the device: Gaussian blur (sigma 10.61 =
    25 / 2.355 bins) of the FLATTENED measurement with a replicate border, then a Poisson draw per sample."""
    _need_cuda(meas, "add_noise")
    x = meas.contiguous().float()
    taps, r = gaussian_taps(sigma)
    t = torch.from_numpy(taps).to(x.device)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().hp_noise_blur_poisson(x.data_ptr(), y.data_ptr(), x.numel(), t.data_ptr(), r, 1 if poisson else 0,
                                                    int(seed) & 0xFFFFFFFFFFFFFFFF, _stream(x)), "hp_noise_blur_poisson")
    return y


class _BceDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, eps):
        _need_cuda(logits, "bce_dice")
        logits, targets = logits.contiguous(), targets.contiguous().float()
        acc = torch.empty(4, dtype=torch.float64, device=logits.device)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        with torch.cuda.device(logits.device):
            _lib.check(_lib.lib().hp_bce_dice_forward(logits.data_ptr(), targets.data_ptr(), logits.numel(), eps,
                                                      acc.data_ptr(), loss.data_ptr(), _stream(logits)),
                       "hp_bce_dice_forward")
        ctx.save_for_backward(logits, targets, acc)
        ctx.eps = eps
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        logits, targets, acc = ctx.saved_tensors
        d = torch.empty_like(logits)
        gl = gl.reshape(1).contiguous().float()
        with torch.cuda.device(logits.device):
            _lib.check(_lib.lib().hp_bce_dice_backward(logits.data_ptr(), targets.data_ptr(), acc.data_ptr(), gl.data_ptr(),
                                                       d.data_ptr(), logits.numel(), ctx.eps, _stream(logits)),
                       "hp_bce_dice_backward")
        return d, None, None


class _BceDiceGlobal(torch.autograd.Function):
    """BCEDice whose Dice sums span every rank's samples (utils/criterion.py:358-368 is batch-global).  The value
    returned on a rank is local-BCE-mean + 1 - Dice_global, so the mean over ranks IS the single-process loss of the
    concatenated batch; the gradient w.r.t. the local logits is scaled so that AVERAGING gradients over ranks
    (GradBucketReducer) gives exactly that loss's gradient.  One all-reduce of 3 doubles in forward, none in backward."""

    @staticmethod
    def forward(ctx, logits, targets, eps, group):
        import torch.distributed as dist

        _need_cuda(logits, "bce_dice")
        logits, targets = logits.contiguous(), targets.contiguous().float()
        L = _lib.lib()
        acc = torch.empty(4, dtype=torch.float64, device=logits.device)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        world = 1
        with torch.cuda.device(logits.device):
            st = _stream(logits)
            _lib.check(L.hp_bce_dice_partial(logits.data_ptr(), targets.data_ptr(), logits.numel(), acc.data_ptr(), st),
                       "hp_bce_dice_partial")
            if dist.is_available() and dist.is_initialized():
                world = dist.get_world_size(group)
                dist.all_reduce(acc[1:4], group=group)
            _lib.check(L.hp_bce_dice_finalize(acc.data_ptr(), logits.numel(), eps, loss.data_ptr(), _stream(logits)),
                       "hp_bce_dice_finalize")
        ctx.save_for_backward(logits, targets, acc)
        ctx.eps, ctx.world = eps, world
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        logits, targets, acc = ctx.saved_tensors
        d = torch.empty_like(logits)
        gl = gl.reshape(1).contiguous().float()
        with torch.cuda.device(logits.device):
            _lib.check(_lib.lib().hp_bce_dice_backward_scaled(logits.data_ptr(), targets.data_ptr(), acc.data_ptr(),
                                                              gl.data_ptr(), d.data_ptr(), logits.numel(), ctx.eps,
                                                              float(ctx.world), _stream(logits)),
                       "hp_bce_dice_backward_scaled")
        return d, None, None, None


def bce_dice(logits, targets, eps=1e-9, global_batch=False, group=None):
    if global_batch:
        return _BceDiceGlobal.apply(logits, targets, eps, group)
    return _BceDice.apply(logits, targets, eps)
