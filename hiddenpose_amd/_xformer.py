"""Building blocks shared by the transformer heads (transformer.TimeSformer, tokenpose.TokenPose_L_base): every
arithmetic step is a kernel of libhiddenpose_hip.so (Linear = the MFMA GEMM, LayerNorm, qkv split + rotary,
flash-style attention, GEGLU / GELU); PyTorch only owns the buffers.  Inference (no autograd): the reference has no
training loop for these orphan heads."""
from __future__ import annotations

import torch

from . import _lib

PREC = {"fp32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3}


def _st(t):
    return _lib.current_stream_handle(t.device)


def linear(x2d, weight, bias=None, precision=0, residual=None):
    """y = x @ W^T + b (+ residual, accumulated in place into `residual`)."""
    M, K = x2d.shape
    N = weight.shape[0]
    y = residual if residual is not None else torch.empty(M, N, dtype=torch.float32, device=x2d.device)
    _lib.check(_lib.lib().hp_linear_forward(x2d.data_ptr(), weight.data_ptr(), _lib.ptr(bias), _lib.ptr(residual), y.data_ptr(),
                                            M, K, N, precision, _st(x2d)), "hp_linear_forward")
    return y


def layernorm(x2d, norm, rows=None, rows_per_batch=0, batch_stride_rows=0):
    rows = x2d.shape[0] if rows is None else rows
    dim = x2d.shape[-1]
    y = torch.empty(rows, dim, dtype=torch.float32, device=x2d.device)
    _lib.check(_lib.lib().hp_layernorm_forward(x2d.data_ptr(), y.data_ptr(), rows, dim, norm.weight.data_ptr(), norm.bias.data_ptr(),
                                               norm.eps, rows_per_batch, batch_stride_rows, _st(x2d)), "hp_layernorm_forward")
    return y


def attention(h2d, to_qkv, b, ntok, heads, dh, nj, n, frames, scale, sin_t=None, cos_t=None, precision=0):
    """Multi-head attention over tokens [nj class / joint tokens | frames groups of n tokens]: the class tokens attend
    to every token (keys without rotary embedding), a group's tokens to [class tokens | their group] with the rotary
    tables (n, rot_dim) applied to q and k.  h2d: (b * ntok, dim) normalised input -> (b, ntok, heads * dh)."""
    L = _lib.lib()
    dev = h2d.device
    inner = heads * dh
    qkv = linear(h2d, to_qkv.weight, None, precision)
    q = torch.empty(b, heads, ntok, dh, dtype=torch.float32, device=dev)
    k, k0, v = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    rot_dim = 0 if sin_t is None else sin_t.shape[-1]
    _lib.check(L.hp_sformer_qkv_prepare(qkv.data_ptr(), q.data_ptr(), k.data_ptr(), k0.data_ptr(), v.data_ptr(), b, ntok, heads, dh, nj,
                                        n, scale, _lib.ptr(sin_t), _lib.ptr(cos_t), rot_dim, _st(h2d)), "hp_sformer_qkv_prepare")
    att = torch.empty(b, ntok, inner, dtype=torch.float32, device=dev)
    ws = torch.empty(max(1, int(L.hp_sformer_attention_workspace_bytes(b, heads, dh)) // 4), dtype=torch.float32, device=dev)
    _lib.check(L.hp_sformer_attention(q.data_ptr(), k.data_ptr(), k0.data_ptr(), v.data_ptr(), att.data_ptr(), b, heads, dh, ntok, nj, n,
                                      frames, 0, ws.data_ptr(), _st(h2d)), "hp_sformer_attention")
    return att


def geglu_ff(x2d_resid, h2d, lin_in, lin_out, precision=0):
    """x += W2 (u[:, :H] * gelu(u[:, H:])),  u = W1 h  (models/transformer.py:58-74).  With a hidden width that is a multiple
    of 64 the GEGLU rides in the first GEMM's epilogue (u is never written); otherwise Linear, hp_geglu_forward, Linear."""
    hid = lin_out.weight.shape[1]
    rows = h2d.shape[0]
    g = torch.empty(rows, hid, dtype=torch.float32, device=h2d.device)
    if hid % 64 == 0 and lin_in.weight.shape[0] == 2 * hid:
        # the Linear's own weight / bias: the kernel pairs value and gate rows inside its gather (no cached reordered copy)
        _lib.check(_lib.lib().hp_linear_geglu_forward(h2d.data_ptr(), lin_in.weight.data_ptr(), _lib.ptr(lin_in.bias), g.data_ptr(), rows,
                                                      h2d.shape[1], 2 * hid, precision, _st(h2d)), "hp_linear_geglu_forward")
    else:
        u = linear(h2d, lin_in.weight, lin_in.bias, precision)
        _lib.check(_lib.lib().hp_geglu_forward(u.data_ptr(), g.data_ptr(), rows, hid, _st(h2d)), "hp_geglu_forward")
    return linear(g, lin_out.weight, lin_out.bias, precision, residual=x2d_resid)


def gelu_ff(x2d_resid, h2d, lin_in, lin_out, precision=0):
    """x += W2 gelu(W1 h)  (models/tokenpose.py:268-281)."""
    u = linear(h2d, lin_in.weight, lin_in.bias, precision)
    _lib.check(_lib.lib().hp_gelu_forward(u.data_ptr(), u.data_ptr(), u.numel(), _st(h2d)), "hp_gelu_forward")
    return linear(u, lin_out.weight, lin_out.bias, precision, residual=x2d_resid)


def patchify(video, patch):
    """'b f c (h p1) (w p2) -> (b f h w) (p1 p2 c)'"""
    b, f, c, H, W = video.shape
    tokens = torch.empty(b * f * (H // patch) * (W // patch), patch * patch * c, dtype=torch.float32, device=video.device)
    _lib.check(_lib.lib().hp_sformer_patchify(video.data_ptr(), tokens.data_ptr(), b, f, c, H, W, patch, _st(video)), "hp_sformer_patchify")
    return tokens
