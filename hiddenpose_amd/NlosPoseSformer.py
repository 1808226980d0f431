"""NlosPoseSformer: divided space-time transformer head with joint tokens and axial RoPE.

Drop-in (inference) for models/NlosPoseSformer.py `NlosPoseSformer(**kwargs)` (:11-151): same constructor
keywords, same state_dict keys (including the time-attention weights that the reference allocates
but never runs, :66,:133-134), `forward(video (b, f, c, H, W)) -> (b, num_joints, 4, out_dim/4)`.
All arithmetic runs in libhiddenpose_hip.so (csrc/sformer_kernels.hip + the fp32 MFMA GEMM); the
reference has no training loop for this orphan head, so only the forward pass is provided.
"""
from __future__ import annotations

import ctypes as C
from math import log, pi

import torch
from torch import nn

from . import _lib
from . import _xformer as _xf


class RotaryEmbedding(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.register_buffer("inv_freqs", 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim)))


class AxialRotaryEmbedding(nn.Module):
    """:219-247.  tables(): (hp*wp, dim) sin / cos with every frequency duplicated pairwise."""

    def __init__(self, dim, max_freq=10):
        super().__init__()
        self.dim = dim
        self.register_buffer("scales", torch.logspace(0.0, log(max_freq / 2) / log(2), dim // 4, base=2))

    def tables(self, h, w, device):
        sc = self.scales.to(device=device, dtype=torch.float32)
        hs = torch.linspace(-1.0, 1.0, steps=h, device=device).unsqueeze(-1) * sc.unsqueeze(0) * pi
        ws = torch.linspace(-1.0, 1.0, steps=w, device=device).unsqueeze(-1) * sc.unsqueeze(0) * pi
        ang = torch.cat((hs[:, None, :].expand(h, w, -1), ws[None, :, :].expand(h, w, -1)), dim=-1)
        ang = ang.reshape(h * w, -1).repeat_interleave(2, dim=-1).contiguous()
        return ang.sin().contiguous(), ang.cos().contiguous()


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)


class _Attention(nn.Module):
    def __init__(self, dim, dim_head, heads):
        super().__init__()
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        inner = dim_head * heads
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Identity())


class _FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), nn.Identity(), nn.Identity(), nn.Linear(dim * mult, dim))


_LINEAR_PRECISION = {"fp32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3}  # HP_PRECISION_*


def _linear(x2d, weight, bias, precision=0, residual=None):
    """y = x @ W^T + b (+ residual, written in place into `residual`) through hp_linear_forward.  x2d (M, K) contiguous."""
    M, K = x2d.shape
    N = weight.shape[0]
    y = residual if residual is not None else torch.empty(M, N, dtype=torch.float32, device=x2d.device)
    _lib.check(_lib.lib().hp_linear_forward(x2d.data_ptr(), weight.data_ptr(), _lib.ptr(bias), _lib.ptr(residual), y.data_ptr(),
                                            M, K, N, precision, _lib.current_stream_handle(x2d.device)), "hp_linear_forward")
    return y


class NlosPoseSformer(nn.Module):
    # arithmetic of the transformer-layer Linear GEMMs: "fp32" (exact, default) or the bf16 matrix-core modes of
    # hp_conv_desc.precision; attention, LayerNorm, GEGLU, patch embedding and the output head stay fp32
    linear_precision = "fp32"
    # patch-token attention: "fp32" (exact-fp32 MFMA, default), "bf16" or "fp16" (16-bit matrix cores, fp32 soft-max; dim_head 32)
    attention_precision = "fp32"

    def __init__(self, *, dim, num_frames, num_joints=24, image_size=224, patch_size=16, channels=2, depth=12, heads=8,
                 dim_head=64, attn_dropout=0.0, ff_dropout=0.0, rotary_emb=True, out_dim=64 * 2 * 3, batch_size=2):
        super().__init__()
        assert image_size % patch_size == 0, "Image dimensions must be divisible by the patch size."
        assert rotary_emb, "only the rotary-embedding variant (config_noise.py:51) is built"
        _lib.lib()
        self.heads, self.dim_head, self.patch_size, self.num_joints = heads, dim_head, patch_size, num_joints
        patch_dim = channels * patch_size ** 2
        self.to_patch_embedding = nn.Linear(patch_dim, dim)
        self.joints_token = nn.Parameter(torch.zeros(1, num_joints, dim))
        nn.init.trunc_normal_(self.joints_token, std=0.02)
        self.frame_rot_emb = RotaryEmbedding(dim_head)
        self.image_rot_emb = AxialRotaryEmbedding(dim_head)
        self.layers = nn.ModuleList([
            nn.ModuleList([_PreNorm(dim, _Attention(dim, dim_head, heads)), _PreNorm(dim, _Attention(dim, dim_head, heads)),
                           _PreNorm(dim, _FeedForward(dim))]) for _ in range(depth)])
        self.to_out = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, out_dim))

    @torch.no_grad()
    def forward(self, video, mask=None):
        assert mask is None, "frame masks are ignored by the reference's attention (:177-179) and not supported"
        if not video.is_cuda:
            raise _lib.HiddenPoseHipError("NlosPoseSformer.forward needs a tensor on a HIP device; there is no CPU path")
        L = _lib.lib()
        video = video.contiguous().float()
        b, f, c, H, W = video.shape
        ps, nj, heads, dh = self.patch_size, self.num_joints, self.heads, self.dim_head
        hp, wp = H // ps, W // ps
        n = hp * wp
        ntok = nj + f * n
        dim = self.joints_token.shape[-1]
        dev = video.device
        st = _lib.current_stream_handle(dev)
        prec = _LINEAR_PRECISION[self.linear_precision]
        aprec = {"fp32": 0, "bf16": 1, "fp16": 4}[self.attention_precision]
        if aprec and dh != 32:
            raise _lib.HiddenPoseHipError("bf16 / fp16 attention is built for dim_head 32 only")
        with torch.cuda.device(dev):
            tokens = torch.empty(b * f * n, ps * ps * c, dtype=torch.float32, device=dev)
            _lib.check(L.hp_sformer_patchify(video.data_ptr(), tokens.data_ptr(), b, f, c, H, W, ps, st), "hp_sformer_patchify")
            emb = _linear(tokens, self.to_patch_embedding.weight, self.to_patch_embedding.bias)
            x = torch.empty(b, ntok, dim, dtype=torch.float32, device=dev)
            x[:, :nj] = self.joints_token          # token-matrix assembly: plain copies
            x[:, nj:] = emb.view(b, f * n, dim)
            sin_t, cos_t = self.image_rot_emb.tables(hp, wp, dev)
            rot_dim = sin_t.shape[-1]
            rows = b * ntok
            inner = heads * dh
            h = torch.empty_like(x)
            q = torch.empty(b, heads, ntok, dh, dtype=torch.float32, device=dev)
            k = torch.empty_like(q)
            k0 = torch.empty_like(q)
            v = torch.empty_like(q)
            att = torch.empty(b, ntok, inner, dtype=torch.float32, device=dev)
            aws = torch.empty(int(L.hp_sformer_attention_workspace_bytes(b, heads, dh)) // 4, dtype=torch.float32, device=dev)
            for _time_attn, spatial, ff in self.layers:
                a = spatial.fn
                _lib.check(L.hp_layernorm_forward(x.data_ptr(), h.data_ptr(), rows, dim, spatial.norm.weight.data_ptr(),
                                                  spatial.norm.bias.data_ptr(), spatial.norm.eps, 0, 0, st), "hp_layernorm_forward")
                qkv = _linear(h.view(rows, dim), a.to_qkv.weight, None, prec)
                _lib.check(L.hp_sformer_qkv_prepare(qkv.data_ptr(), q.data_ptr(), k.data_ptr(), k0.data_ptr(), v.data_ptr(), b, ntok, heads, dh,
                                                    nj, n, a.scale, sin_t.data_ptr(), cos_t.data_ptr(), rot_dim, st),
                           "hp_sformer_qkv_prepare")
                _lib.check(L.hp_sformer_attention(q.data_ptr(), k.data_ptr(), k0.data_ptr(), v.data_ptr(), att.data_ptr(), b, heads, dh, ntok, nj,
                                                  n, f, aprec, aws.data_ptr(), st), "hp_sformer_attention")
                _linear(att.view(rows, inner), a.to_out[0].weight, a.to_out[0].bias, prec, residual=x.view(rows, dim))
                _lib.check(L.hp_layernorm_forward(x.data_ptr(), h.data_ptr(), rows, dim, ff.norm.weight.data_ptr(),
                                                  ff.norm.bias.data_ptr(), ff.norm.eps, 0, 0, st), "hp_layernorm_forward")
                # feed-forward: Linear -> GEGLU (in the GEMM's epilogue) -> Linear + residual
                _xf.geglu_ff(x.view(rows, dim), h.view(rows, dim), ff.fn.net[0], ff.fn.net[3], prec)
            jt = torch.empty(b * nj, dim, dtype=torch.float32, device=dev)
            _lib.check(L.hp_layernorm_forward(x.data_ptr(), jt.data_ptr(), b * nj, dim, self.to_out[0].weight.data_ptr(),
                                              self.to_out[0].bias.data_ptr(), self.to_out[0].eps, nj, ntok, st), "hp_layernorm_forward")
            out = _linear(jt, self.to_out[1].weight, self.to_out[1].bias)
        return out.view(b, nj, 4, -1)
