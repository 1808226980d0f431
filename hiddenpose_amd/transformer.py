"""TimeSformer: divided space-time attention with one class token, axial / temporal rotary embeddings, GEGLU
feed-forward and optional token shift.

Drop-in (inference) for models/transformer.py `TimeSformer(**kwargs)` (:152-257): same constructor keywords, same
state_dict keys, `forward(video (b, f, c, H, W)) -> (b, 72)`.  Per layer (:251-254): x += TimeAttn(LN(x)) over the f
tokens that share a patch position; x += SpaceAttn(LN(x)) over the n patches of a frame; x += GEGLU-FF(LN(x)).  In
both attentions the class token attends to all tokens with un-rotated keys, the patch tokens to [class | their group]
(:110-141).  All arithmetic is in libhiddenpose_hip.so (see _xformer.py); the '(b n) f' regrouping of the time
attention is a transposed copy of the token matrix.  `mask` (frame masks for ragged clips) is not supported."""
from __future__ import annotations

from math import log, pi

import torch
from torch import nn

from . import _lib
from . import _xformer as X
from .NlosPoseSformer import AxialRotaryEmbedding, RotaryEmbedding, _Attention, _FeedForward, _PreNorm


class _PreTokenShift(nn.Module):
    """models/transformer.py:33-54 -- parameter-free wrapper: the first three thirds of the feature dim of the patch
    tokens are taken from the previous / same / next frame (zero at the clip ends)."""

    def __init__(self, frames, fn):
        super().__init__()
        self.frames, self.fn = frames, fn


def _token_shift(x, frames, nj=1):
    """x (b, nj + f*n, d) -> shifted copy (plain slice copies: data movement only)."""
    b, ntok, d = x.shape
    n = (ntok - nj) // frames
    p = x[:, nj:].view(b, frames, n, d)
    out = x.clone()
    o = out[:, nj:].view(b, frames, n, d)
    c = d // 3
    o[:, :, :, :c] = 0
    o[:, :-1, :, :c] = p[:, 1:, :, :c]            # shift(t, -1): frame j takes frame j + 1
    o[:, :, :, 2 * c:3 * c] = 0
    o[:, 1:, :, 2 * c:3 * c] = p[:, :-1, :, 2 * c:3 * c]   # shift(t, +1): frame j takes frame j - 1
    return out


class TimeSformer(nn.Module):
    linear_precision = "fp32"

    def __init__(self, *, dim, num_frames, num_classes=None, image_size=224, patch_size=16, channels=3, depth=12, heads=8,
                 dim_head=64, attn_dropout=0.0, ff_dropout=0.0, rotary_emb=True, shift_tokens=False):
        super().__init__()
        assert image_size % patch_size == 0, "Image dimensions must be divisible by the patch size."
        assert rotary_emb, "only the rotary-embedding variant is built"
        _lib.lib()
        self.heads, self.dim_head, self.patch_size, self.num_frames, self.shift_tokens = heads, dim_head, patch_size, num_frames, shift_tokens
        self.to_patch_embedding = nn.Linear(channels * patch_size ** 2, dim)
        self.cls_token = nn.Parameter(torch.randn(1, dim))
        self.frame_rot_emb = RotaryEmbedding(dim_head)
        self.image_rot_emb = AxialRotaryEmbedding(dim_head)
        wrap = (lambda m: _PreTokenShift(num_frames, m)) if shift_tokens else (lambda m: m)
        self.layers = nn.ModuleList([
            nn.ModuleList([_PreNorm(dim, wrap(_Attention(dim, dim_head, heads))), _PreNorm(dim, wrap(_Attention(dim, dim_head, heads))),
                           _PreNorm(dim, wrap(_FeedForward(dim)))]) for _ in range(depth)])
        self.to_out = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, 24 * 3))

    def _frame_tables(self, f, device):
        freqs = torch.arange(f, device=device, dtype=torch.float32)[:, None] * self.frame_rot_emb.inv_freqs.to(device)[None, :]
        freqs = torch.cat((freqs, freqs), dim=-1).contiguous()     # models/rotary.py:57-61
        return freqs.sin().contiguous(), freqs.cos().contiguous()

    @torch.no_grad()
    def forward(self, video, mask=None):
        assert mask is None, "frame masks are not supported"
        if not video.is_cuda:
            raise _lib.HiddenPoseHipError("TimeSformer.forward needs a tensor on a HIP device; there is no CPU path")
        video = video.contiguous().float()
        b, f, c, H, W = video.shape
        ps, heads, dh = self.patch_size, self.heads, self.dim_head
        hp, wp = H // ps, W // ps
        n = hp * wp
        ntok = 1 + f * n
        dim = self.cls_token.shape[-1]
        prec = X.PREC[self.linear_precision]
        dev = video.device
        with torch.cuda.device(dev):
            emb = X.linear(X.patchify(video, ps), self.to_patch_embedding.weight, self.to_patch_embedding.bias)
            x = torch.empty(b, ntok, dim, dtype=torch.float32, device=dev)
            x[:, :1] = self.cls_token
            x[:, 1:] = emb.view(b, f * n, dim)
            sin_s, cos_s = self.image_rot_emb.tables(hp, wp, dev)
            sin_t, cos_t = self._frame_tables(f, dev)
            rows = b * ntok
            for time_attn, spatial_attn, ff in self.layers:
                unwrap = (lambda m: m.fn) if self.shift_tokens else (lambda m: m)
                # ---- time attention: groups = the f tokens of one patch position ('b (f n) d -> (b n) f d')
                a = unwrap(time_attn.fn)
                h = X.layernorm(x.view(rows, dim), time_attn.norm).view(b, ntok, dim)
                if self.shift_tokens:
                    h = _token_shift(h, f)
                hperm = torch.empty_like(h)
                hperm[:, :1] = h[:, :1]
                hperm[:, 1:] = h[:, 1:].view(b, f, n, dim).transpose(1, 2).reshape(b, n * f, dim)
                att = X.attention(hperm.view(rows, dim), a.to_qkv, b, ntok, heads, dh, 1, f, n, a.scale, sin_t, cos_t, prec)
                back = torch.empty_like(att)
                back[:, :1] = att[:, :1]
                back[:, 1:] = att[:, 1:].view(b, n, f, heads * dh).transpose(1, 2).reshape(b, f * n, heads * dh)
                X.linear(back.view(rows, heads * dh), a.to_out[0].weight, a.to_out[0].bias, prec, residual=x.view(rows, dim))
                # ---- spatial attention: groups = the n patches of one frame
                a = unwrap(spatial_attn.fn)
                h = X.layernorm(x.view(rows, dim), spatial_attn.norm).view(b, ntok, dim)
                if self.shift_tokens:
                    h = _token_shift(h, f)
                att = X.attention(h.view(rows, dim), a.to_qkv, b, ntok, heads, dh, 1, n, f, a.scale, sin_s, cos_s, prec)
                X.linear(att.view(rows, heads * dh), a.to_out[0].weight, a.to_out[0].bias, prec, residual=x.view(rows, dim))
                # ---- feed-forward
                m = unwrap(ff.fn)
                h = X.layernorm(x.view(rows, dim), ff.norm).view(b, ntok, dim)
                if self.shift_tokens:
                    h = _token_shift(h, f)
                X.geglu_ff(x.view(rows, dim), h.view(rows, dim), m.net[0], m.net[3], prec)
            cls = X.layernorm(x.view(rows, dim), self.to_out[0], rows=b, rows_per_batch=1, batch_stride_rows=ntok)
            return X.linear(cls, self.to_out[1].weight, self.to_out[1].bias)
