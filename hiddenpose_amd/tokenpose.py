"""TokenPose-L: keypoint tokens + patch tokens through three stacked vanilla transformers, heat-map MLP head.

Drop-in (inference) for models/tokenpose.py `TokenPose_L_base(**kwargs)` (:66-227) and `TokenPose_L(cfg)` (:30-63):
same constructor keywords, same state_dict keys, `forward(feature (b, c, H, W)) -> (b, num_keypoints, h_hm, w_hm)`.
Patchify 'b c (h p1)(w p2) -> b (h w)(p1 p2 c)' -> Linear -> [keypoint tokens | patches] (+ position embedding;
'sine-full' re-adds it to the patch tokens before every layer but the first, :311-313) -> 3 x Transformer(depth) of
{x += MHA(LN(x)); x += W2 gelu(W1 LN(x))} -> concat of the keypoint tokens of the three stages -> LayerNorm + Linear
(+ LayerNorm + Linear) -> heat-maps.  All arithmetic in libhiddenpose_hip.so (_xformer.py); `mask` is not supported."""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib
from . import _xformer as X
from . import hip_ops as ops


class _Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.Identity(), nn.Identity(), nn.Linear(hidden_dim, dim), nn.Identity())


class _Attention(nn.Module):
    def __init__(self, dim, heads=8, scale_with_head=False):
        super().__init__()
        self.heads = heads
        self.scale = (dim // heads) ** -0.5 if scale_with_head else dim ** -0.5
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim, dim), nn.Identity())


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, mlp_dim, dropout=0.0, num_keypoints=None, all_attn=False, scale_with_head=False):
        super().__init__()
        self.all_attn, self.num_keypoints = all_attn, num_keypoints
        self.layers = nn.ModuleList([
            nn.ModuleList([_Residual(_PreNorm(dim, _Attention(dim, heads, scale_with_head))),
                           _Residual(_PreNorm(dim, _FeedForward(dim, mlp_dim)))]) for _ in range(depth)])

    def run(self, x, pos, prec):
        """x (b, ntok, dim) -> new tensor (the input is kept: the three stages' keypoint tokens are concatenated)."""
        b, ntok, dim = x.shape
        rows = b * ntok
        x = x.clone()
        for idx, (attn, ff) in enumerate(self.layers):
            if idx > 0 and self.all_attn:
                patches = x[:, self.num_keypoints:]
                patches.copy_(ops.add(patches.contiguous(), pos.expand(b, -1, -1).contiguous()))
            a = attn.fn.fn
            dh = dim // a.heads
            h = X.layernorm(x.view(rows, dim), attn.fn.norm)
            # all-to-all attention = one group of ntok tokens, no class tokens, no rotary embedding
            att = X.attention(h, a.to_qkv, b, ntok, a.heads, dh, 0, ntok, 1, a.scale, None, None, prec)
            X.linear(att.view(rows, dim), a.to_out[0].weight, a.to_out[0].bias, prec, residual=x.view(rows, dim))
            h = X.layernorm(x.view(rows, dim), ff.fn.norm)
            X.gelu_ff(x.view(rows, dim), h, ff.fn.fn.net[0], ff.fn.fn.net[3], prec)
        return x


class TokenPose_L_base(nn.Module):
    linear_precision = "fp32"

    def __init__(self, *, feature_size, patch_size, num_keypoints, dim, depth, heads, mlp_dim, apply_init=False,
                 hidden_heatmap_dim=64 * 6, heatmap_dim=64 * 48, heatmap_size=(64, 48), channels=3, dropout=0.0, emb_dropout=0.0,
                 pos_embedding_type="learnable"):
        super().__init__()
        assert feature_size[0] % patch_size[0] == 0 and feature_size[1] % patch_size[1] == 0
        assert patch_size[0] == patch_size[1], "square patches (hp_sformer_patchify)"
        assert pos_embedding_type in ("sine", "learnable", "sine-full")
        _lib.lib()
        h, w = feature_size[0] // patch_size[0], feature_size[1] // patch_size[1]
        self.num_patches = h * w
        self.patch_size, self.heatmap_size, self.num_keypoints = list(patch_size), list(heatmap_size), num_keypoints
        self.pos_embedding_type = pos_embedding_type
        self.all_attn = pos_embedding_type == "sine-full"
        self.keypoint_token = nn.Parameter(torch.zeros(1, num_keypoints, dim))
        if pos_embedding_type == "learnable":
            self.pos_embedding = nn.Parameter(torch.zeros(1, self.num_patches + num_keypoints, dim))
            nn.init.trunc_normal_(self.pos_embedding, std=0.02)
        else:
            self.pos_embedding = nn.Parameter(self._sine_embedding(h, w, dim), requires_grad=False)
        self.patch_to_embedding = nn.Linear(channels * patch_size[0] * patch_size[1], dim)
        mk = lambda: Transformer(dim, depth, heads, mlp_dim, dropout, num_keypoints=num_keypoints, all_attn=self.all_attn,
                                 scale_with_head=True)
        self.transformer1, self.transformer2, self.transformer3 = mk(), mk(), mk()
        # (:111-118; the reference's first branch needs an undefined name and a configuration it never takes)
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim * 3), nn.Linear(dim * 3, heatmap_dim))
        nn.init.trunc_normal_(self.keypoint_token, std=0.02)

    @staticmethod
    def _sine_embedding(h, w, d_model, temperature=10000, scale=2 * math.pi):
        """:146-170 (constant table, built on the host once)."""
        area = torch.ones(1, h, w)
        y_embed, x_embed = area.cumsum(1, dtype=torch.float32), area.cumsum(2, dtype=torch.float32)
        half = d_model // 2
        eps = 1e-6
        y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
        x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
        dim_t = torch.arange(half, dtype=torch.float32)
        dim_t = temperature ** (2 * (dim_t // 2) / half)
        pos_x, pos_y = x_embed[:, :, :, None] / dim_t, y_embed[:, :, :, None] / dim_t
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2).flatten(2).permute(0, 2, 1).contiguous()

    @torch.no_grad()
    def forward(self, feature, mask=None):
        assert mask is None, "masks are not supported"
        if not feature.is_cuda:
            raise _lib.HiddenPoseHipError("TokenPose.forward needs a tensor on a HIP device; there is no CPU path")
        feature = feature.contiguous().float()
        b, c, H, W = feature.shape
        nk, dim = self.num_keypoints, self.keypoint_token.shape[-1]
        prec = X.PREC[self.linear_precision]
        dev = feature.device
        with torch.cuda.device(dev):
            tok = X.patchify(feature.view(b, 1, c, H, W), self.patch_size[0])
            emb = X.linear(tok, self.patch_to_embedding.weight, self.patch_to_embedding.bias).view(b, -1, dim)
            n = emb.shape[1]
            x = torch.empty(b, nk + n, dim, dtype=torch.float32, device=dev)
            x[:, :nk] = self.keypoint_token
            if self.pos_embedding_type in ("sine", "sine-full"):
                x[:, nk:] = ops.add(emb.contiguous(), self.pos_embedding[:, :n].expand(b, -1, -1).contiguous())
            else:
                x[:, nk:] = emb
                x = ops.add(x, self.pos_embedding[:, :n + nk].expand(b, -1, -1).contiguous())
            pos = self.pos_embedding
            x1 = self.transformer1.run(x, pos, prec)
            x2 = self.transformer2.run(x1, pos, prec)
            x3 = self.transformer3.run(x2, pos, prec)
            cat = torch.cat((x1[:, :nk], x2[:, :nk], x3[:, :nk]), dim=2).contiguous().view(b * nk, 3 * dim)
            y = X.layernorm(cat, self.mlp_head[0])
            y = X.linear(y, self.mlp_head[1].weight, self.mlp_head[1].bias)
        return y.view(b, nk, self.heatmap_size[0], self.heatmap_size[1])


class TokenPose_L(nn.Module):
    """models/tokenpose.py:30-63: the same network configured from a yacs-style node (cfg.MODEL.*)."""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        m = cfg.MODEL
        self.transformer = TokenPose_L_base(
            feature_size=[m.IMAGE_SIZE[1] // 4, m.IMAGE_SIZE[0] // 4], patch_size=[m.PATCH_SIZE[1], m.PATCH_SIZE[0]],
            num_keypoints=m.NUM_JOINTS, dim=m.DIM, channels=m.BASE_CHANNEL, depth=m.TRANSFORMER_DEPTH, heads=m.TRANSFORMER_HEADS,
            mlp_dim=m.DIM * m.TRANSFORMER_MLP_RATIO, apply_init=getattr(m, "INIT", False), hidden_heatmap_dim=m.HIDDEN_HEATMAP_DIM,
            heatmap_dim=m.HEATMAP_SIZE[1] * m.HEATMAP_SIZE[0], heatmap_size=[m.HEATMAP_SIZE[1], m.HEATMAP_SIZE[0]],
            pos_embedding_type=m.POS_EMBEDDING_TYPE)

    def forward(self, x):
        return self.transformer(x)
