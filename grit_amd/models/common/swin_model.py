"""Swin-B backbone of GRIT, laid out for the fused gfx950 window-attention kernel.

Public names, constructor arguments and state-dict keys are those of the reference
(models/common/swin_model.py: Mlp :19-37, WindowAttention :108-186, SwinTransformerBlock :189-300,
PatchMerging :303-349, BasicLayer :352-456, PatchEmbed :459-499, SwinTransformer :502-672,
swin_base_win7_384 :690-720) so its checkpoints load unchanged.  What differs is the dataflow:

  * the reference pads, rolls and partitions the map, runs qkv/attention/proj per window and reverses all of
    it again -- five full-map permute copies per block plus a materialised [B_, nH, 144, 144] score tensor.
    Here `qkv` and `proj` (pointwise Linears) run on the map in its natural token order and the whole
    pad / roll(-s) / partition / bias / shift-mask / softmax / AV / reverse / roll(+s) / crop chain happens
    inside ONE kernel through address arithmetic (grit_amd.ops.window_attention).  Padded tokens are
    zeros *after* norm1 in the reference, hence their q/k/v equal the qkv bias: the kernel is handed
    that bias as `pad_qkv`;
  * the relative-position bias is gathered once per forward per block into [nH, N, N] (reference: gathered
    inside every window-attention call, :168-171) and the shift mask is never materialised;
  * stochastic depth (timm DropPath in the reference) is a local module with the same semantics.
"""
import math
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

import ctypes

from grit_amd import lib as _lib
from grit_amd.ops import backend as _backend
from grit_amd.ops.layer_norm import LayerNorm, add_layer_norm, linear_add_layer_norm, merge_layer_norm
from grit_amd.ops import transposed as _transposed
from grit_amd.ops.linear import Linear, linear, mark_single_use, park_weight_grad_for_partner
from grit_amd.ops.mlp import hidden as fused_hidden, mlp as fused_mlp, mlp_add_layer_norm
from grit_amd.ops.rel_bias import relative_position_bias, relative_position_bias_grouped
from grit_amd.ops.window_attention import window_attention


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


_POOLED_DROP_PATH = os.environ.get('GRIT_POOLED_DROP_PATH', '1') != '0'  # A/B knob
_PATCH_EMBED_FUSED = os.environ.get('GRIT_PATCH_EMBED_FUSED', '1') != '0'  # A/B knob: conv + bias + LayerNorm of PatchEmbed in one pass
_MERGE_LN = os.environ.get('GRIT_MERGE_LN', '1') != '0'  # A/B knob: patch-merging LayerNorm on the gathering kernels
# A/B knobs (round 6): d(pad_qkv) inside the qkv Linear's bias-gradient sum instead of a cast + an autograd add per block; one zero fill for the
# d(bias) | d(pad) accumulators of all blocks instead of one per block
_PAD_GRAD_VIA_BIAS = os.environ.get('GRIT_WINATTN_PAD_VIA_BIAS', '1') != '0'
_ZERO_ARENA = os.environ.get('GRIT_WINATTN_ZERO_ARENA', '1') != '0'
_GROUPED_REL_BIAS = os.environ.get('GRIT_GROUPED_REL_BIAS', '1') != '0'  # one relative-position gather launch for all blocks
_FUSED_MLP = os.environ.get('GRIT_FUSED_MLP', '1') != '0'  # A/B knob: Mlp on the fused-epilogue GEMM (grit_amd/ops/mlp.py)


class DropPath(nn.Module):
    """Per-sample stochastic depth (timm.models.layers.DropPath semantics: keep w.p. 1-p, rescale by 1/(1-p))."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask.div_(keep)

    def extra_repr(self):
        return 'drop_prob=%.3f' % self.drop_prob


class Mlp(nn.Module):

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        hidden_features = hidden_features or in_features
        self.fc1 = Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def hidden(self, x):
        """fc1 -> act -> drop: everything before fc2."""
        return self.drop(self.act(self.fc1(x)))

    def forward(self, x):
        return self.drop(self.fc2(self.hidden(x)))

    def run(self, x):
        """forward(x) through the fused-epilogue GEMM node when it applies (grit_amd/ops/mlp.py)."""
        return fused_mlp(x, self) if _FUSED_MLP else self(x)


def masked_sin_pos_encoding(x, mask, num_pos_feats, temperature=10000, scale=2 * math.pi):
    """Sine position code over the un-masked extent of each image (kept for API parity; unused by GRIT's path)."""
    half = num_pos_feats // 2
    ok = ~mask
    y = ok.cumsum(1, dtype=torch.float32)
    x_ = ok.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * scale
    x_ = x_ / (x_[:, :, -1:] + 1e-6) * scale
    freq = temperature**(2 * (torch.arange(half, dtype=torch.float32, device=x.device) // 2) / half)
    px, py = x_[..., None] / freq, y[..., None] / freq
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3)


def window_partition(x, window_size):
    """(B, H, W, C) -> (num_windows*B, ws, ws, C).  Utility only: the block forward never calls it."""
    B, H, W, C = x.shape
    x = x.view(B, H // window_size, window_size, W // window_size, window_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W):
    """Inverse of window_partition -> (B, H, W, C)."""
    B = windows.shape[0] // ((H // window_size) * (W // window_size))
    x = windows.view(B, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def _relative_position_index(wh, ww):
    """[wh*ww, wh*ww] index into the (2wh-1)(2ww-1) bias table: (dy + wh-1)*(2ww-1) + (dx + ww-1)."""
    ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing='ij')
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + wh - 1
    dx = xs[:, None] - xs[None, :] + ww - 1
    return dy * (2 * ww - 1) + dx


class WindowAttention(nn.Module):
    """Window multi-head self-attention with relative position bias; shifted or not."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim = dim
        self.window_size = window_size  # (Wh, Ww)
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads)**-0.5
        n_rel = (2 * window_size[0] - 1) * (2 * window_size[1] - 1)
        self.relative_position_bias_table = nn.Parameter(torch.zeros(n_rel, num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(*window_size))
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)
        if attn_drop > 0:
            raise NotImplementedError("attention-probability dropout is not fused (GRIT uses attn_drop_rate=0)")

    def relative_position_bias(self):
        """[nH, N, N] float32 = table[relative_position_index] (reference :168-171): one gather kernel forward, a
        sorted-position segment sum backward (grit_amd/ops/rel_bias.py) instead of torch's gather + permute + cast and
        its sort-based index_put gradient."""
        return relative_position_bias(self.relative_position_bias_table, self.relative_position_index,
                                      given=self.__dict__.pop("_grit_rel_bias", None))  # (the backbone's one gather for all blocks)

    def pad_qkv(self, dtype):
        if self.qkv.bias is None:
            return torch.zeros(3 * self.dim, dtype=dtype, device=self.qkv.weight.device)
        return self.qkv.bias.to(dtype)

    def attend_heads(self, x, H, W, shift, row_scale=None):
        """x: normalised tokens [B, H*W, C] in map order -> concatenated head outputs [B, H*W, C] (before proj).
        row_scale: (drop-path factors [B], rows per sample) of the attention branch, when the caller multiplies the branch by them: the
        gradient that comes back to qkv is then zero in the rows of dropped samples and its weight gradient skips them."""
        qkv = self.qkv(x, row_scale=row_scale) if row_scale is not None else self.qkv(x)
        bias = self.qkv.bias
        # d(pad_qkv) -- the pad rows ARE the qkv bias -- goes into the qkv Linear's own bias-gradient sum when that Linear's backward is
        # the node of grit_amd/ops/linear.py (it tags its result); the zeroed d(bias) | d(pad) workspace comes from the backbone's one fill
        owner = bias if (_PAD_GRAD_VIA_BIAS and bias is not None and getattr(qkv, "_grit_bias_node", None) is bias
                         and bias.dtype == qkv.dtype) else None
        acc = self.__dict__.pop("_grit_acc", None)
        return window_attention(qkv, self.relative_position_bias(), self.pad_qkv(qkv.dtype), H, W, self.num_heads,
                                self.window_size[0], shift, self.scale, row_scale=None if row_scale is None else row_scale[0],
                                pad_owner=owner, acc=acc)

    def attend_map(self, x, H, W, shift):
        """x: normalised tokens [B, H*W, C] in map order -> attention output [B, H*W, C] (after proj)."""
        return self.proj_drop(self.proj(self.attend_heads(x, H, W, shift)))

    def forward(self, x, mask=None):
        """Reference call form: x (num_windows*B, N, C) already partitioned, mask (nW, N, N) additive or None."""
        ws = self.window_size[0]
        qkv = self.qkv(x)
        out = window_attention(qkv, self.relative_position_bias(), self.pad_qkv(qkv.dtype), ws, ws, self.num_heads, ws,
                               0, self.scale, mask=mask)
        return self.proj_drop(self.proj(out))


class SwinTransformerBlock(nn.Module):

    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=LayerNorm):
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.dim, self.num_heads = dim, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=to_2tuple(window_size), num_heads=num_heads, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        # these Linears run once per forward pass: inside a gradient-bucket scope the reductions of their backward nodes
        # (weight-gradient partials, bias / LayerNorm sums) are left to the scope's grouped launch (grit_amd/ops/linear.py).
        # NOT attn.qkv: its bias receives a second gradient in every pass (the q / k / v rows of the window-padding tokens ARE the
        # bias: WindowAttention's pad_qkv) -- found by the deferral's own check on the first try
        mark_single_use(self.attn.proj, self.mlp.fc1, self.mlp.fc2)
        # backward reaches proj two kernels before qkv: their weight gradients share one launch (grit_amd/ops/linear.py GRIT_WGRAD_PARK)
        park_weight_grad_for_partner(self.attn.proj, self.attn.qkv)
        self.H = None
        self.W = None

    def forward(self, x, mask_matrix=None, normed=None, next_norm=None):
        """x (B, H*W, C) with self.H/self.W set by the stage.  `mask_matrix` is accepted for signature parity;
        the kernel derives the shift mask from (H, W, window, shift) itself.
        Each residual connection is fused with the LayerNorm that consumes its result (one pass over the map instead of
        two): `normed` is norm1(x) when the previous block already produced it, and with `next_norm` (the next block's
        norm1) the return value is (x, next_norm(x)) instead of x."""
        B, L, C = x.shape
        H, W = self.H, self.W
        assert L == H * W, "input feature has wrong size"
        n1 = self.norm1(x) if normed is None else normed
        if self.attn.proj_drop.p == 0. and self.mlp.drop.p == 0.:
            # output projection + residual + following LayerNorm as one node (the norm's backward kernel then also
            # yields the projection's bias gradient): attn.proj -> norm2, mlp.fc2 -> the next block's norm1
            scale_a = self._drop_path_scale(x, torch.float32)  # (drawn BEFORE the attention runs: its qkv Linear is told about it)
            heads = self.attn.attend_heads(n1, H, W, self.shift_size, row_scale=None if scale_a is None else (scale_a, L))
            x, n2 = self._residual_linear_norm(x, heads, self.attn.proj, self.norm2, scale=scale_a, drawn=True)
            if next_norm is None:
                return self._residual(x, self.mlp.run(n2))
            # fc1 + GELU + fc2 + drop-path + residual + the next block's norm1 as one node (fused-epilogue GEMMs)
            scale = self._drop_path_scale(x, torch.float32)
            fused = mlp_add_layer_norm(n2, self.mlp, x, scale, next_norm) if _FUSED_MLP else None
            if fused is not None:
                return fused
            hidden = fused_hidden(n2, self.mlp) if _FUSED_MLP else self.mlp.hidden(n2)
            return self._residual_linear_norm(x, hidden, self.mlp.fc2, next_norm, scale=scale, drawn=True)
        x, n2 = self._residual_norm(x, self.attn.attend_map(n1, H, W, self.shift_size), self.norm2)
        h = self.mlp(n2)
        if next_norm is None:
            return self._residual(x, h)
        return self._residual_norm(x, h, next_norm)

    def _drop_path_scale(self, x, dtype):
        """Per-sample stochastic-depth factors (0 or 1/keep) of this call, or None when drop-path is inactive."""
        dp = self.drop_path
        if isinstance(dp, DropPath) and dp.drop_prob > 0. and self.training:
            ready = getattr(self, '_drop_path_ready', None)
            if ready:  # drawn for the whole backbone in one go (SwinTransformer._draw_drop_path): no launch here
                scale = ready.pop()
                if scale.shape[0] == x.shape[0]:
                    return scale if scale.dtype == dtype else scale.to(dtype)
            keep = 1.0 - dp.drop_prob
            return torch.empty(x.shape[0], dtype=dtype, device=x.device).bernoulli_(keep).div_(keep)
        return None

    def _residual(self, x, branch):
        """x + drop_path(branch) as ONE elementwise kernel (addcmul with the per-sample keep/scale mask)."""
        scale = self._drop_path_scale(x, x.dtype)
        if scale is not None:
            return torch.addcmul(x, branch, scale.view(-1, 1, 1))
        return x + branch

    def _residual_linear_norm(self, x, inp, linear, norm, scale=None, drawn=False):
        """(x + drop_path(linear(inp)), norm(x + drop_path(linear(inp)))); `scale` (with drawn=True) = this call's
        stochastic-depth factors when the caller has already drawn them."""
        if not drawn:
            scale = self._drop_path_scale(x, torch.float32)
        if isinstance(norm, LayerNorm) and norm.elementwise_affine and len(norm.normalized_shape) == 1:
            return linear_add_layer_norm(inp, linear, x, scale, norm.weight, norm.bias, norm.eps)
        branch = linear(inp)
        x = x + branch if scale is None else torch.addcmul(x, branch, scale.to(x.dtype).view(-1, 1, 1))
        return x, norm(x)

    def _residual_norm(self, x, branch, norm):
        """(x + drop_path(branch), norm(x + drop_path(branch)))."""
        if isinstance(norm, LayerNorm) and norm.elementwise_affine and len(norm.normalized_shape) == 1:
            return add_layer_norm(x, branch, self._drop_path_scale(x, torch.float32), norm.weight, norm.bias, norm.eps)
        x = self._residual(x, branch)
        return x, norm(x)


class PatchMerging(nn.Module):
    """2x2 neighbourhood concat (order (0,0),(1,0),(0,1),(1,1)) -> LayerNorm(4C) -> Linear(4C -> 2C | pos_dim)."""

    def __init__(self, dim, norm_layer=LayerNorm, expand=True, pos_dim=768):
        super().__init__()
        self.dim = dim
        out_dim = 2 * dim if expand else pos_dim
        self.reduction = nn.Linear(4 * dim, out_dim, bias=False)
        self.norm = norm_layer(4 * dim)
        # dead parameters of the reference, kept so checkpoints round-trip
        self.expansion = nn.Linear(dim, out_dim, bias=False)
        self.norm2 = norm_layer(dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        if _MERGE_LN and isinstance(self.norm, LayerNorm):
            # the 2 x 2 concat is a VIEW the LayerNorm kernels address directly (grit_merge_layernorm_*): no permute + reshape copy of
            # the map, no scatter of its gradient (0.3 ms per step)
            y = merge_layer_norm(x, H, W, self.norm.weight, self.norm.bias, self.norm.eps)
            if y is not None:
                return linear(y, self.reduction.weight, None)
        x = x.view(B, H, W, C)
        if (H % 2) or (W % 2):
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        Hh, Wh = (H + 1) // 2, (W + 1) // 2
        # [B, Hh, 2(dy), Wh, 2(dx), C] -> channel blocks ordered (dy,dx) = (0,0),(1,0),(0,1),(1,1)
        x = x.view(B, Hh, 2, Wh, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, Hh * Wh, 4 * C)
        return linear(self.norm(x), self.reduction.weight, None)  # split-M weight gradient (grit_amd/ops/linear.py)


class BasicLayer(nn.Module):
    """One Swin stage: `depth` blocks alternating shift 0 / window//2, then the (always present) PatchMerging."""

    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=LayerNorm, downsample=None, last=False,
                 use_checkpoint=False):
        super().__init__()
        self.window_size, self.shift_size = window_size, window_size // 2
        self.depth, self.dim, self.use_checkpoint = depth, dim, use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if i % 2 == 0 else window_size // 2, mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer) for i in range(depth)
        ])
        self.downsample = None if downsample is None else downsample(dim=dim, norm_layer=norm_layer,
                                                                     expand=(not last))

    def attention_mask(self, H, W, device=None):
        """The reference's materialised shift mask [nW, N, N] (0 / -100).  Not used by forward()."""
        ws, s = self.window_size, self.shift_size
        Hp, Wp = int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws
        region = torch.zeros(Hp, Wp, device=device)
        edges_h, edges_w = (0, Hp - ws, Hp - s, Hp), (0, Wp - ws, Wp - s, Wp)
        for i in range(3):
            for j in range(3):
                region[edges_h[i]:edges_h[i + 1], edges_w[j]:edges_w[j + 1]] = 3 * i + j
        ids = window_partition(region.view(1, Hp, Wp, 1), ws).view(-1, ws * ws)
        diff = ids[:, None, :] - ids[:, :, None]
        return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))

    def forward(self, x, H, W):
        normed = None  # norm1 of the next block, produced together with the residual sum that feeds it
        for i, blk in enumerate(self.blocks):
            blk.H, blk.W = H, W
            if self.use_checkpoint and x.requires_grad:
                x, normed = checkpoint.checkpoint(blk, x, None, use_reentrant=False), None
            elif i + 1 < len(self.blocks):
                x, normed = blk(x, None, normed, self.blocks[i + 1].norm1)
            else:
                x = blk(x, None, normed)
        if self.downsample is None:
            return x, H, W, x, H, W
        return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2


class PatchEmbed(nn.Module):
    """4x4/4 conv patchify + LayerNorm, output (B, C, H/4, W/4)."""

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size = to_2tuple(patch_size)
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def tokens(self, x):
        """(B, 3, H, W) -> normalised patch tokens (B, H/4 * W/4, C), plus the token-map size.
        A stride-p p x p convolution is a GEMM on non-overlapping patches: unfold by a view + one permute copy of the
        (small) input, then F.linear with the conv weight flattened in its own (c, kh, kw) order.  MIOpen's bf16
        convolution for this shape is two orders of magnitude slower than the GEMM on MI355X."""
        ph, pw = self.patch_size
        H, W = x.shape[-2:]
        fused = self._fused_tokens(x)
        if fused is not None:
            return fused, H // ph, W // pw
        x = x.to(self.proj.weight.dtype)
        if W % pw or H % ph:
            x = F.pad(x, (0, (pw - W % pw) % pw, 0, (ph - H % ph) % ph))
        B, Cin, H, W = x.shape
        Wh, Ww = H // ph, W // pw
        patches = x.view(B, Cin, Wh, ph, Ww, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, Wh * Ww, Cin * ph * pw)
        t = F.linear(patches, self.proj.weight.view(self.embed_dim, -1), self.proj.bias)
        if self.norm is not None:
            t = self.norm(t)
        return t, Wh, Ww

    def _fused_tokens(self, x):
        """conv + bias + LayerNorm in one pass over the image (grit_patch_embed_ln_fwd) where it applies: a frozen / no-grad bf16
        patch embedding on the device with a 4 x 4 patch, 3 input channels, W a multiple of 64; None otherwise."""
        w = self.proj.weight
        if not (_PATCH_EMBED_FUSED and x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and self.patch_size == (4, 4) and self.in_chans == 3
                and isinstance(self.norm, LayerNorm) and self.embed_dim in (96, 128, 192) and w.dtype == torch.bfloat16
                and self.proj.bias is not None and self.norm.weight.dtype == torch.bfloat16 and x.dtype in (torch.float32, torch.bfloat16)
                and x.shape[2] % 4 == 0 and x.shape[3] % 64 == 0 and x.is_contiguous() and _backend.override() is None
                and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or self.norm.weight.requires_grad))):
            return None
        B, _, H, W = x.shape
        out = torch.empty((B, (H // 4) * (W // 4), self.embed_dim), dtype=torch.bfloat16, device=x.device)
        w2 = w.view(self.embed_dim, 48)
        ptr = lambda t_: ctypes.c_void_p(t_.data_ptr())
        with _lib.device_guard(x.device):
            st = _lib.load().grit_patch_embed_ln_fwd(ptr(x), int(x.dtype == torch.bfloat16), B, H, W, self.embed_dim, ptr(w2), ptr(self.proj.bias),
                                                     ptr(self.norm.weight), ptr(self.norm.bias), float(self.norm.eps), ptr(out),
                                                     _lib.current_stream_ptr())
        if st == 2:  # GRIT_ERR_UNSUPPORTED (shape outside the kernel): the GEMM path
            return None
        _lib.check(st, "grit_patch_embed_ln_fwd")
        return out

    def forward(self, x):
        t, Wh, Ww = self.tokens(x)
        return t.transpose(1, 2).reshape(x.shape[0], self.embed_dim, Wh, Ww)


class SwinTransformer(nn.Module):
    """Returns 4 maps (NCHW): outputs of stages 1,2,3 and the extra merged H/64 map (pos_dim channels)."""

    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=LayerNorm, ape=False, patch_norm=True,
                 out_indices=[1, 2, 3], frozen_stages=-1, use_checkpoint=False, pos_dim=768):
        super().__init__()
        self.pretrain_img_size = pretrain_img_size
        self.num_layers = len(depths)
        self.embed_dim, self.ape, self.patch_norm = embed_dim, ape, patch_norm
        self.out_indices, self.frozen_stages = out_indices, frozen_stages
        self.patch_embed = PatchEmbed(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_layer=norm_layer if patch_norm else None)
        if ape:
            size, patch = to_2tuple(pretrain_img_size), to_2tuple(patch_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, size[0] // patch[0], size[1] // patch[1]))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)

        rates = [r.item() for r in torch.linspace(0, drop_path_rate, sum(depths))]  # stochastic-depth decay
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(
                BasicLayer(dim=int(embed_dim * 2**i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                           mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                           attn_drop=attn_drop_rate, drop_path=rates[sum(depths[:i]):sum(depths[:i + 1])],
                           norm_layer=norm_layer,
                           downsample=partial(PatchMerging, pos_dim=pos_dim),  # after EVERY stage (Q5)
                           last=None if i < self.num_layers - 1 else True, use_checkpoint=use_checkpoint))
        self.num_features = [int(embed_dim * 2**i) for i in range(self.num_layers)]
        for i in out_indices:  # dead LayerNorms of the reference, kept for checkpoint keys
            self.add_module(f'norm{i}', norm_layer(self.num_features[i]))
        self._freeze_stages()
        self.pos_dim = pos_dim
        self.num_channels = self.num_features[1:] + [pos_dim]

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 1 and self.ape:
            self.absolute_pos_embed.requires_grad = False
        if self.frozen_stages >= 2:
            self.pos_drop.eval()
            for stage in self.layers[:self.frozen_stages - 1]:
                stage.eval()
                for p in stage.parameters():
                    p.requires_grad = False

    def _draw_drop_path(self, B, device):
        """Stochastic-depth factors of every trainable block for this forward pass -- two per block (attention branch, MLP
        branch), each a per-sample 0 or 1/keep -- drawn with four launches for the whole backbone instead of two tiny
        launches per branch (88 per step).  Same distribution as DropPath.forward; blocks pick theirs up in call order."""
        blocks = [blk for stage in self.layers for blk in stage.blocks
                  if blk.training and isinstance(blk.drop_path, DropPath) and blk.drop_path.drop_prob > 0.]
        if not blocks:
            return
        keep = getattr(self, '_drop_path_keep', None)
        if keep is None or keep.device != device or keep.shape[0] != 2 * len(blocks):
            probs = [1.0 - blk.drop_path.drop_prob for blk in blocks for _ in range(2)]
            keep = self._drop_path_keep = torch.tensor(probs, dtype=torch.float32).to(device)[:, None]  # once per device
        # `drop_path_uniforms` ([2 * blocks, B] in [0, 1), attention row then Mlp row per block): the draw itself, injectable so that two
        # runs -- bf16 with the skipped tiles, fp32 kernels, skip paths off -- see the SAME per-sample keep mask (tests/test_drop_path_gpu.py)
        u = getattr(self, 'drop_path_uniforms', None)
        if u is None:
            u = torch.rand(keep.shape[0], B, device=device)
        elif tuple(u.shape) != (keep.shape[0], B):
            raise ValueError("drop_path_uniforms must be [%d, %d], got %s" % (keep.shape[0], B, tuple(u.shape)))
        scales = (u.to(device=device, dtype=torch.float32) < keep).to(torch.float32) / keep
        for j, blk in enumerate(blocks):
            blk._drop_path_ready = [scales[2 * j + 1], scales[2 * j]]  # popped from the end: attention first

    def _hand_out_backward_workspaces(self, device):
        """One zero fill for the d(relative-position bias) | d(pad_qkv) accumulators of every trainable block's window-attention backward
        (float atomics across workgroups need zeros: 22 fills of ~5 us each inside the backward otherwise); a block's attention takes its
        slice at its next call.  Also drops a bias-gradient term a previous, interrupted backward may have left (ops/linear.py)."""
        blocks = [blk for stage in self.layers for blk in stage.blocks if blk.attn.qkv.weight.requires_grad]
        for blk in blocks:
            if blk.attn.qkv.bias is not None:
                blk.attn.qkv.bias.__dict__.pop("_grit_bias_extra", None)
        if not (_ZERO_ARENA and blocks):
            return
        sizes = [blk.attn.num_heads * blk.attn.relative_position_index.numel() + 3 * blk.attn.dim for blk in blocks]
        arena = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
        off = 0
        for blk, n in zip(blocks, sizes):
            blk.attn.__dict__["_grit_acc"] = arena[off:off + n]
            off += n

    def forward(self, x):
        B = x.shape[0]
        if _POOLED_DROP_PATH and self.training and x.is_cuda and torch.is_grad_enabled():
            self._draw_drop_path(B, x.device)
        if x.is_cuda and torch.is_grad_enabled():
            # fc2.weight^T of every block (the K-contiguous operand of the fused GELU' input-gradient GEMM): one launch here instead
            # of a transpose inside each block's backward (grit_amd/ops/transposed.py)
            # ... and the other three weights of a block: every input gradient dx = dy W of the long maps runs as an NT product on
            # W^T (own kernel or the library's NT kernel, both ahead of the NN form: grit_amd/ops/gemm.py long_input_grad)
            _transposed.refresh([w for stage in self.layers for blk in stage.blocks
                                 for w in (blk.mlp.fc2.weight, blk.attn.proj.weight, blk.attn.qkv.weight, blk.mlp.fc1.weight)
                                 if w.requires_grad])
            self._hand_out_backward_workspaces(x.device)
        if _GROUPED_REL_BIAS and x.is_cuda:
            # the relative-position gathers of all blocks in one launch (24 dependent ~5 us launches otherwise); a block's attention
            # takes its slab at its next call
            attns = [blk.attn for stage in self.layers for blk in stage.blocks]
            slabs = relative_position_bias_grouped([a.relative_position_bias_table for a in attns],
                                                   [a.relative_position_index for a in attns])
            if slabs is not None:
                for a, slab in zip(attns, slabs):
                    a.__dict__["_grit_rel_bias"] = slab
        x, Wh, Ww = self.patch_embed.tokens(x)  # (casts to the weights' dtype itself unless the fused pass reads the image as it is)
        if self.ape:
            pos = F.interpolate(self.absolute_pos_embed, size=(Wh, Ww), mode='bicubic')
            x = x + pos.flatten(2).transpose(1, 2)
        x = self.pos_drop(x)
        outs = []
        for i, stage in enumerate(self.layers):
            x_out, H, W, x, Wh, Ww = stage(x, Wh, Ww)
            if i > 0:
                outs.append(x_out.view(B, H, W, -1).permute(0, 3, 1, 2))
        outs.append(x.view(B, Wh, Ww, -1).permute(0, 3, 1, 2))
        return outs

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self


def swin_base_win7_384(pretrained=None, pos_dim=1024, **kwargs):
    """Name kept from the reference; it builds the window-12 Swin-B (embed 128, depths 2/2/18/2, heads 4/8/16/32)."""
    model = SwinTransformer(pretrain_img_size=[384, 384], embed_dim=128, depths=[2, 2, 18, 2],
                            num_heads=[4, 8, 16, 32], window_size=12, drop_path_rate=0.3,
                            pos_dim=1024 if pos_dim is None else pos_dim, **kwargs)
    if pretrained not in (None, 'none'):
        if pretrained == 'imagenet':
            pretrained = os.path.join(os.environ.get('HOME', '.'), 'checkpoints/eccv',
                                      'pretrained_weights/swin_base_patch4_window7_384_22k.pth')
            if not os.path.exists(pretrained):
                raise FileNotFoundError(f"ImageNet-22K Swin-B weights expected at {pretrained} (no network here)")
        state = torch.load(pretrained, map_location='cpu')
        model.load_state_dict(state['model'], strict=False)
    return model, 1024


def build_backbone(backbone_name='swin_base_win7_384_22k', frozen_stages=2, pre_trained='imagenet', pos_dim=None):
    if backbone_name != 'swin_base_win7_384_22k':
        raise ValueError(f'backbone {backbone_name} not supported')
    return swin_base_win7_384(pretrained=pre_trained, frozen_stages=frozen_stages, pos_dim=pos_dim)[0]
