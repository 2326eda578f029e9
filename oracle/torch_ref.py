"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Plain PyTorch (CPU, float32/float64) restatements of the three fused ops of the hot path, written to
follow the reference's own composition step by step.  Used (a) as the checker for the HIP kernels,
(b) injected through grit_amd.ops.backend.use_reference_ops() so that the host-side modules can run
on CPU in tests / config 1 / the cpu_baseline leg of bench.py.  grit_amd/ never imports this file.

Reference followed (paths into /root/reference):
  msda_core ............ models/ops/functions/ms_deform_attn_func.py:41-61 (grid_sample statement)
  window_attention ..... models/common/swin_model.py:155-186 (WindowAttention.forward),
                         :244-300 (pad / roll / partition / reverse / crop in SwinTransformerBlock.forward),
                         :424-441 (shift mask of BasicLayer.forward, value -100), :76-105 (partition/reverse)
  attention ............ models/common/attention.py:71-88 (scores / masked_fill(-inf) / softmax / dropout / PV)

Pinned by tests/test_torch_ref.py against fixtures produced by the imported reference modules
(tests/golden/make_golden.py: win_g4.npz, attn_g6.npz, msda_g1/g2.npz).
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------
# MSDA core through grid_sample
# ---------------------------------------------------------------------------------------------------
def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes]
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (v_l, (h, w)) in enumerate(zip(value.split(sizes, dim=1), spatial_shapes)):
        img = v_l.flatten(2).transpose(1, 2).reshape(B * M, D, int(h), int(w))
        grid = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)  # [B*M, Lq, P, 2]
        sampled.append(F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False))
    w = attention_weights.transpose(1, 2).reshape(B * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * w).sum(-1).view(B, M * D, Lq)
    return out.transpose(1, 2).contiguous()


def msda(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step=64):
    return msda_core(value, spatial_shapes, sampling_locations, attention_weights)


# ---------------------------------------------------------------------------------------------------
# Swin (shifted-)window attention on token-ordered q/k/v
# ---------------------------------------------------------------------------------------------------
def _partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def _reverse(windows, ws, H, W):
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def shift_mask(Hp, Wp, ws, shift, device, dtype=torch.float32):
    """[nW, N, N] additive mask, 0 / -100 (swin_model.py:424-441)."""
    img = torch.zeros((1, Hp, Wp, 1), device=device, dtype=dtype)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = _partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, float(-100.0)).masked_fill(m == 0, float(0.0))


def window_attention(qkv, rel_bias, pad_qkv, H, W, num_heads, window, shift, scale, mask=None):
    """qkv [B, H*W, 3C] in token order (as produced by `qkv` Linear on the un-partitioned map); tokens
    outside H x W (window padding) carry pad_qkv = the Linear's bias, because the reference pads the
    *normalised* map with zeros before the Linear (swin_model.py:257-262).  rel_bias [nH, N, N].
    mask: optional explicit additive mask [nW_mask, N, N] used instead of the analytic shift mask
    (WindowAttention.forward(x, mask) API, windows already partitioned).  Returns [B, H*W, C]."""
    B, T, C3 = qkv.shape
    C = C3 // 3
    hd = C // num_heads
    N = window * window
    pad_b = (window - H % window) % window
    pad_r = (window - W % window) % window
    Hp, Wp = H + pad_b, W + pad_r
    x = qkv.view(B, H, W, C3)
    if pad_b or pad_r:
        full = pad_qkv.to(qkv.dtype).view(1, 1, 1, C3).expand(B, Hp, Wp, C3).clone()
        full[:, :H, :W] = x
        x = full
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = _partition(x, window).view(-1, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)  # [3, B_, nH, N, hd]
    q, k, v = xw[0] * scale, xw[1], xw[2]
    attn = q @ k.transpose(-2, -1) + rel_bias.to(q.dtype).unsqueeze(0)
    if mask is None and shift > 0:
        mask = shift_mask(Hp, Wp, window, shift, qkv.device, q.dtype)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(-1, nW, num_heads, N, N) + mask.to(q.dtype).unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, num_heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(-1, window, window, C)
    o = _reverse(o, window, Hp, Wp)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o[:, :H, :W, :].contiguous().view(B, H * W, C)


# ---------------------------------------------------------------------------------------------------
# scaled-dot attention core of the caption side (and of nn.MultiheadAttention in the det decoder)
# ---------------------------------------------------------------------------------------------------
def attention(q, k, v, mask=None, scale=None, dropout_p=0.0, training=False):
    """q [B,Tq,H,D], k/v [B,Nk,H,D]; mask bool broadcastable to [B,H,Tq,Nk], True = masked.  -> [B,Tq,H*D]."""
    B, Tq, Hh, D = q.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    scores = torch.matmul(q.permute(0, 2, 1, 3), k.permute(0, 2, 3, 1)) * scale  # [B,H,Tq,Nk]
    if mask is not None:
        scores = scores.masked_fill(mask.bool(), float('-inf'))
    p = torch.softmax(scores, -1)
    if training and dropout_p > 0:
        p = F.dropout(p, dropout_p, True)
    out = torch.matmul(p, v.permute(0, 2, 1, 3))  # [B,H,Tq,D]
    return out.permute(0, 2, 1, 3).reshape(B, Tq, Hh * D)


def layer_norm(x, weight, bias, eps=1e-5):
    """torch.nn.functional.layer_norm over the last dimension (what nn.LayerNorm does in the reference)."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


# ---------------------------------------------------------------------------------------------------
# Flat Adam step (grit_adam_flat): torch.optim.Adam's arithmetic (torch/optim/adam.py _single_tensor_adam, amsgrad off,
# weight decay 0, the reference's optimizer: engine/caption_engine.py:59-69) on slices of the flat training state, so that
# the sharded / per-parameter-age logic of grit_amd.amp.FlatAdam can be exercised on CPU (gloo tests).
# ---------------------------------------------------------------------------------------------------
def adam_flat(master, grad, mom, var, compute, lr, b1, b2, eps, bc1, bc2s, grad_scale):
    g = grad.float() * grad_scale
    mom.mul_(b1).add_(g, alpha=1.0 - b1)
    var.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    denom = (var.sqrt() / bc2s).add_(eps)
    master.addcdiv_(mom, denom, value=-lr / bc1)
    compute.copy_(master)
