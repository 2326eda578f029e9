"""MSDeformAttn module (reference models/ops/modules/ms_deform_attn.py:23-119): value / offset / weight / output
projections around the gfx950 sampling kernel.  Same constructor, parameter names, initialisation and
forward contract."""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from grit_amd.ops.linear import Linear

from grit_amd.ops.glue import sampling_geometry
from grit_amd.ops.msda import ms_deform_attn_stacked

from ..functions.ms_deform_attn_func import deformable_sample


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class MSDeformAttn(nn.Module):

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError('d_model must be divisible by n_heads, but got {} and {}'.format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: head dims 64 and 32 take the wide-load kernel; other sizes use the generic one.")
        self.im2col_step = 64  # accepted for interface parity; the HIP op does not chunk the batch
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = Linear(d_model, d_model)  # M = B*S rows: split-K weight gradient
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """Offsets start as a ring of directions per head, scaled 1..n_points; weights uniform (zero logits)."""
        constant_(self.sampling_offsets.weight.data, 0.)
        theta = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        ring = torch.stack([theta.cos(), theta.sin()], -1)
        ring = ring / ring.abs().max(-1, keepdim=True)[0]
        ring = ring.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        ring = ring * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(ring.reshape(-1))
        constant_(self.attention_weights.weight.data, 0.)
        constant_(self.attention_weights.bias.data, 0.)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, project=True, value=None):
        """query (N, Lq, C); reference_points (N, Lq, L, 2|4) in [0,1]; input_flatten (N, sum H_l*W_l, C);
        input_spatial_shapes (L, 2) int64 (H, W); input_level_start_index (L,) int64; input_padding_mask (N, S) bool
        -> (N, Lq, C).  `value`: this layer's value_proj(input_flatten) computed by the caller -- either the tensor itself
        (the padding mask is still applied here) or `(StackedValueMaps, layer)`: the decoder projected the map for all its
        layers with one GEMM (mask already applied) and this layer samples its slice in place (grit_amd/ops/msda.py)."""
        N, Len_q, _ = query.shape
        _, Len_in, _ = input_flatten.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        stacked = value if isinstance(value, tuple) else None
        if stacked is None:
            if value is None:
                value = self.value_proj(input_flatten)
            if input_padding_mask is not None:
                value = value.masked_fill(input_padding_mask[..., None], float(0))
            value = value.view(N, Len_in, M, self.d_model // M)
        # the (tiny) query-side arithmetic runs in fp32 whatever the projections' dtype: locations need sub-pixel precision
        cdt = torch.float64 if (stacked is None and value.dtype == torch.float64) else torch.float32
        fused = None
        if cdt == torch.float32 and reference_points.shape[-1] in (2, 4):
            # offsets / softmax / reference-point arithmetic below as ONE launch forward and one backward (grit_amd/ops/glue.py)
            fused = sampling_geometry(self.sampling_offsets(query), self.attention_weights(query), reference_points,
                                      input_spatial_shapes, M, L, P)
        if fused is not None:
            locations, weights = fused
            if stacked is not None:
                sampled = ms_deform_attn_stacked(stacked[0], stacked[1], input_spatial_shapes, input_level_start_index,
                                                 locations, weights)
            else:
                sampled = deformable_sample(value, input_spatial_shapes, input_level_start_index, locations, weights,
                                            self.im2col_step).to(value.dtype)
            return self.output_proj(sampled) if project else sampled
        offsets = self.sampling_offsets(query).view(N, Len_q, M, L, P, 2).to(cdt)
        weights = F.softmax(self.attention_weights(query).view(N, Len_q, M, L * P).to(cdt), -1).view(N, Len_q, M, L, P)
        reference_points = reference_points.to(cdt)
        if reference_points.shape[-1] == 2:
            wh = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1).to(cdt)
            locations = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError('Last dim of reference_points must be 2 or 4, but get {} instead.'.format(
                reference_points.shape[-1]))
        if stacked is not None:
            sampled = ms_deform_attn_stacked(stacked[0], stacked[1], input_spatial_shapes, input_level_start_index,
                                             locations, weights)
        else:
            sampled = deformable_sample(value, input_spatial_shapes, input_level_start_index, locations, weights,
                                        self.im2col_step)
            sampled = sampled.to(value.dtype)
        return self.output_proj(sampled) if project else sampled
