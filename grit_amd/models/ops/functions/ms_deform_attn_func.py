"""`models.ops.functions` of the reference (ms_deform_attn_func.py:21-61) on the gfx950 op.

MSDeformAttnFunction is grit_amd.ops.msda.MSDeformAttnFunction (same apply() signature and gradient tuple).
ms_deform_attn_core_pytorch keeps its reference role -- "for debug and test only": a device-agnostic
grid_sample statement of the op that no module in this package calls."""
import torch
import torch.nn.functional as F

from grit_amd.ops import backend
from grit_amd.ops.msda import MSDeformAttnFunction  # noqa: F401


def deformable_sample(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step=64):
    """What MSDeformAttn.forward calls: the HIP Function, or the injected test implementation."""
    ov = backend.override()
    if ov is not None:
        return ov.msda(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step)
    return MSDeformAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
                                      im2col_step)


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Debug/test-only statement of the op via F.grid_sample (align_corners=False, zero padding)."""
    N_, S_, M_, D_ = value.shape
    _, Lq_, _, L_, P_, _ = sampling_locations.shape
    per_level = value.split([int(h) * int(w) for h, w in value_spatial_shapes], dim=1)
    grids = 2 * sampling_locations - 1
    cols = []
    for lid, (h, w) in enumerate(value_spatial_shapes):
        plane = per_level[lid].flatten(2).transpose(1, 2).reshape(N_ * M_, D_, int(h), int(w))
        grid = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)
        cols.append(F.grid_sample(plane, grid, mode='bilinear', padding_mode='zeros', align_corners=False))
    weights = attention_weights.transpose(1, 2).reshape(N_ * M_, 1, Lq_, L_ * P_)
    out = (torch.stack(cols, dim=-2).flatten(-2) * weights).sum(-1).view(N_, M_ * D_, Lq_)
    return out.transpose(1, 2).contiguous()
