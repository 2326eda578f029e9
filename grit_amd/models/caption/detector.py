"""Caption-side Detector: Swin backbone -> masks -> grid feature (coarsest map) + region features (last decoder
layer).  Mirror of reference models/caption/detector.py (Detector :11-62, build_detector :65-84)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from grit_amd.models.common.swin_model import swin_base_win7_384
from grit_amd.models.detection.det_module import build_det_module_with_config
from grit_amd.ops.group_norm import group_norm_levels
from grit_amd.ops.linear import linear
from grit_amd.utils.misc import NestedTensor, nested_tensor_from_tensor_list


class Detector(nn.Module):

    def __init__(self, backbone, det_module=None, use_gri_feat=True, use_reg_feat=True, hidden_dim=256):
        super().__init__()
        self.backbone = backbone
        self.use_gri_feat, self.use_reg_feat = use_gri_feat, use_reg_feat
        if use_reg_feat:
            self.det_module = det_module
            self.input_proj = nn.ModuleList([
                nn.Sequential(nn.Conv2d(c, hidden_dim, kernel_size=1), nn.GroupNorm(32, hidden_dim))
                for c in backbone.num_channels
            ])

    def project_levels(self, features):
        """All feature levels through input_proj, as ONE flattened token map [B, sum_l H_l*W_l, hidden] (the layout
        DetectionModule.prepare_od_inputs builds with flatten + cat, det_module.py:172-175) plus the level shapes.
        input_proj[l] = Conv2d 1x1 + GroupNorm(32) (reference models/caption/detector.py:28-33,58).  The 1x1 convolutions run
        as GEMMs on the token views of the maps (the backbone hands out NCHW *views* of token-major tensors, so the view is
        free; MIOpen's bf16 1x1 conv + weight-gradient kernels cost ~30 ms per training step here, the GEMMs ~1 ms); the
        GroupNorms normalise token-major and write straight into the level's slice of the flat map
        (grit_amd/ops/group_norm.py) -- no NCHW round trip, no concatenation.  Same parameters, same result."""
        tokens, shapes = [], []
        for (conv, _), feature in zip(self.input_proj, features):
            B, C, H, W = feature.shape
            t = feature.permute(0, 2, 3, 1).reshape(B, H * W, C)
            # grit_amd.ops.linear: split-M weight gradient (M = 204 800 on the finest level) + streaming bias gradient
            tokens.append(linear(t, conv.weight.view(conv.out_channels, C), conv.bias))
            shapes.append((H, W))
        norms = [gn for _, gn in self.input_proj]
        flat = group_norm_levels(tokens, [gn.weight for gn in norms], [gn.bias for gn in norms], norms[0].num_groups,
                                 norms[0].eps)
        return flat, tuple(shapes)

    def forward(self, images: NestedTensor):
        """images.tensors [B,3,H,W], images.mask [B,H,W] (True on padding) ->
        {gri_feat [B,h*w,1024], gri_mask [B,1,1,h*w], reg_feat [B,150,512], reg_mask [B,1,1,150] (all False)}."""
        if isinstance(images, (list, tuple, torch.Tensor)):  # the reference's list branch is broken (Q13); fixed here
            images = nested_tensor_from_tensor_list(list(images))
        x, mask = images.tensors, images.mask
        features = self.backbone(x)
        if getattr(images, 'any_padding', None) is False:
            # the batch builder knows no pixel is padding: the down-sampled masks are all False, no need to convert and
            # resample the full-resolution mask once per level
            masks = [mask.new_zeros((mask.shape[0],) + tuple(f.shape[-2:])) for f in features]
        else:
            masks = [F.interpolate(mask[None].float(), size=f.shape[-2:]).to(torch.bool)[0] for f in features]
        out = {
            'gri_feat': features[-1].flatten(2).transpose(1, 2),
            'gri_mask': masks[-1].flatten(1)[:, None, None, :],
        }
        if not self.use_reg_feat:
            return out
        no_padding = getattr(images, 'any_padding', None) is False
        flat, shapes = self.project_levels(features)
        last, _, _ = self.det_module(None, masks, no_padding=no_padding, src_flatten=flat, shapes=shapes, last_only=True)
        out['reg_feat'] = last
        out['reg_mask'] = last.new_zeros((last.shape[0], 1, 1, last.shape[1])).bool()
        return out


def build_detector(config):
    pos_dim = getattr(config.model.detector, 'pos_dim', None)
    backbone, _ = swin_base_win7_384(frozen_stages=config.model.frozen_stages, pos_dim=pos_dim)
    det_module = build_det_module_with_config(config.model.detector) if config.model.use_reg_feat else None
    detector = Detector(backbone, det_module=det_module, hidden_dim=config.model.d_model,
                        use_gri_feat=config.model.use_gri_feat, use_reg_feat=config.model.use_reg_feat)
    ckpt = config.model.detector.checkpoint
    if ckpt and os.path.exists(ckpt):
        state = torch.load(ckpt, map_location='cpu')
        missing, unexpected = detector.load_state_dict(state['model'], strict=False)
        print(f"Loading weights for detector: missing: {len(missing)}, unexpected: {len(unexpected)}.")
    return detector
