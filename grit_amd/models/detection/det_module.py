"""DetectionModule: 150 object queries refined by 6 deformable decoder layers over the 4 backbone levels.

Mirror of reference models/detection/det_module.py (MLP :24-35, DetectionModule :38-213,
DeformableTransformerDecoderLayer :274-349, build_det_module_with_config :366-381): identical parameter
names (decoder_layers.N.{cross_attn,self_attn,norm1..3,linear1,linear2}, bbox_embed, class_embed,
query_embed, reference_points, level_embed), identical arithmetic including the quirks the captioner
depends on: level_embed is never added, box refinement feeds `.detach()`ed references that are
4-d from the first layer on, only hs[-1] is consumed downstream.

MI355X specifics: the 150x150 self-attention runs through the fused attention kernel (the
nn.MultiheadAttention object only owns the parameters, so in_proj_weight/in_proj_bias/out_proj keys are
unchanged), and the cross-attention samples the value maps with the HIP MSDeformAttn op.
"""
import copy
import math
import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import normal_

from grit_amd.models.common.swin_model import DropPath
from grit_amd.models.ops.modules import MSDeformAttn
from grit_amd.ops.attention import attention as fused_attention
from grit_amd.ops.glue import box_refine, relu_dropout
from grit_amd.ops.layer_norm import linear_add_layer_norm
from grit_amd.ops.linear import Linear, linear, linear_relu_dropout, mark_single_use, packed_in_proj, shared_input_linears
from grit_amd.ops.msda import StackedValueMaps
from grit_amd.ops import transposed as _transposed
from grit_amd.ops import backend as _backend
from grit_amd.ops import gemm as _gemm

_SHARED_VALUE_PROJ = os.environ.get('GRIT_SHARED_VALUE_PROJ', '1') != '0'  # A/B knobs
_STACKED_VALUE_MAPS = os.environ.get('GRIT_STACKED_VALUE_MAPS', '1') != '0'
from grit_amd.utils.misc import inverse_sigmoid

_PACKED_IN_PROJ = os.environ.get("GRIT_DET_PACKED_IN_PROJ", "1") != "0"  # A/B knob (round 6): 0 = split weights, two Linear nodes
_MLP_OWN = os.environ.get("GRIT_DET_MLP_OWN", "1") != "0"  # A/B knob: 0 = torch._addmm_activation for the detached box-refinement MLP
_VALUE_DGRAD_NT = os.environ.get("GRIT_DET_VALUE_DGRAD_NT", "1") != "0"  # A/B knob: 0 = torch.mm on the concatenated weight
_QK_LINEAR = os.environ.get("GRIT_DET_QK_LINEAR", "1") != "0"  # A/B knob: 0 = F.linear for the self-attention in-projections


class MLP(nn.Module):
    """Linear-ReLU stack; the last layer is linear."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for layer in self.layers[:-1]:
            if x.is_cuda and not torch.is_grad_enabled() and x.dim() >= 2 and x.dtype == layer.weight.dtype and layer.bias is not None:
                # no autograd (box refinement is detached, evaluation): bias + ReLU in the GEMM's epilogue, one launch -- the own short-map
                # tile where its policy takes the shape (grit_gemm_bf16_nt_relu with p = 0), else the library's fused call
                x2 = x.reshape(-1, x.shape[-1])
                if (_MLP_OWN and x.dtype == torch.bfloat16 and layer.bias.dtype == torch.bfloat16 and _gemm.OWN
                        and _gemm.prefers_own_short(x2.shape[0], layer.weight.shape[0], layer.weight.shape[1])
                        and _gemm.supported(x2 if x2.is_contiguous() else x2.contiguous(), layer.weight) and layer.bias.data_ptr() % 16 == 0
                        and _backend.override() is None):
                    x = _gemm.gemm_nt_relu(x2 if x2.is_contiguous() else x2.contiguous(), layer.weight, _gemm.BIAS_RELU_DROP,
                                           bias=layer.bias, p=0.0).view(*x.shape[:-1], -1)
                    continue
                x = torch._addmm_activation(layer.bias, x2, layer.weight.t()).view(*x.shape[:-1], -1)
            else:
                x = F.relu(layer(x))
        return self.layers[-1](x)


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _get_activation_fn(activation):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[activation]
    except KeyError:
        raise RuntimeError(F"activation should be relu/gelu, not {activation}.")


class DeformableTransformerDecoderLayer(nn.Module):

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4,
                 drop_path=0.):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)  # parameter container
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else None

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        tgt2 = self.linear2(self.dropout3(self.activation(self.linear1(tgt))))
        return self.norm3(tgt + self.dropout4(tgt2))

    def query_self_attention(self, tgt, query_pos, project=True):
        """nn.MultiheadAttention(q = k = tgt + pos, v = tgt) with its own packed weights, batch-first, fused core.
        project=False returns the concatenated heads before out_proj."""
        mha = self.self_attn
        E, h = mha.embed_dim, mha.num_heads
        B, Lq, _ = tgt.shape
        qk_in = self.with_pos_embed(tgt, query_pos)
        if _PACKED_IN_PROJ and self.training and torch.is_grad_enabled() and tgt.is_cuda:
            # ONE node over the packed parameter: both row ranges of its weight / bias gradient join the bucket's grouped launch
            # (grit_amd/ops/linear.py _PackedInProjFn) instead of two library TN GEMMs, two column sums and three concatenations
            qk, v = packed_in_proj(qk_in, tgt, mha.in_proj_weight, mha.in_proj_bias)
        else:
            # split (not slices): the backward of a split is one concatenation, a slice's is a zero fill + copy + add each
            w_qk, w_v = mha.in_proj_weight.split([2 * E, E])
            b_qk, b_v = mha.in_proj_bias.split([2 * E, E])
            # (ops.linear.linear, not F.linear: its backward sums the bias gradient with the column-sum kernel -- autograd's reduce
            # kernel needs 23 us for a [4 800, 1 024] column sum, eleven of them per step)
            lin = linear if _QK_LINEAR else F.linear
            qk = lin(qk_in, w_qk, b_qk)  # one GEMM for q and k
            v = lin(tgt, w_v, b_v)
        q, k = (t.view(B, Lq, h, E // h) for t in qk.split(E, -1))
        out = fused_attention(q, k, v.view(B, Lq, h, E // h), None, scale=1.0 / math.sqrt(E // h),
                              dropout_p=mha.dropout, training=self.training)
        out = out.reshape(B, Lq, E)
        return mha.out_proj(out) if project else out

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, src_level_start_index,
                src_valid_ratios, src_padding_mask=None, value=None):
        if reference_points.shape[-1] == 4:
            ratios = src_valid_ratios if src_valid_ratios.shape[-1] == 4 else torch.cat([src_valid_ratios, src_valid_ratios], -1)
        else:
            assert reference_points.shape[-1] == 2
            ratios = src_valid_ratios[..., :2]
        reference_points = reference_points[:, :, None].float() * ratios[:, None]  # per level, fp32

        if self.drop_path is None and self.training and torch.is_grad_enabled() and tgt.is_cuda:
            # training step: each "projection -> dropout -> residual -> LayerNorm" tail (self-attention out_proj,
            # cross-attention output_proj, FFN linear2) is one autograd node (grit_amd/ops/layer_norm.py)
            def tail(inp, linear, shortcut, drop, norm):
                return linear_add_layer_norm(inp, linear, shortcut, None, norm.weight, norm.bias, norm.eps, drop.p, True)[1]
            tgt = tail(self.query_self_attention(tgt, query_pos, project=False), self.self_attn.out_proj, tgt, self.dropout2,
                       self.norm2)
            sampled = self.cross_attn(self.with_pos_embed(tgt, query_pos), reference_points, src, src_spatial_shapes,
                                      src_level_start_index, src_padding_mask, project=False, value=value)
            tgt = tail(sampled, self.cross_attn.output_proj, tgt, self.dropout1, self.norm1)
            if self.activation is F.relu:
                hidden = linear_relu_dropout(tgt, self.linear1, self.dropout3.p)  # ReLU + dropout in the GEMMs' epilogues (ops/linear.py)
            else:
                hidden = self.dropout3(self.activation(self.linear1(tgt)))
            return tail(hidden, self.linear2, tgt, self.dropout4, self.norm3)
        tgt = self.norm2(tgt + self.dropout2(self.query_self_attention(tgt, query_pos)))
        tgt2 = self.cross_attn(self.with_pos_embed(tgt, query_pos), reference_points, src, src_spatial_shapes,
                               src_level_start_index, src_padding_mask, value=value)
        if self.drop_path is None:
            return self.forward_ffn(self.norm1(tgt + self.dropout1(tgt2)))
        tgt = tgt + self.drop_path(self.dropout1(tgt2))
        tgt2 = self.linear2(self.dropout3(self.activation(self.linear1(tgt))))
        return self.norm3(tgt + self.drop_path(self.dropout4(tgt2)))


class DetectionModule(nn.Module):

    def __init__(self, d_model=256, nhead=8, num_decoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", return_intermediate_dec=True, num_feature_levels=4, dec_n_points=4, drop_path=0.,
                 num_classes=81, aux_loss=True, with_box_refine=True, num_queries=100):
        super().__init__()
        self.aux_loss, self.with_box_refine = aux_loss, with_box_refine
        self.d_model, self.nhead = d_model, nhead
        layer = DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels,
                                                  nhead, dec_n_points, drop_path=drop_path)
        self.decoder_layers = _get_clones(layer, num_decoder_layers)
        # every Linear of a decoder layer is applied once per forward pass (the value projections go through project_values
        # and are long maps anyway): their weight gradients may run beside the backward chain (grit_amd/ops/linear.py)
        for l in self.decoder_layers:
            mark_single_use(l.linear1, l.linear2, l.self_attn.out_proj, l.cross_attn.sampling_offsets,
                            l.cross_attn.attention_weights, l.cross_attn.output_proj)
        self.num_decoder_layers = num_decoder_layers
        self.return_intermediate = return_intermediate_dec
        self.reference_points = nn.Linear(d_model, 2)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.class_embed = nn.Linear(d_model, num_classes)
        self.bbox_embed = MLP(d_model, d_model, 4, 3)
        self.query_embed = nn.Embedding(num_queries, d_model * 2)

        prior = 0.01
        self.class_embed.bias.data = torch.ones(num_classes) * (-math.log((1 - prior) / prior))
        nn.init.constant_(self.bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(self.bbox_embed.layers[-1].bias.data, 0)
        if with_box_refine:
            self.class_embed = _get_clones(self.class_embed, num_decoder_layers + 1)
            self.bbox_embed = _get_clones(self.bbox_embed, num_decoder_layers + 1)
            nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        normal_(self.level_embed)

    def bbox_refine(self, bbox_embed, output, reference_points):
        """Iterative box refinement; the new references are detached (no gradient reaches bbox_embed)."""
        if bbox_embed is None:
            return reference_points
        if output.is_cuda:  # the result is detached below: no autograd graph is needed for the MLP either
            with torch.no_grad():
                fused = box_refine(bbox_embed(output), reference_points)  # one launch for the logit / add / sigmoid chain
            if fused is not None:
                return fused
        # box arithmetic in fp32 (logit / sigmoid of coordinates), whatever dtype the MLP computes in
        delta = bbox_embed(output).float()
        reference_points = reference_points.float()
        if reference_points.shape[-1] == 4:
            new = (delta + inverse_sigmoid(reference_points)).sigmoid()
        else:
            assert reference_points.shape[-1] == 2
            new = torch.cat([delta[..., :2] + inverse_sigmoid(reference_points), delta[..., 2:]], -1).sigmoid()
        return new.detach()

    def get_valid_ratio(self, mask):
        _, H, W = mask.shape
        valid_h = torch.sum(~mask[:, :, 0], 1).float() / H
        valid_w = torch.sum(~mask[:, 0, :], 1).float() / W
        return torch.stack([valid_w, valid_h], -1)

    def _level_geometry(self, shapes, device):
        """(spatial_shapes [L, 2], level_start_index [L]) int64 on `device`, built once per distinct level layout: a
        host list -> device tensor copy is a *synchronous* pageable H2D transfer that stalls the host until the whole
        backbone forward has drained, i.e. it throws away the launch lead the small-kernel head depends on."""
        cache = self.__dict__.setdefault('_geometry_cache', {})
        key = (shapes, str(device))
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=device)
            start = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
            cache[key] = (spatial_shapes, start)
        return cache[key]

    def prepare_od_inputs(self, srcs, masks, src_flatten=None, shapes=None, no_padding=False):
        """srcs: NCHW maps per level (reference call form), or `src_flatten` [B, S, C] already in the flattened
        level-major token layout together with `shapes` = ((H_0, W_0), ...).  no_padding: every mask is known to be
        all-False, so the valid ratios are 1 and no padding mask is built (two dozen tiny launches less)."""
        B = masks[0].shape[0]
        query_pos, query_tgt = torch.split(self.query_embed.weight, self.d_model, dim=1)
        query_pos = query_pos.unsqueeze(0).expand(B, -1, -1)
        query_tgt = query_tgt.unsqueeze(0).expand(B, -1, -1)
        if src_flatten is None:
            shapes = tuple(tuple(s.shape[-2:]) for s in srcs)
            src_flatten = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)  # [B, S, C]; level_embed NOT added
        spatial_shapes, level_start_index = self._level_geometry(tuple(tuple(s) for s in shapes), src_flatten.device)
        if no_padding:
            mask_flatten = None
            valid_ratios = src_flatten.new_ones((B, len(masks), 2), dtype=torch.float32)
        else:
            mask_flatten = torch.cat([m.flatten(1) for m in masks], 1)
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
        reference_points = self.reference_points(query_pos).float().sigmoid()
        reference_points = self.bbox_refine(self.bbox_embed[0], query_tgt, reference_points)
        return {
            'tgt': query_tgt,
            'src': src_flatten,
            'src_spatial_shapes': spatial_shapes,
            'src_level_start_index': level_start_index,
            'src_valid_ratios': valid_ratios,
            'src_padding_mask': mask_flatten,
            'query_pos': query_pos,
            'reference_points': reference_points,
        }

    def project_values(self, src, padding_mask):
        """value_proj of every decoder layer on the flat map.  bf16 training: ONE GEMM over the concatenated weights into a
        [B, S, layers, M, D] tensor that the layers sample in place and whose gradient the layers' backward kernels fill
        in place (grit_amd/ops/msda.py StackedValueMaps) -- the input gradient is then one GEMM as well.  Otherwise: the
        layers' projections as one node with a GEMM-accumulated input gradient (shared_input_linears)."""
        projs = [l.cross_attn.value_proj for l in self.decoder_layers]
        heads, points = self.decoder_layers[0].cross_attn.n_heads, self.decoder_layers[0].cross_attn.n_points
        levels, n = self.decoder_layers[0].cross_attn.n_levels, len(projs)
        B, S, C = src.shape
        if _STACKED_VALUE_MAPS and src.dtype == torch.bfloat16 and C // heads == 64 and levels * points <= 16 \
                and B * S * n * C * 2 < (1 << 32) and all(p.weight.dtype == src.dtype and p.bias is not None for p in projs):
            weight = torch.cat([p.weight for p in projs])  # [layers * C, C]: 3 MB, the split of its gradient is free
            bias = torch.cat([p.bias for p in projs])
            if _VALUE_DGRAD_NT:
                _transposed.refresh([weight])  # its input gradient as ONE NT product on W^T (K = layers * C) instead of the library's NN form
            stacked = linear(src, weight, bias)  # [B, S, layers * C]
            if padding_mask is not None:
                stacked = stacked.masked_fill(padding_mask[..., None], float(0))
            maps = StackedValueMaps(stacked.view(B, S, n, heads, C // heads), n)
            return [(maps, l) for l in range(n)]
        return shared_input_linears(src, projs)

    def forward(self, srcs, masks, no_padding=False, src_flatten=None, shapes=None, last_only=False):
        """last_only=True (the captioner: it consumes hs[-1] alone, reference models/caption/detector.py:73): returns (hs[-1], None, None)
        without stacking the intermediate outputs and without the last layer's box refinement, whose result nobody reads.
        no_padding=True (caller knows every mask is all-False) drops the padding mask handed to MSDeformAttn, whose
        masked_fill would be a full copy of each value map that changes nothing.  `src_flatten` / `shapes`: the levels
        already flattened into one [B, S, C] map (grit_amd.ops.group_norm writes it directly), `srcs` is then unused."""
        od = self.prepare_od_inputs(srcs, masks, src_flatten, shapes, no_padding)
        if self.training and torch.is_grad_enabled() and od['src'].is_cuda:
            _transposed.refresh_linears(self)  # W^T of every Linear: the short maps' input gradients as NT products (ops/gemm.py)
        init_reference_out = od['reference_points']
        hs, refs = [od['tgt']], [init_reference_out]
        values = None
        if _SHARED_VALUE_PROJ and self.training and torch.is_grad_enabled() and od['src'].is_cuda and all(l.drop_path is None for l in self.decoder_layers):
            values = self.project_values(od['src'], od['src_padding_mask'])
        if od['reference_points'].shape[-1] == 4:  # the (w, h) part of the ratios once, not once per layer
            od['src_valid_ratios'] = torch.cat([od['src_valid_ratios'], od['src_valid_ratios']], -1)
        n = len(self.decoder_layers)
        for lid, layer in enumerate(self.decoder_layers):
            od['tgt'] = layer(**od) if values is None else layer(value=values[lid], **od)
            if last_only and lid == n - 1:
                return od['tgt'], None, None
            refine = self.bbox_embed[lid + 1] if self.bbox_embed is not None else None
            od['reference_points'] = self.bbox_refine(refine, od['tgt'], od['reference_points'])
            hs.append(od['tgt'])
            refs.append(od['reference_points'])
        if self.return_intermediate:
            return torch.stack(hs), init_reference_out, torch.stack(refs)
        return od['tgt'], init_reference_out, od['reference_points']


def build_det_module_with_config(cfg):
    return DetectionModule(
        d_model=cfg.d_model,
        nhead=cfg.num_heads,
        num_decoder_layers=cfg.num_layers,
        dim_feedforward=cfg.dim_feedforward,
        dropout=cfg.dropout,
        activation=cfg.activation,
        num_classes=cfg.num_classes,
        num_feature_levels=cfg.num_levels,
        dec_n_points=cfg.num_points,
        num_queries=cfg.num_queries,
        return_intermediate_dec=cfg.return_intermediate,
        aux_loss=getattr(cfg, 'aux_loss', False),
        with_box_refine=cfg.with_box_refine,
    )
