"""Caption-side attention blocks on top of the fused gfx950 attention core.

Names / constructor arguments / parameter keys follow the reference models/common/attention.py
(Attention :25-88, MultiHeadAttention :152-184).  The score / mask / softmax / dropout / P.V chain of
Attention.forward (:71-84) is one kernel call (grit_amd.ops.attention.attention); the four Linears stay
plain GEMMs.  Memory slots (n_memories > 0, Meshed-Memory style) are never enabled by GRIT
(grid_net.n_memories is ignored) and are not implemented.
"""
import os

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from grit_amd.models.caption.containers import Module
from grit_amd.ops import backend
from grit_amd.ops import kv_cache
from grit_amd.ops import weights_epoch
from grit_amd.ops.attention import attention as fused_attention
from grit_amd.ops.layer_norm import add_layer_norm, linear_add_layer_norm
from grit_amd.ops.linear import Linear, own_or_library_linear


# GRIT_DECODE_KV_CACHE=0: step-wise decoding re-projects the raw key / value history on every step, as the reference does
_KV_CACHE = os.environ.get('GRIT_DECODE_KV_CACHE', '1') != '0'
# GRIT_DECODE_KV_APPEND=0: the cache grows by torch.cat and is re-gathered by the surviving beam like every other state
_KV_FUSED_APPEND = os.environ.get('GRIT_DECODE_KV_APPEND', '1') != '0'


def init_params(module):
    for name, param in module.named_parameters():
        if 'weight' in name:
            nn.init.xavier_uniform_(param)
        elif 'bias' in name:
            nn.init.constant_(param, 0)
        elif 'm_' in name:
            nn.init.normal_(param, mean=0, std=0.01)


class Attention(nn.Module):
    """Scaled dot-product attention with its q/k/v/o projections."""

    def __init__(self, d_model, n_heads, dropout=0.2, n_memories=0):
        super().__init__()
        if n_memories > 0:
            raise NotImplementedError("memory slots are not used by GRIT (n_memories is always 0 on its path)")
        self.fc_q = Linear(d_model, d_model)
        self.fc_k = Linear(d_model, d_model)
        self.fc_v = Linear(d_model, d_model)
        self.fc_o = Linear(d_model, d_model)
        self.dropout = nn.Dropout(p=dropout)
        self.d_model, self.n_heads, self.n_memories = d_model, n_heads, n_memories
        self.d_k = d_model // n_heads
        self.apply(init_params)
        # Step-wise decoding re-projects the (unchanging) visual memory on every step in the reference
        # (attention.py:72-73).  While `hoist_kv` is set (Transformer.beam_search does it for
        # the two cross-attentions) fc_k / fc_v results are kept for as long as the very same key tensor comes back.
        self.hoist_kv = False
        self._kv = None
        self._fused = {}

    def fused_weights(self, names):
        """Concatenated weight / bias of the named projections (('fc_q', 'fc_k', 'fc_v'): one GEMM instead of three), rebuilt
        when a parameter is replaced, written in place, or when grit_amd.ops.weights_epoch moved (flat optimizer steps and
        checkpoint loads under Bf16Compute do not touch the views' version counters).  Inference only: no gradient."""
        params = [p for n in names for p in (getattr(self, n).weight, getattr(self, n).bias)]
        if params[0].is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a captured decode graph the concatenation is part of the graph: every replay reads the parameters as they
            # are then (optimizer steps and load_state_dict write them in place), and the graph owns the tensor it reads
            with torch.no_grad():
                return torch.cat(params[0::2], 0), torch.cat(params[1::2], 0)
        tag = (weights_epoch.current(),) + tuple((p.data_ptr(), 0 if p.is_inference() else p._version, p.dtype) for p in params)
        hit = self._fused.get(names)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                hit = self._fused[names] = (tag, torch.cat(params[0::2], 0), torch.cat(params[1::2], 0))
        return hit[1], hit[2]

    def forward(self, q, k, v, attention_mask=None, project=True, q_proj=None, kv_proj=None):
        """q (b, nq, d_model), k/v (b, nk, d_model); attention_mask broadcastable to (b, h, nq, nk), True = masked.
        project=False returns the concatenated heads before fc_o (the caller fuses fc_o with what follows).
        q_proj: fc_q(q) computed by the caller (batched with other projections of the same input); kv_proj: (fc_k(k), fc_v(v))
        kept by the caller (the key / value cache of step-wise decoding)."""
        b, nq, nk, h = q.shape[0], q.shape[1], k.shape[1], self.n_heads
        bk = k.shape[0]
        qh = (self.fc_q(q) if q_proj is None else q_proj).view(b, nq, h, self.d_k)
        if bk != b:
            # beam search: the `g` beams of an image attend to ONE copy of its visual memory (Transformer.iter does not
            # replicate it per beam) -- g queries per image instead of g images with one query each; K / V are read once
            if b % bk or (attention_mask is not None and (attention_mask.shape[0] not in (1, bk) or attention_mask.shape[2] != 1)):
                raise RuntimeError("queries [%d, %d] do not group onto keys [%d, %d]" % (b, nq, bk, nk))
            qh = qh.view(bk, (b // bk) * nq, h, self.d_k)
        if kv_proj is not None:
            kh, vh = kv_proj[0].view(bk, nk, h, self.d_k), kv_proj[1].view(bk, nk, h, self.d_k)
        elif self.hoist_kv and not self.training and k is v:
            # inference tensors carry no version counter; beam search never writes the visual memory in place
            tag = (k.data_ptr(), tuple(k.shape), 0 if k.is_inference() else k._version)
            if self._kv is None or self._kv[0] != tag:
                self._kv = (tag, self.fc_k(k).view(bk, nk, h, self.d_k), self.fc_v(v).view(bk, nk, h, self.d_k))
            kh, vh = self._kv[1], self._kv[2]
        else:
            kh = self.fc_k(k).view(bk, nk, h, self.d_k)
            vh = self.fc_v(v).view(bk, nk, h, self.d_k)
        out = fused_attention(qh, kh, vh, attention_mask, scale=1.0 / np.sqrt(self.d_k), dropout_p=self.dropout.p,
                              training=self.training)
        if bk != b:
            out = out.view(b, nq, h * self.d_k)
        return self.fc_o(out) if project else out


class MultiHeadAttention(Module):
    """Attention + dropout + residual LayerNorm; optionally stateful for step-wise decoding.

    Stateful mode reproduces the reference exactly: the *raw* keys/values of every step are
    appended to running_keys / running_values and re-projected on each call."""

    def __init__(self, d_model, n_heads, dropout=.1, n_memories=0, can_be_stateful=False):
        super().__init__()
        self.attention = Attention(d_model=d_model, n_heads=n_heads, dropout=dropout, n_memories=n_memories)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model)
        self.can_be_stateful = can_be_stateful
        self._kv_deferred, self._beam_src = False, None  # fused cache update: see forward / Transformer.iter
        if can_be_stateful:
            self.register_state('running_keys', torch.zeros((1, d_model)))
            self.register_state('running_values', torch.zeros((1, d_model)))

    def forward(self, queries, keys, values, attention_mask=None, q_proj=None):
        kv_proj = None
        if self.can_be_stateful and self._is_stateful:
            cached = (_KV_CACHE and queries is keys and keys is values and queries.is_cuda and not self.training
                      and not torch.is_grad_enabled() and backend.override() is None and q_proj is None)
            if cached:
                # Inference on the device: the running states hold the PROJECTED keys / values (the usual K/V cache) and the
                # new token's q / k / v come from one GEMM.  The reference (and every other path here) appends the raw
                # inputs and re-projects the whole history on every step: the same rows through the same
                # weights -- equal up to the GEMM kernel the library picks for t times the rows.
                w, b = self.attention.fused_weights(('fc_q', 'fc_k', 'fc_v'))
                d = queries.shape[-1]
                qkv = own_or_library_linear(queries, w, b)  # (inference only: no autograd here)
                q_proj, keys, values = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
            self._kv_deferred = bool(cached and _KV_FUSED_APPEND)
            if self._kv_deferred:
                # one launch: history of the source beam (Transformer.iter hands over the surviving-beam index instead of
                # re-gathering these two states) + the new token's projected key / value
                first = self.timestep == 0
                src = None if first else self._beam_src  # [B, beam] source beam of every surviving beam, or None: same rows
                self.running_keys, self.running_values = kv_cache.append(
                    None if first else self.running_keys, None if first else self.running_values, src, keys, values,
                    beam=1 if src is None else src.shape[-1])
                self._beam_src = None
            else:
                self.running_keys = torch.cat([self.running_keys, keys], 1)
                self.running_values = torch.cat([self.running_values, values], 1)
                if self.timestep == 0:  # drop the placeholder row the state was initialised with
                    self.running_keys = self.running_keys[:, 1:]
                    self.running_values = self.running_values[:, 1:]
            keys, values = self.running_keys, self.running_values
            if cached:
                kv_proj = (keys, values)
            self.timestep += 1
        if self.training and torch.is_grad_enabled() and queries.is_cuda:
            # training step: fc_o + dropout + residual + LayerNorm as one autograd node (grit_amd/ops/layer_norm.py)
            heads = self.attention(queries, keys, values, attention_mask, project=False)
            ln = self.layer_norm
            return linear_add_layer_norm(heads, self.attention.fc_o, queries, None, ln.weight, ln.bias, ln.eps,
                                         self.dropout.p, True)[1]
        out = self.dropout(self.attention(queries, keys, values, attention_mask, q_proj=q_proj, kv_proj=kv_proj))
        ln = self.layer_norm
        return add_layer_norm(queries, out, None, ln.weight, ln.bias, ln.eps)[1]  # residual + LayerNorm in one launch on the device
