"""Caption decoder: masked self-attention + two *parallel* cross-attentions (grid, region) gated by sigmoids.

Mirror of reference models/caption/cap_generator.py (GeneratorLayer :11-18, ParallelAttentionLayer :20-56,
CaptionGenerator :90-175).  Reproduced on purpose:
  * both gates go through `fc_alpha1`; `fc_alpha2` is a dead parameter that stays in the state dict (Q1);
  * only the 'parallel' layer exists on the reference's reachable path (Q2) -- 'concat' / 'sequential' raise;
  * stateful decoding grows `running_mask_x` / `running_seq` exactly like the reference (:134-142).
"""
import os

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from grit_amd.models.caption.containers import Module, ModuleList
from grit_amd.models.common.attention import MultiHeadAttention
from grit_amd.models.common.pos_embed import FeedForward, sinusoid_encoding_table
from grit_amd.ops import backend
from grit_amd.ops import decode_inputs
from grit_amd.ops import gate as gate_ops
from grit_amd.ops import glue
from grit_amd.ops import transposed as _transposed
from grit_amd.ops import weights_epoch
from grit_amd.ops.linear import Linear, mark_single_use, own_or_library_linear


# GRIT_FUSED_STEP_INPUTS=0: masks, step counter and embedding sum of a decoding step as the reference's separate torch ops
_FUSED_STEP_INPUTS = os.environ.get('GRIT_FUSED_STEP_INPUTS', '1') != '0'


class GeneratorLayer(Module):

    def __init__(self, d_model=512, n_heads=8, d_ff=2048, dropout=.1, n_memories=0):
        super().__init__()
        self.self_att = MultiHeadAttention(d_model, n_heads, dropout, n_memories=n_memories, can_be_stateful=True)
        self.pwff = FeedForward(d_model, d_ff, dropout)


class ParallelAttentionLayer(GeneratorLayer):

    def __init__(self, d_model=512, n_heads=8, d_ff=2048, dropout=.1, activation='sigmoid', n_memories=0):
        super().__init__(d_model=d_model, n_heads=n_heads, d_ff=d_ff, dropout=dropout, n_memories=0)
        self.vis_att1 = MultiHeadAttention(d_model, n_heads, dropout, can_be_stateful=False, n_memories=n_memories)
        self.vis_att2 = MultiHeadAttention(d_model, n_heads, dropout, can_be_stateful=False, n_memories=n_memories)
        self.fc_alpha1 = Linear(d_model + d_model, d_model)
        self.fc_alpha2 = nn.Linear(d_model + d_model, d_model)  # never used in forward (reference quirk)
        self.activation = activation
        self._q12 = None
        self.init_weights()
        # teacher forcing applies every Linear of the layer once per forward pass -- except fc_alpha1, which gates both
        # cross-attentions (two gradients that autograd adds).  Step-wise decoding with gradient suspends the declaration
        # (Transformer.beam_search, grit_amd/ops/linear.py suspend_single_use)
        mark_single_use(self.self_att, self.vis_att1, self.vis_att2, self.pwff)

    def init_weights(self):
        for fc in (self.fc_alpha1, self.fc_alpha2):
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)

    def _cross_query_weights(self):
        a, b = self.vis_att1.attention.fc_q, self.vis_att2.attention.fc_q
        if a.weight.is_cuda and torch.cuda.is_current_stream_capturing():  # part of the captured graph: see Attention.fused_weights
            with torch.no_grad():
                return torch.cat([a.weight, b.weight], 0), torch.cat([a.bias, b.bias], 0)
        tag = (weights_epoch.current(),) + tuple((p.data_ptr(), 0 if p.is_inference() else p._version, p.dtype)
                                                 for p in (a.weight, a.bias, b.weight, b.bias))
        if self._q12 is None or self._q12[0] != tag:
            with torch.no_grad():
                self._q12 = (tag, torch.cat([a.weight, b.weight], 0), torch.cat([a.bias, b.bias], 0))
        return self._q12[1], self._q12[2]

    def forward(self, x, y1, y2, mask_pad, mask_x, mask_y1, mask_y2):
        self_att = self.self_att(x, x, x, mask_x) * mask_pad
        if gate_ops.supported(self_att, self_att, self_att, mask_pad, self.fc_alpha1):
            # inference on the device: the gate arithmetic below as pack -> ONE fc_alpha1 GEMM -> fuse (grit_amd/ops/gate.py)
            d = self_att.shape[-1]
            q12 = own_or_library_linear(self_att, *self._cross_query_weights())  # fc_q of both cross-attentions: one GEMM (inference)
            enc1 = self.vis_att1(self_att, y1, y1, mask_y1, q_proj=q12[..., :d])
            enc2 = self.vis_att2(self_att, y2, y2, mask_y2, q_proj=q12[..., d:])
            return self.pwff(gate_ops.gated_merge(self_att, enc1, enc2, mask_pad, self.fc_alpha1)) * mask_pad
        if self.training and torch.is_grad_enabled() and self_att.is_cuda:
            # training step on the device: masks, concatenations, sigmoids, products, sum and scale of the merge below as pack /
            # ONE fc_alpha1 GEMM / fuse forward and two launches + the GEMMs backward (grit_amd/ops/glue.py)
            e2 = self.vis_att2(self_att, y2, y2, mask_y2)
            e1 = self.vis_att1(self_att, y1, y1, mask_y1)
            merged = glue.gated_merge_train(self_att, e1, e2, mask_pad, self.fc_alpha1)
            if merged is not None:
                return self.pwff(merged) * mask_pad
            enc1, enc2 = e1 * mask_pad, e2 * mask_pad
        else:
            enc1 = self.vis_att1(self_att, y1, y1, mask_y1) * mask_pad  # grid branch
            enc2 = self.vis_att2(self_att, y2, y2, mask_y2) * mask_pad  # region branch
        gate1 = torch.sigmoid(self.fc_alpha1(torch.cat([self_att, enc1], -1)))
        gate2 = torch.sigmoid(self.fc_alpha1(torch.cat([self_att, enc2], -1)))  # fc_alpha1 again, as in the reference
        fused = (enc1 * gate1 + enc2 * gate2) / np.sqrt(2)
        return self.pwff(fused * mask_pad) * mask_pad


class CaptionGenerator(Module):
    GENERATOR_LAYER = {'parallel': ParallelAttentionLayer}

    def __init__(self, vocab_size, max_len, n_layers, pad_idx, d_model=512, n_heads=8, d_ff=2048, dropout=.1,
                 decoder_name='parallel', cfg=None):
        super().__init__()
        if decoder_name not in self.GENERATOR_LAYER:
            raise NotImplementedError("only the 'parallel' decoder is reachable in GRIT (decoder_name is never forwarded)")
        self.d_model = d_model
        self.word_emb = nn.Embedding(vocab_size, d_model, padding_idx=pad_idx)
        self.pos_emb = nn.Embedding.from_pretrained(sinusoid_encoding_table(max_len + 1, d_model, 0), freeze=True)
        self.cfg, self.decoder_name = cfg, decoder_name
        layer_cls = self.GENERATOR_LAYER[decoder_name]
        self.layers = ModuleList([layer_cls(d_model, n_heads, d_ff, dropout) for _ in range(n_layers)])
        self.fc = nn.Linear(d_model, vocab_size, bias=False)
        self.max_len, self.pad_idx, self.N = max_len, pad_idx, n_layers
        self.register_state('running_mask_x', torch.zeros((1, 1, 0)).byte())
        self.register_state('running_seq', torch.zeros((1,)).long())

    def get_seq_inputs(self, input):
        """tokens (b, T) -> embeddings (b, T, d), self-attention mask (b, 1, T, T|t) True = masked, pad mask (b, T, 1)."""
        b_s, seq_len = input.shape[:2]
        is_pad = input == self.pad_idx
        mask_pad = (~is_pad).unsqueeze(-1).float()
        future = torch.triu(torch.ones((seq_len, seq_len), dtype=torch.bool, device=input.device), diagonal=1)
        mask_x = future[None, None] | is_pad[:, None, None, :]
        if self._is_stateful:
            self.running_mask_x = torch.cat([self.running_mask_x.bool(), mask_x], -1)
            mask_x = self.running_mask_x
        seq = torch.arange(1, seq_len + 1, device=input.device).view(1, -1).expand(b_s, -1)
        seq = seq.masked_fill(is_pad, 0)
        if self._is_stateful:
            self.running_seq.add_(1)
            seq = self.running_seq
        return self.word_emb(input) + self.pos_emb(seq), mask_x, mask_pad

    def _step_inputs(self, input):
        """get_seq_inputs + the embedding sum for ONE new token per row in stateful mode, as one launch (ops/decode_inputs.py)."""
        w, p = self.word_emb.weight, self.pos_emb.weight
        if not (_FUSED_STEP_INPUTS and self._is_stateful and input.dim() == 2 and input.shape[1] == 1
                and decode_inputs.supported(input, w, p, self.running_seq, self.running_mask_x) and w.is_contiguous() and p.is_contiguous()):
            return None
        if self.layers[0].self_att.timestep + 1 >= p.shape[0]:
            raise IndexError("decoding step %d exceeds the position table (%d rows)" % (self.layers[0].self_att.timestep + 1, p.shape[0]))
        x, self.running_mask_x, mask_pad = decode_inputs.step_inputs(input, self.pad_idx, w, p, self.running_seq, self.running_mask_x)
        return x, self.running_mask_x, mask_pad

    def forward(self, input, vis_inputs):
        if self.training and torch.is_grad_enabled() and input.is_cuda:
            _transposed.refresh_linears(self)  # W^T of every Linear: the short maps' input gradients as NT products (ops/gemm.py)
        fused = self._step_inputs(input)
        x, mask_x, mask_pad = fused if fused is not None else self.get_seq_inputs(input)
        mask_pad = mask_pad.to(x.dtype)
        y1, y2 = vis_inputs['gri_feat'], vis_inputs['reg_feat']
        m1, m2 = vis_inputs['gri_mask'], vis_inputs['reg_mask']
        for layer in self.layers:
            x = layer(x, y1, y2, mask_pad, mask_x, m1, m2)
        # vocabulary projection + log-softmax stay in float32: beam-search token identity depends on them
        with torch.autocast(x.device.type, enabled=False):
            w = self.fc.weight
            if x.is_cuda and x.dtype == w.dtype == torch.bfloat16 and not torch.is_grad_enabled() and backend.override() is None:
                # products of two bf16 values are exact in float32, so the bf16 GEMM with float32 accumulation and float32
                # output IS the float32 projection of the same operands (summation order aside) -- without up-casting the
                # [V, 512] weight on every step and at the bf16 matrix-core rate
                logits = torch.mm(x.reshape(-1, x.shape[-1]), w.t(), out_dtype=torch.float32).view(*x.shape[:-1], w.shape[0])
            else:
                logits = F.linear(x.float(), w.float())
            return F.log_softmax(logits, dim=-1)
