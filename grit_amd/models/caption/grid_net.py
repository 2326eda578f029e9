"""Grid feature network of the captioner (reference models/caption/grid_net.py:9-42).

    tokens [B, N, d_in]  --fc + ReLU + dropout + LayerNorm-->  [B, N, d_model]
                         --n_layers x (self-attention MHA -> position-wise FFN)-->  one [B, N, d_model] per layer

N = (H/64)*(W/64) grid tokens of the coarsest backbone map (100 at 640x640).  Parameter names follow the reference
(`fc`, `layer_norm`, `layers.<i>.mhatt`, `layers.<i>.pwff`) so its checkpoints load unchanged.  The captioner only consumes
the last layer (transformer.py:69), so the per-layer outputs are written into one preallocated [B, n_layers, N, d_model]
tensor instead of being collected and concatenated; in training the MHA / FFN tails run as fused
projection + dropout + residual + LayerNorm nodes (grit_amd/ops/layer_norm.py)."""
import torch
import torch.nn.functional as F
from torch import nn

from grit_amd.models.common.attention import MultiHeadAttention
from grit_amd.models.common.pos_embed import FeedForward
from grit_amd.ops import transposed as _transposed
from grit_amd.ops.linear import Linear, mark_single_use


class TransformerLayer(nn.Module):
    """Post-norm encoder layer: LayerNorm(x + MHA(x)) followed by LayerNorm(y + FFN(y)); both norms live in the sub-modules."""

    def __init__(self, d_model=512, n_heads=8, d_ff=2048, dropout=.1, n_memories=0):
        super().__init__()
        self.mhatt = MultiHeadAttention(d_model, n_heads, dropout, n_memories=n_memories)
        self.pwff = FeedForward(d_model, d_ff, dropout)

    def forward(self, q, k, v, mask=None):
        attended = self.mhatt(q, k, v, mask)
        return self.pwff(attended)


class GridFeatureNetwork(nn.Module):

    def __init__(self, n_layers, d_in=1024, d_model=512, n_heads=8, d_ff=2048, dropout=0.1, n_memories=0):
        super().__init__()
        self.d_model = d_model
        self.fc = Linear(d_in, d_model)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model)
        stack = (TransformerLayer(d_model, n_heads, d_ff, dropout, n_memories=n_memories) for _ in range(n_layers))
        self.layers = nn.ModuleList(stack)
        mark_single_use(self)  # every Linear here runs once per forward pass: weight gradients beside the backward chain

    def embed(self, tokens):
        """The input stage: project to d_model, ReLU, dropout, LayerNorm."""
        return self.layer_norm(self.dropout(F.relu(self.fc(tokens))))

    def last(self, input, mask=None):
        """Output of the LAST layer only, [B, N, d_model] -- all the captioner consumes (reference transformer.py:69 takes
        out[:, -1]): the training step skips the [B, n_layers, N, d_model] collection and the slice / copy gradients behind it."""
        if self.training and torch.is_grad_enabled() and input.is_cuda:
            _transposed.refresh_linears(self)  # W^T of every Linear: the short maps' input gradients as NT products (ops/gemm.py)
        x = self.embed(input)
        for layer in self.layers:
            x = layer(x, x, x, mask)
        return x

    def forward(self, input, mask=None):
        """-> (outs [B, n_layers, N, d_model], mask), outs[:, i] being the output of layer i."""
        if self.training and torch.is_grad_enabled() and input.is_cuda:
            _transposed.refresh_linears(self)
        x = self.embed(input)
        if len(self.layers) == 0:
            return x.new_empty(x.shape[0], 0, x.shape[1], self.d_model), mask
        outs = None
        for i, layer in enumerate(self.layers):
            x = layer(x, x, x, mask)
            if outs is None:
                outs = x.new_empty(x.shape[0], len(self.layers), x.shape[1], x.shape[2])
            outs[:, i] = x
        return outs, mask
