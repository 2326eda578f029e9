"""GridFeatureNetwork: Linear(d_in->512)+ReLU+dropout+LayerNorm, then n_layers x (MHA + FeedForward) of
self-attention over the H/64 x W/64 grid tokens (reference models/caption/grid_net.py:9-42)."""
import torch
from torch import nn
from torch.nn import functional as F

from grit_amd.models.common.attention import MultiHeadAttention
from grit_amd.models.common.pos_embed import FeedForward


class TransformerLayer(nn.Module):

    def __init__(self, d_model=512, n_heads=8, d_ff=2048, dropout=.1, n_memories=0):
        super().__init__()
        self.mhatt = MultiHeadAttention(d_model, n_heads, dropout, n_memories=n_memories)
        self.pwff = FeedForward(d_model, d_ff, dropout)

    def forward(self, q, k, v, mask=None):
        return self.pwff(self.mhatt(q, k, v, mask))


class GridFeatureNetwork(nn.Module):

    def __init__(self, n_layers, d_in=1024, d_model=512, n_heads=8, d_ff=2048, dropout=0.1, n_memories=0):
        super().__init__()
        self.fc = nn.Linear(d_in, d_model)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model)
        self.layers = nn.ModuleList(
            [TransformerLayer(d_model, n_heads, d_ff, dropout, n_memories=n_memories) for _ in range(n_layers)])

    def forward(self, input, mask=None):
        """-> (outs [B, n_layers, N, d_model], mask); the captioner keeps outs[:, -1]."""
        out = self.layer_norm(self.dropout(F.relu(self.fc(input))))
        per_layer = []
        for layer in self.layers:
            out = layer(out, out, out, mask)
            per_layer.append(out)
        return torch.stack(per_layer, 1), mask
