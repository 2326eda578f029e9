"""Sinusoid position table and the position-wise FeedForward block (reference models/common/pos_embed.py)."""
import torch
from torch import nn
from torch.nn import functional as F
from grit_amd.ops.glue import relu_dropout
from grit_amd.ops.layer_norm import add_layer_norm, linear_add_layer_norm
from grit_amd.ops.linear import Linear, linear_relu_dropout, own_or_library_linear


def position_embedding(input, d_model):
    """positions (any shape) -> [n, d_model]; channel 2i = sin(p / 10000^(2i/d)), channel 2i+1 = cos(same angle)."""
    positions = input.reshape(-1, 1)
    exponent = 2 * torch.arange(d_model // 2, dtype=torch.float32, device=positions.device).view(1, -1) / d_model
    angle = positions / 10000**exponent
    # interleave (sin, cos) pairs along the channel axis
    return torch.stack((torch.sin(angle), torch.cos(angle)), dim=-1).flatten(1)


def sinusoid_encoding_table(max_len, d_model, padding_idx=None):
    """[max_len, d_model] table for positions 0..max_len-1 (row `padding_idx` zeroed)."""
    table = position_embedding(torch.arange(max_len, dtype=torch.float32), d_model)
    if padding_idx is not None:
        table[padding_idx].zero_()
    return table


class FeedForward(nn.Module):
    """LayerNorm(x + drop(fc2(drop(relu(fc1(x))))))."""

    def __init__(self, d_model=512, d_ff=2048, dropout=0.1):
        super().__init__()
        self.fc1 = Linear(d_model, d_ff)
        self.fc2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(p=dropout)
        self.dropout_2 = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, input):
        if self.training and torch.is_grad_enabled() and input.is_cuda:
            hidden = linear_relu_dropout(input, self.fc1, self.dropout_2.p)  # ReLU + dropout in the GEMMs' epilogues (ops/linear.py)
        else:
            hidden = self.dropout_2(F.relu(self.fc1(input)))
        if self.training and torch.is_grad_enabled() and input.is_cuda:
            ln = self.layer_norm  # fc2 + dropout + residual + LayerNorm as one autograd node
            return linear_add_layer_norm(hidden, self.fc2, input, None, ln.weight, ln.bias, ln.eps, self.dropout.p, True)[1]
        ln = self.layer_norm
        branch = own_or_library_linear(hidden, self.fc2.weight, self.fc2.bias) if not torch.is_grad_enabled() else self.fc2(hidden)
        return add_layer_norm(input, self.dropout(branch), None, ln.weight, ln.bias, ln.eps)[1]
