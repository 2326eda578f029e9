"""BaseCaptioner (reference models/caption/base.py): the stateful-module root the Transformer derives from."""
import torch

from grit_amd.models.caption.containers import Module


class BaseCaptioner(Module):

    def __init__(self):
        super().__init__()

    def init_weights(self):
        raise NotImplementedError

    def step(self, t, prev_output, visual, seq, mode='teacher_forcing', **kwargs):
        raise NotImplementedError

    def forward(self, images, seq, *args):
        """Generic step-wise teacher forcing (unused by GRIT's Transformer, which overrides forward)."""
        state = self.init_state(images.size(0), images.device)
        out, outputs = None, []
        for t in range(seq.size(1)):
            out, state = self.step(t, state, out, images, seq, *args, mode='teacher_forcing')
            outputs.append(out)
        return torch.stack(outputs, 1)
