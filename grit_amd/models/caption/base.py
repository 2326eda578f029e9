"""BaseCaptioner (reference models/caption/base.py): the stateful-module root the Transformer derives from.

The reference also carries a generic step-by-step teacher-forcing `forward` here; GRIT's Transformer overrides `forward`
(transformer.py:53-74) and never reaches it, so only the abstract hooks remain."""
from grit_amd.models.caption.containers import Module


class BaseCaptioner(Module):

    def init_weights(self):
        raise NotImplementedError

    def step(self, t, prev_output, visual, seq, mode='teacher_forcing', **kwargs):
        raise NotImplementedError
