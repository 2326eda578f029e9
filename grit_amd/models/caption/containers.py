"""Stateful modules for step-wise decoding (reference models/caption/containers.py:13-92).

A *state* is a registered buffer with a default value.  Beam search switches a module tree into stateful mode
(`statefulness(batch)`: every state becomes `default` repeated over the batch), mutates and re-gathers the states every
step (`apply_to_states(fn)`: state <- fn(state), depth first) and leaves the mode again (states back to their defaults,
step counters to zero).  The states are ordinary buffers, hence part of the state dict -- which is why data parallel
training never broadcasts buffers (train_caption.py:61, broadcast_buffers=False).

Kept from the reference: the public surface (`register_state`, `states`, `apply_to_states`, `enable_statefulness`,
`disable_statefulness`, `statefulness`, the `_is_stateful` / `timestep` attributes, `ModuleList`, `ModuleDict`) and the
visiting order (own states first, then children in registration order).  Different here: the defaults are converted to
the buffer's device / dtype once and cached there, so entering and leaving beam search issues no host-to-device copy
(the reference re-uploads every default twice per beam search: two synchronising pageable transfers per state)."""
from contextlib import contextmanager

from torch import nn

from grit_amd.utils.typing import TensorOrNone


class _State(object):
    """Default value of one registered state, with per-(device, dtype) resident copies."""
    __slots__ = ("name", "default", "resident")

    def __init__(self, name, default):
        self.name = name
        self.default = None if default is None else default.detach().clone()
        self.resident = {}

    def fresh(self, like):
        """A new tensor holding the default, on `like`'s device (and dtype, for floating states); None stays None."""
        if self.default is None:
            return None
        device = self.default.device if like is None else like.device
        follow = like is not None and like.is_floating_point() and self.default.is_floating_point()
        dtype = like.dtype if follow else self.default.dtype
        key = (str(device), dtype)
        if key not in self.resident:
            self.resident[key] = self.default.to(device=device, dtype=dtype)
        return self.resident[key].clone()


class Module(nn.Module):

    def __init__(self):
        super().__init__()
        self._is_stateful = False
        self._states = []
        self.timestep = 0

    # ------------------------------------------------------------------ registration / traversal
    def register_state(self, name: str, default: TensorOrNone):
        self._states.append(_State(name, default))
        self.register_buffer(name, default)

    def _walk(self):
        """This module, then every stateful descendant reachable through stateful children (registration order)."""
        yield self
        for child in self.children():
            if isinstance(child, Module):
                yield from child._walk()

    def states(self):
        for module in self._walk():
            for st in module._states:
                yield module._buffers[st.name]

    def apply_to_states(self, fn):
        for module in self._walk():
            for st in module._states:
                module._buffers[st.name] = fn(module._buffers[st.name])

    # ------------------------------------------------------------------ mode switches
    def _load_defaults(self, batch_size=None):
        for st in self._states:
            value = st.fresh(self._buffers[st.name])
            if value is not None and batch_size is not None:
                value = value.unsqueeze(0).expand(batch_size, *value.shape).contiguous()
            self._buffers[st.name] = value

    def enable_statefulness(self, batch_size: int):
        for module in self._walk():
            module._load_defaults(batch_size)
            module._is_stateful = True

    def disable_statefulness(self):
        for module in self._walk():
            module.timestep = 0
            module._load_defaults()
            module._is_stateful = False

    @contextmanager
    def statefulness(self, batch_size: int):
        self.enable_statefulness(batch_size)
        try:
            yield
        finally:
            self.disable_statefulness()


class ModuleList(nn.ModuleList, Module):
    pass


class ModuleDict(nn.ModuleDict, Module):
    pass
