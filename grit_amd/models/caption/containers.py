"""nn.Module with registered *states*: buffers that beam search expands to the batch, re-gathers per step and
resets afterwards (reference models/caption/containers.py:13-92; the beam caches are part of the state dict,
which is why DDP is built with broadcast_buffers=False)."""
from contextlib import contextmanager

from torch import nn

from grit_amd.utils.typing import TensorOrNone


class Module(nn.Module):

    def __init__(self):
        super().__init__()
        self._is_stateful = False
        self._state_names = []
        self._state_defaults = dict()
        self.timestep = 0

    def register_state(self, name: str, default: TensorOrNone):
        self._state_names.append(name)
        self._state_defaults[name] = None if default is None else default.clone().detach()
        self.register_buffer(name, default)

    def _stateful_children(self):
        return (m for m in self.children() if isinstance(m, Module))

    def states(self):
        for name in self._state_names:
            yield self._buffers[name]
        for m in self._stateful_children():
            yield from m.states()

    def apply_to_states(self, fn):
        for name in self._state_names:
            self._buffers[name] = fn(self._buffers[name])
        for m in self._stateful_children():
            m.apply_to_states(fn)

    def _fresh(self, name):
        default = self._state_defaults[name]
        if default is None:
            return None
        buf = self._buffers[name]
        # follow the module's device and (for floating states) dtype: .to(bf16) / .cuda() convert the registered buffer
        # but not this private default.  The converted default is kept per (device, dtype): a host -> device copy on
        # every beam search is a synchronising pageable transfer (and illegal while a hipGraph is being captured)
        device = buf.device if buf is not None else default.device
        dtype = buf.dtype if (buf is not None and default.is_floating_point() and buf.is_floating_point()) else default.dtype
        cache = self.__dict__.setdefault('_state_defaults_on', {})
        key = (name, str(device), dtype)
        if key not in cache:
            cache[key] = default.clone().detach().to(device=device, dtype=dtype)
        return cache[key].clone()

    def _init_states(self, batch_size: int):
        for name in self._state_names:
            t = self._fresh(name)
            if t is not None:
                t = t.unsqueeze(0).expand([batch_size] + list(t.shape)).contiguous()
            self._buffers[name] = t

    def _reset_states(self):
        for name in self._state_names:
            self._buffers[name] = self._fresh(name)

    def enable_statefulness(self, batch_size: int):
        for m in self._stateful_children():
            m.enable_statefulness(batch_size)
        self._init_states(batch_size)
        self._is_stateful = True

    def disable_statefulness(self):
        self.timestep = 0
        for m in self._stateful_children():
            m.disable_statefulness()
        self._reset_states()
        self._is_stateful = False

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
