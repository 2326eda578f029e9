"""grit_amd: MI355X-native hot path of the GRIT image-captioning model (reference: davidnvq/grit).

Layout
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/grit_hip.h) -> libgrit_hip.so
  lib.py     ctypes binding of that ABI (fails loudly when the library is absent)
  ops/       autograd wrappers around the kernels (MSDA, window attention, decoder attention)
  models/, engine/, utils/   host-side mirror of the reference's Python interface (same class names,
             argument meaning and state-dict keys), importable also as top-level `models`, `engine`, `utils`.
"""
__version__ = "0.1.0"
