"""Shared test plumbing: build this repo's model with the name-keyed deterministic fill, run it on CPU with the
oracle ops injected (the ONLY way the modules run without a HIP device)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLDEN)

from fill import deterministic_fill_  # noqa: E402

from grit_amd.config import default_config  # noqa: E402
from grit_amd.ops.backend import use_reference_ops  # noqa: E402


def oracle_ops():
    from oracle import torch_ref
    return use_reference_ops(torch_ref)


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def build_model(n_layers=3, fill=True, **over):
    from grit_amd.models.caption import Transformer
    from grit_amd.models.caption.detector import build_detector
    cfg = default_config(**{'model.cap_generator.n_layers': n_layers, **over})
    model = Transformer(build_detector(cfg), cfg)
    if fill:
        deterministic_fill_(model)
    return model, cfg


def disable_drop_path(model):
    from grit_amd.models.common.swin_model import DropPath
    for m in model.modules():
        if isinstance(m, DropPath):
            m.drop_prob = 0.0


def t(a, dtype=None, device="cpu"):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return (x if dtype is None else x.to(dtype)).to(device)
