"""Inputs of one step of step-wise decoding as one launch (grit_decode_step_inputs, include/grit_hip.h): the pad / key masks, the step
counter and the word + position embedding sum of CaptionGenerator.get_seq_inputs in stateful mode (reference
models/caption/cap_generator.py:116-137,148)."""
import ctypes

import torch

from grit_amd import lib as _lib
from grit_amd.ops import backend


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def supported(tokens, word_emb, pos_emb, running_seq, running_mask):
    return (backend.override() is None and tokens.is_cuda and tokens.dim() == 2 and tokens.shape[1] == 1 and tokens.dtype == torch.int64
            and not torch.is_grad_enabled() and word_emb.dtype == pos_emb.dtype and word_emb.dtype in (torch.bfloat16, torch.float32)
            and running_seq.dtype == torch.int64 and running_seq.numel() == tokens.shape[0]
            and running_mask.dim() == 4 and running_mask.shape[0] == tokens.shape[0] and running_mask.shape[1:3] == (1, 1)
            and running_mask.dtype in (torch.bool, torch.uint8) and not torch.is_autocast_enabled())


def step_inputs(tokens, pad_idx, word_emb, pos_emb, running_seq, running_mask):
    """tokens [R, 1] -> x [R, 1, d], key mask [R, 1, 1, t + 1] bool (True = masked), mask_pad [R, 1, 1]; running_seq advanced in place."""
    R = tokens.shape[0]
    d = word_emb.shape[1]
    t_old = running_mask.shape[-1]
    tok = tokens.reshape(-1).contiguous()
    seq = running_seq if running_seq.is_contiguous() else None
    if seq is None:
        raise _lib.GritHipError("running_seq must be contiguous (it is advanced in place)")
    old = running_mask.contiguous().view(torch.uint8) if t_old else None
    x = torch.empty((R, 1, d), dtype=word_emb.dtype, device=tokens.device)
    mask_pad = torch.empty((R, 1, 1), dtype=word_emb.dtype, device=tokens.device)
    new_mask = torch.empty((R, 1, 1, t_old + 1), dtype=torch.uint8, device=tokens.device)
    with _lib.device_guard(tokens.device):
        st = _lib.load().grit_decode_step_inputs(_ptr(tok), int(pad_idx), _ptr(word_emb), word_emb.shape[0], _ptr(pos_emb),
                                                 pos_emb.shape[0], d, int(word_emb.dtype == torch.bfloat16), _ptr(seq), _ptr(old), t_old,
                                                 R, _ptr(x), _ptr(mask_pad), _ptr(new_mask), _lib.current_stream_ptr())
    _lib.check(st, "grit_decode_step_inputs")
    return x, new_mask.view(torch.bool), mask_pad
