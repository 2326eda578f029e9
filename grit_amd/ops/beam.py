"""One beam-search step after the decoder as two launches (grit_beam_step_f32, include/grit_hip.h): the finished-beam masking,
candidate scores, top-k over beam x vocabulary, beam / word split and the score / mask / log-prob gathers of the reference's
Transformer.iter (models/caption/transformer.py:208-240), bit-identical to the composed torch form."""
import ctypes

import torch

from grit_amd import lib as _lib
from grit_amd.ops import backend


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def supported(word_logprob, cur_beam, beam_size):
    return (word_logprob.is_cuda and word_logprob.dtype == torch.float32 and not word_logprob.requires_grad
            and not torch.is_grad_enabled() and backend.override() is None and beam_size <= 8
            and cur_beam * beam_size * (8 if cur_beam == 1 else 2) <= 128 and word_logprob.shape[0] <= 65535)


def beam_step(word_logprob, seq_logprob, seq_mask, prev_words, eos_idx, beam_size):
    """word_logprob [B, cur_beam, V] f32, seq_logprob [B, cur_beam, 1], seq_mask [B, cur_beam, 1] and prev_words [B * cur_beam, 1]
    (both None at the first step) -> sel_beam, sel_word [B, beam] int64, seq_logprob, seq_mask, picked log-prob [B, beam, 1] f32."""
    B, cur, V = word_logprob.shape
    lp = word_logprob if word_logprob.is_contiguous() else word_logprob.contiguous()
    first = prev_words is None
    dev = lp.device
    L = _lib.load()
    nbytes = L.grit_beam_step_workspace(B, cur, beam_size)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    sel_beam = torch.empty((B, beam_size), dtype=torch.int64, device=dev)
    sel_word = torch.empty((B, beam_size), dtype=torch.int64, device=dev)
    new_lp = torch.empty((B, beam_size, 1), dtype=torch.float32, device=dev)
    new_mask = torch.empty((B, beam_size, 1), dtype=torch.float32, device=dev)
    picked = torch.empty((B, beam_size, 1), dtype=torch.float32, device=dev)
    slp = seq_logprob.reshape(-1).float().contiguous()
    if slp.numel() != B * cur:
        raise _lib.GritHipError("seq_logprob has %d entries for %d x %d beams" % (slp.numel(), B, cur))
    sm = pw = None
    if not first:
        sm = seq_mask.reshape(-1).float().contiguous()
        pw = prev_words.reshape(-1).contiguous()
        if sm.numel() != B * cur or pw.numel() != B * cur or pw.dtype != torch.int64:
            raise _lib.GritHipError("beam state does not match %d x %d beams" % (B, cur))
    with _lib.device_guard(dev):
        st = L.grit_beam_step_f32(_ptr(lp), lp.stride(1), _ptr(slp), _ptr(sm), _ptr(pw), int(eos_idx), int(first), B, cur, V,
                                  beam_size, _ptr(ws), nbytes, _ptr(sel_beam), _ptr(sel_word), _ptr(new_lp), _ptr(new_mask),
                                  _ptr(picked), _lib.current_stream_ptr())
    _lib.check(st, "grit_beam_step_f32")
    return sel_beam, sel_word, new_lp, new_mask, picked
