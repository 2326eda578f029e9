"""grit_topk_rows_f32 (beam-search candidate selection) against the reference's statement of `select`
(models/caption/transformer.py:184-188: descending torch.sort of the flattened candidates, head kept) -- bit-exact indices."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _select(x, k):
    from grit_amd import lib
    idx = torch.empty((x.shape[0], k), dtype=torch.int64, device=x.device)
    val = torch.empty((x.shape[0], k), dtype=torch.float32, device=x.device)
    st = lib.load().grit_topk_rows_f32(ctypes.c_void_p(x.data_ptr()), x.stride(0), x.shape[0], x.shape[1], k,
                                       ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(val.data_ptr()), lib.current_stream_ptr())
    lib.check(st, "grit_topk_rows_f32")
    return idx, val


@pytest.mark.parametrize("rows,n,k", [(64, 51005, 5), (3, 10201, 5), (7, 13, 5), (2, 257, 8), (5, 4099, 1), (1, 5, 5)])
def test_matches_descending_sort(rows, n, k):
    g = torch.Generator(device='cuda').manual_seed(rows * n + k)
    x = torch.randn(rows, n, device='cuda', generator=g)
    idx, val = _select(x, k)
    sv, si = torch.sort(x, -1, descending=True)
    assert torch.equal(val, sv[:, :k])
    assert torch.equal(idx, si[:, :k])  # continuous random values: no ties


def test_finished_beams_ties_and_nan():
    """The candidate rows of beam search: a finished beam is -999 everywhere except index 0 (exact ties), scores of other beams
    are ordinary; ties resolve by ascending index; NaN ranks first (torch's order)."""
    V = 1000
    x = torch.full((2, 5 * V), -999.0, device='cuda')
    x[0, 0 * V] = -3.5           # finished beam 0 keeps its score at vocabulary index 0
    x[0, 1 * V:2 * V] = torch.linspace(-20, -4, V, device='cuda')  # a live beam
    x[0, 3 * V] = -3.5           # another finished beam with the same score: tie -> lower index first
    idx, val = _select(x, 5)
    assert idx[0].tolist() == [0, 3 * V, 2 * V - 1, 2 * V - 2, 2 * V - 3]
    assert val[0, :2].tolist() == [-3.5, -3.5]
    # a row of all -999: the first five positions
    assert idx[1].tolist() == [0, 1, 2, 3, 4]
    y = torch.randn(1, 300, device='cuda')
    y[0, 77] = float('nan')
    assert _select(y, 3)[0][0, 0].item() == 77


def test_strided_and_unaligned_rows():
    base = torch.randn(4, 1003, device='cuda')
    x = base[:, 3:]  # row stride 1003, first element 12 bytes past an aligned address
    idx, val = _select(x, 5)
    sv, si = torch.sort(x, -1, descending=True)
    assert torch.equal(idx, si[:, :5]) and torch.equal(val, sv[:, :5])


# ------------------------------------------------------------------ grit_beam_step_f32: one whole step of Transformer.iter
def _composed_step(word_logprob, seq_logprob, seq_mask, prev_words, eos, beam, first):
    """The reference's arithmetic, op by op (models/caption/transformer.py:208-240)."""
    B, cur, V = word_logprob.shape
    candidates = seq_logprob + word_logprob
    if not first:
        alive = (prev_words.view(B, cur) != eos).float().unsqueeze(-1)
        seq_mask = seq_mask * alive
        word_logprob = word_logprob * seq_mask
        frozen = seq_logprob.expand_as(candidates).contiguous()
        frozen[:, :, 1:] = -999
        candidates = seq_mask * candidates + frozen * (1 - seq_mask)
    sv, si = torch.sort(candidates.view(B, -1), dim=-1, descending=True, stable=True)
    sel_lp, sel_idx = sv[:, :beam], si[:, :beam]
    sel_beam = torch.div(sel_idx, V, rounding_mode='floor')
    sel_word = sel_idx - sel_beam * V
    col = sel_beam.unsqueeze(-1)
    new_mask = torch.gather(seq_mask, 1, col) if not first else torch.ones(B, beam, 1, device=word_logprob.device)
    picked = torch.gather(torch.gather(word_logprob, 1, col.expand(B, beam, V)), 2, sel_word.unsqueeze(-1))
    return sel_beam, sel_word, sel_lp.unsqueeze(-1), new_mask, picked


@pytest.mark.parametrize("B,cur,V,beam", [(64, 5, 10201, 5), (64, 1, 10201, 5), (3, 5, 10201, 5), (2, 3, 517, 3), (5, 8, 1031, 8),
                                          (1, 1, 9, 5), (4, 2, 6, 4)])
def test_beam_step_matches_composed_arithmetic(B, cur, V, beam):
    from grit_amd.ops.beam import beam_step
    g = torch.Generator(device='cuda').manual_seed(B * 131 + cur * 17 + V)
    first = cur == 1
    lp = torch.log_softmax(torch.randn(B, cur, V, device='cuda', generator=g) * 3, -1)
    seq_lp = -torch.rand(B, cur, 1, device='cuda', generator=g) * 10
    seq_mask = (torch.rand(B, cur, 1, device='cuda', generator=g) > 0.3).float()
    eos = 3
    prev = torch.randint(0, 6, (B * cur, 1), device='cuda', generator=g)  # a third of the beams just emitted <eos>
    if first:
        seq_lp = torch.zeros(B, 1, 1, device='cuda')
    want = _composed_step(lp, seq_lp, seq_mask, prev, eos, beam, first)
    got = beam_step(lp, seq_lp, None if first else seq_mask, None if first else prev, eos, beam)
    names = ("sel_beam", "sel_word", "seq_logprob", "seq_mask", "picked")
    for n, a, b in zip(names, got, want):
        assert a.shape == b.shape, n
        # bit-exact, signed zeros included (picked of a finished beam is logp * 0 = -0.0)
        assert torch.equal(a, b), n
        if a.dtype == torch.float32:
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), n


def test_beam_step_all_beams_finished_and_ties():
    """Every beam finished: the candidates are the running scores at word 0 and -999 elsewhere; equal scores keep beam order."""
    from grit_amd.ops.beam import beam_step
    B, cur, V, beam = 2, 5, 300, 5
    lp = torch.log_softmax(torch.randn(B, cur, V, device='cuda'), -1)
    seq_lp = torch.tensor([[-1.0, -2.0, -2.0, -0.5, -7.0], [-3.0] * 5], device='cuda').unsqueeze(-1)
    seq_mask = torch.zeros(B, cur, 1, device='cuda')
    prev = torch.zeros(B * cur, 1, dtype=torch.int64, device='cuda')
    sel_beam, sel_word, new_lp, new_mask, picked = beam_step(lp, seq_lp, seq_mask, prev, 3, beam)
    assert sel_beam[0].tolist() == [3, 0, 1, 2, 4] and sel_word[0].tolist() == [0] * 5
    assert sel_beam[1].tolist() == [0, 1, 2, 3, 4]
    assert new_lp[0, :, 0].tolist() == [-0.5, -1.0, -2.0, -2.0, -7.0]
    assert new_mask.abs().sum().item() == 0 and picked.abs().sum().item() == 0
    want = _composed_step(lp, seq_lp, seq_mask, prev, 3, beam, False)
    for a, b in zip((sel_beam, sel_word, new_lp, new_mask, picked), want):
        assert torch.equal(a, b)
