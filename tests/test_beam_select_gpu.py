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
