"""A12 on the branch that decides when a caption ends: fixture G15 (tests/golden/make_golden.py::make_g15) is the imported
reference's beam-5 x 20 decode of EIGHT images from cached features with ragged grid masks, the EOS row of the vocabulary
projection scaled so that beams finish at different steps (x6: one image never, two with every beam finished; x10: EOS as
the first word).  Reference lines: models/caption/transformer.py:211-222 (`seq_mask`, the -999 fill, the index-0 survivor),
:184-188 (full descending sort), :113-126 (final re-sort).  Tokens must be bit-exact for every image whose recorded
candidate margin exceeds the fp32 noise of the decoder (all of them: the smallest recorded margin is 1.2e-4 against
scores of magnitude <= 60, i.e. 30 ulp)."""
import numpy as np
import pytest
import torch

from tests.helpers import build_model, load, oracle_ops, t

NOISE = 5e-5  # fp32 round-off of a candidate score through the 3-layer decoder (|score| <= 60: 1 ulp = 4e-6)
FEATS = ("gri_feat", "gri_mask", "reg_feat", "reg_mask")


def _model(scale, device="cpu"):
    model, cfg = build_model(3)
    with torch.no_grad():
        model.cap_generator.fc.weight[3] *= float(scale)
    return model.eval().to(device), cfg


def _decode(model, vis, out_size=5):
    model.cached_features = True
    try:
        with torch.no_grad():
            return model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=out_size)
    finally:
        model.cached_features = False


def _pinned_rows(g, scale):
    """images whose every recorded gap between consecutive candidates (top 6 per step) is above the noise"""
    top = g[f"s{scale}_top"]
    return np.abs(top[..., :-1] - top[..., 1:]).min((1, 2)) > 2 * NOISE


def _check(tokens, lps, g, scale):
    rows = _pinned_rows(g, scale)
    assert rows.all(), "fixture regenerated with a near-tie: every image is meant to be pinned"
    ref = g[f"s{scale}_tokens"]
    np.testing.assert_array_equal(tokens[rows], ref[rows])
    np.testing.assert_allclose(lps[rows], g[f"s{scale}_logprobs"][rows], rtol=1e-4, atol=1e-4)


def test_fixture_reaches_eos_on_different_steps():
    g = load("beam_g15.npz")
    for scale in (6, 10):
        tok = g[f"s{scale}_tokens"]  # [8, 5, 20]
        has = (tok == 3).any(-1)
        first = np.where(has, (tok == 3).argmax(-1), -1)
        assert has.any() and not has.all()  # finished and unfinished beams side by side
        assert has.all(-1).any()  # an image with every beam finished
        assert len(set(first[has].tolist())) >= 5  # at different steps
        for b, k in zip(*np.nonzero(has)):  # behind EOS the reference keeps selecting vocabulary index 0 at log-prob 0
            assert (tok[b, k, first[b, k] + 1:] == 0).all()
            assert (g[f"s{scale}_logprobs"][b, k, first[b, k] + 1:] == 0).all()
    assert (g["s10_tokens"][:, :, 0] == 3).any()  # EOS as the first word
    assert g["gri_mask"].any() and not g["gri_mask"].all(-1).any()  # ragged keys


@pytest.mark.parametrize("scale", [6, 10])
def test_cpu_beam_search_reaches_eos_like_the_reference(scale):
    g = load("beam_g15.npz")
    model, _ = _model(scale)
    vis = {k: t(g[k]) for k in FEATS}
    with oracle_ops():
        tokens, lps = _decode(model, vis)
    _check(tokens.numpy(), lps.numpy(), g, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [6, 10])
def test_hip_beam_search_reaches_eos_like_the_reference(scale):
    """the default device path: captured decode graph, projected K/V cache, fused beam step"""
    g = load("beam_g15.npz")
    model, _ = _model(scale, "cuda")
    vis = {k: t(g[k], device="cuda") for k in FEATS}
    tokens, lps = _decode(model, vis)
    _check(tokens.cpu().numpy(), lps.cpu().numpy(), g, scale)
    best, _ = _decode(model, vis, out_size=1)  # what inference_caption asks for
    np.testing.assert_array_equal(best.cpu().numpy(), g[f"s{scale}_tokens"][:, 0])


@pytest.mark.gpu
def test_hip_reference_order_loop_reaches_eos_like_the_reference(monkeypatch):
    """every inference restructuring off (eager loop, composed beam arithmetic, raw-history self-attention, unfused gates)"""
    import grit_amd.models.caption.cap_generator as CG
    import grit_amd.models.caption.transformer as T
    import grit_amd.models.common.attention as A
    from grit_amd.ops import gate as gate_ops
    monkeypatch.setattr(T, "_GRAPH_DECODE", False)
    monkeypatch.setattr(T, "_FUSED_BEAM_STEP", False)
    monkeypatch.setattr(A, "_KV_CACHE", False)
    monkeypatch.setattr(A, "_KV_FUSED_APPEND", False)
    monkeypatch.setattr(CG, "_FUSED_STEP_INPUTS", False)
    monkeypatch.setattr(gate_ops, "supported", lambda *a, **k: False)
    g = load("beam_g15.npz")
    model, _ = _model(6, "cuda")
    tokens, lps = _decode(model, {k: t(g[k], device="cuda") for k in FEATS})
    _check(tokens.cpu().numpy(), lps.cpu().numpy(), g, 6)


@pytest.mark.gpu
def test_hip_images_decode_alone_as_in_the_batch():
    """per-image decodes (5-row GEMMs instead of 40-row ones) give the fixture's rows too: finished beams of one image do not
    leak into another, and a batch of one whose beams have all finished keeps running to max_len"""
    g = load("beam_g15.npz")
    model, _ = _model(6, "cuda")
    for i in range(8):
        vis = {k: t(g[k][i:i + 1], device="cuda") for k in FEATS}
        tokens, _ = _decode(model, vis)
        np.testing.assert_array_equal(tokens.cpu().numpy(), g["s6_tokens"][i:i + 1])
