"""Row-level parity of SURVEY rows A4 / A5 and the self-critical step from cached features (N2):

  G3   MSDeformAttn module (models/ops/modules/ms_deform_attn.py:73-119): 2-d / 4-d reference points, padding mask
  G12  DetectionModule (models/detection/det_module.py:135-213): six decoder layers with box refinement from stored level maps
       with ragged padding masks -> hs and reference boxes
  G13  one self-critical step from the reference detector's cached features (engine/caption_engine.py:421-443)

CPU: host logic with the oracle ops injected.  GPU (through the C ABI): fp32 kernels at 1e-4 as north_star states, and the
stacked-value-map bf16 path the benchmark times at a stated bf16 tolerance."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, build_model, deterministic_fill_, disable_drop_path, load, oracle_ops, t


def _g3_module(device="cpu", dtype=torch.float32):
    from grit_amd.models.ops.modules import MSDeformAttn
    mod = deterministic_fill_(MSDeformAttn(d_model=128, n_levels=3, n_heads=4, n_points=4), "g3.")
    return mod.to(device=device, dtype=dtype)


def _det_module(device="cpu"):
    from grit_amd.config import default_config
    from grit_amd.models.detection.det_module import build_det_module_with_config
    cfg = default_config(**{'model.detector.dropout': 0.0})
    return deterministic_fill_(build_det_module_with_config(cfg.model.detector), "g12.").to(device)


def _g12_inputs(g, device="cpu"):
    return [t(g["src%d" % l], device=device) for l in range(4)], [t(g["mask%d" % l], device=device) for l in range(4)]


def test_detection_module_rows_cpu_oracle():
    g = load("det_g12.npz")
    mod = _det_module().eval()
    srcs, masks = _g12_inputs(g)
    with oracle_ops(), torch.no_grad():
        hs, init_ref, refs = mod(srcs, masks)
    np.testing.assert_allclose(init_ref.numpy(), g["init_ref"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(refs.numpy(), g["inter_refs"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(hs[1].numpy(), g["hs_first"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(hs[-1].numpy(), g["hs_last"], rtol=1e-4, atol=1e-4)


def _sc_from_cached_features(model, cfg, g, device):
    from grit_amd.engine.caption_engine import build_optimizers, train_sc_step
    B, beam, T = g["tokens"].shape
    cfg.model.beam_size, cfg.model.beam_len = beam, T
    opts = build_optimizers(model, cfg, mode='sc')
    seen = {}

    def reward_fn(tokens, batch):
        seen['tokens'] = tokens.cpu()
        return t(g["reward"], device=device)

    feats = {k[5:]: t(g[k], device=device) for k in g.files if k.startswith("feat:")}
    model.cached_features = True
    try:
        loss, reward, baseline = train_sc_step(model, {'samples': feats}, opts, reward_fn, cfg)
    finally:
        model.cached_features = False
    return loss, seen['tokens']


def _check_sc(model, loss, tokens, g, ref, rtol_loss, rtol_norm):
    np.testing.assert_array_equal(tokens.numpy(), g["tokens"])  # same beams, unconditionally
    assert abs(loss.item() - ref["loss"]) < rtol_loss * abs(ref["loss"]) + 1e-7, (loss.item(), ref["loss"])
    params = dict(model.named_parameters())
    norms = {}
    for n, p in params.items():
        if p.requires_grad and p.grad is not None and not n.startswith("detector"):
            top = n.split('.')[0]
            norms[top] = norms.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    for k, v in ref["grad_norms"].items():
        assert abs(norms[k]**0.5 - v) < rtol_norm * v, (k, norms[k]**0.5, v)
    for k in g.files:
        if k.startswith("grad:"):
            got = params[k[5:]].grad.flatten()[:64].float().cpu().numpy()
            assert np.abs(got - g[k]).max() <= rtol_norm * np.abs(g[k]).max() + 1e-7, k
    # the detector is not part of the graph in cached-feature mode
    assert all(p.grad is None for n, p in params.items() if n.startswith("detector"))


def test_self_critical_step_from_cached_features_cpu_oracle():
    g = load("sc_g13.npz")
    ref = json.load(open(os.path.join(GOLDEN, "sc_g13.json")))
    model, cfg = build_model(3, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train()
    with oracle_ops():
        loss, tokens = _sc_from_cached_features(model, cfg, g, "cpu")
    _check_sc(model, loss, tokens, g, ref, 1e-4, 1e-3)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_msdeformattn_module_fp32_on_hip_g3():
    """A4 on the device, fp32, through grit_msda_fwd_f32: 2-d refs without mask, 4-d refs with the padding mask -- 1e-4."""
    g = load("msda_module_g3.npz")
    mod = _g3_module("cuda")
    sh, lsi = t(g["shapes"], device="cuda"), t(g["lsi"], device="cuda")
    with torch.no_grad():
        out2 = mod(t(g["query"], device="cuda"), t(g["ref2"], device="cuda"), t(g["src"], device="cuda"), sh, lsi, None)
        out4 = mod(t(g["query"], device="cuda"), t(g["ref4"], device="cuda"), t(g["src"], device="cuda"), sh, lsi,
                   t(g["pad"], device="cuda"))
    np.testing.assert_allclose(out2.cpu().numpy(), g["out2"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(out4.cpu().numpy(), g["out4"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_msdeformattn_module_fp64_on_hip_g3():
    g = load("msda_module_g3.npz")
    mod = _g3_module("cuda", torch.float64)
    sh, lsi = t(g["shapes"], device="cuda"), t(g["lsi"], device="cuda")
    with torch.no_grad():
        out4 = mod(t(g["query"], device="cuda").double(), t(g["ref4"], device="cuda").double(), t(g["src"], device="cuda").double(),
                   sh, lsi, t(g["pad"], device="cuda"))
    np.testing.assert_allclose(out4.cpu().numpy(), g["out4"], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_detection_module_rows_fp32_on_hip_g12():
    """A5 on the device in fp32 (fp32 MSDA + fp32 attention kernels): hs and boxes of the reference within 1e-4."""
    g = load("det_g12.npz")
    mod = _det_module("cuda").eval()
    srcs, masks = _g12_inputs(g, "cuda")
    with torch.no_grad():
        hs, init_ref, refs = mod(srcs, masks)
    np.testing.assert_allclose(init_ref.cpu().numpy(), g["init_ref"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(refs.cpu().numpy(), g["inter_refs"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(hs[1].cpu().numpy(), g["hs_first"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(hs[-1].cpu().numpy(), g["hs_last"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_detection_module_stacked_bf16_path_g12():
    """The path the benchmark times: train mode, bf16 weights, ONE value-projection GEMM into the stacked [B, S, 6, M, D]
    map sampled in place (DetectionModule.project_values), ragged masks.  bf16 storage through six layers: 3e-2 of the
    data scale at the maximum, 5e-3 on average; gradients flow to the stacked projection weights."""
    from grit_amd.models.detection import det_module as dm
    g = load("det_g12.npz")
    mod = _det_module("cuda").bfloat16().train()
    srcs, masks = _g12_inputs(g, "cuda")
    srcs = [s.bfloat16().requires_grad_(True) for s in srcs]
    used = {}
    orig = dm.DetectionModule.project_values

    def spy(self, src, padding_mask):
        out = orig(self, src, padding_mask)
        used['stacked'] = isinstance(out[0], tuple)
        return out

    dm.DetectionModule.project_values = spy
    try:
        hs, init_ref, refs = mod(srcs, masks)
    finally:
        dm.DetectionModule.project_values = orig
    assert used.get('stacked') is True, "the stacked value-map path did not run"
    for got, key in ((hs[1], "hs_first"), (hs[-1], "hs_last")):
        ref = g[key]
        err = np.abs(got.float().detach().cpu().numpy() - ref)
        assert err.max() < 3e-2 * np.abs(ref).max() and err.mean() < 5e-3 * np.abs(ref).max(), (key, err.max(), err.mean())
    np.testing.assert_allclose(refs.float().cpu().numpy(), g["inter_refs"], atol=2e-2)
    hs[-1].float().pow(2).sum().backward()
    w = mod.decoder_layers[0].cross_attn.value_proj.weight
    assert w.grad is not None and torch.isfinite(w.grad).all() and w.grad.abs().max() > 0
    assert all(s.grad is not None and torch.isfinite(s.grad).all() for s in srcs)


@pytest.mark.gpu
def test_self_critical_step_from_cached_features_on_hip_g13():
    """N2 on the device, fp32: beam search with gradient through the fp32 attention kernels on the reference's own features.
    Same beams as the reference (asserted), so loss and gradients are compared unconditionally."""
    g = load("sc_g13.npz")
    ref = json.load(open(os.path.join(GOLDEN, "sc_g13.json")))
    model, cfg = build_model(3, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train().to("cuda")
    disable_drop_path(model)
    loss, tokens = _sc_from_cached_features(model, cfg, g, "cuda")
    _check_sc(model, loss, tokens, g, ref, 1e-3, 5e-3)
