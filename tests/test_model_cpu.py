"""Whole-model host logic on CPU (oracle ops injected) against the fixtures made by the imported reference:
G7 = BASELINE config 1 (224x224, 2-layer decoder, greedy + beam-5 decode, teacher forcing), G8 = one XE step."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, build_model, disable_drop_path, load, oracle_ops, t


@pytest.fixture(scope="module")
def g7_model():
    model, cfg = build_model(2)
    return model.eval(), cfg


def test_config1_detector_features(g7_model):
    from grit_amd.utils.misc import NestedTensor
    model, _ = g7_model
    g = load("model_g7.npz")
    with oracle_ops(), torch.no_grad():
        vis = model.detector(NestedTensor(t(g["image"]), torch.zeros(1, 224, 224, dtype=torch.bool)))
    assert vis["gri_feat"].shape == (1, 16, 1024) and vis["reg_feat"].shape == (1, 150, 512)  # S=1045 / N_grid=16 probe
    np.testing.assert_allclose(vis["gri_feat"].numpy(), g["gri_feat"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(vis["reg_feat"].numpy(), g["reg_feat"], rtol=1e-3, atol=2e-4)
    assert not vis["reg_mask"].any() and vis["reg_mask"].shape == (1, 1, 1, 150)
    np.testing.assert_array_equal(vis["gri_mask"].numpy(), g["gri_mask"])


def test_config1_teacher_forcing_logprobs(g7_model):
    from grit_amd.utils.misc import NestedTensor
    model, _ = g7_model
    g = load("model_g7.npz")
    with oracle_ops(), torch.no_grad():
        lp = model(NestedTensor(t(g["image"]), torch.zeros(1, 224, 224, dtype=torch.bool)), t(g["seq"]))
    assert lp.shape == (1, 20, 10201)
    top = lp.topk(16, -1)
    np.testing.assert_array_equal(top.indices[..., :4].numpy(), g["tf_top_idx"][..., :4])
    np.testing.assert_allclose(top.values.numpy(), g["tf_top_val"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lp.mean(-1).numpy(), g["tf_row_mean"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(torch.logsumexp(lp, -1).numpy(), 0.0, atol=1e-4)


@pytest.mark.parametrize("beam", [1, 5])
def test_config1_beam_tokens_bit_exact(g7_model, beam):
    """inference_caption plumbing: beam 1 (= greedy, the reference has no separate greedy path) and beam 5.
    Token ids must be identical; the recorded candidate margins (>= 7e-4) dwarf fp32 noise (~1e-5)."""
    from inference_caption import caption_tokens
    model, cfg = g7_model
    g = load("model_g7.npz")
    with oracle_ops():
        tokens, lps = caption_tokens(model, t(g["image"])[0], cfg, beam_size=beam)
    assert tokens.dtype == torch.int64 and tokens.shape == (1, 20)
    np.testing.assert_array_equal(tokens.numpy(), g[f"beam{beam}_tokens"])
    np.testing.assert_allclose(lps.numpy(), g[f"beam{beam}_logprobs"], rtol=1e-3, atol=1e-3)
    # states are reset after decoding (containers.py:63-84)
    assert model.gri_feat is None and model.cap_generator.running_seq.shape == (1,)


def test_beam_search_on_cached_features(g7_model):
    """cached_features branch (transformer.py:64-67,139-142): the decoder alone reproduces the tokens from the
    reference's own visual features."""
    model, cfg = g7_model
    g = load("model_g7.npz")
    model.cached_features = True
    try:
        vis = {k: t(g[k]) for k in ("gri_feat", "gri_mask", "reg_feat", "reg_mask")}
        with oracle_ops(), torch.no_grad():
            tokens, _ = model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=1)
    finally:
        model.cached_features = False
    np.testing.assert_array_equal(tokens.numpy(), g["beam5_tokens"])


def test_one_xe_step_loss_grads_and_unused_set():
    from grit_amd.engine.caption_engine import build_optimizers
    from grit_amd.utils.misc import NestedTensor
    g = load("step_g8.npz")
    ref = json.load(open(os.path.join(GOLDEN, "step_g8.json")))
    model, cfg = build_model(3, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train()
    disable_drop_path(model)
    caps = t(g["caps"])
    with oracle_ops():
        out = model(NestedTensor(t(g["images"]), t(g["mask"])), caps)
        loss = torch.nn.NLLLoss(ignore_index=1)(out[:, :-1].reshape(-1, out.shape[-1]), caps[:, 1:].reshape(-1))
        loss.backward()
    assert abs(loss.item() - ref["loss"]) < 1e-4 * abs(ref["loss"])
    params = dict(model.named_parameters())
    nograd = sorted(n for n, p in params.items() if p.requires_grad and p.grad is None)
    assert nograd == ref["no_grad"]  # the static unused set (fc_alpha2, dead norms, class/bbox heads, ...)
    norms = {}
    for n, p in params.items():
        if p.requires_grad and p.grad is not None:
            top = '.'.join(n.split('.')[:2]) if n.startswith('detector') else n.split('.')[0]
            norms[top] = norms.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    for k, v in ref["grad_norms"].items():
        assert abs(norms[k]**0.5 - v) < 2e-3 * v, (k, norms[k]**0.5, v)
    for key in g.files:
        if key.startswith("grad:"):
            got = params[key[5:]].grad.flatten()[:64].numpy()
            np.testing.assert_allclose(got, g[key], rtol=2e-3, atol=1e-5 + 2e-3 * np.abs(g[key]).max())
    # optimizer split: two Adams, 'detector' in name -> backbone optimizer; weight decay effectively 0 (Q7)
    opts = build_optimizers(model, cfg, mode='xe')
    n_model = sum(len(gp['params']) for gp in opts['model'].param_groups)
    n_back = sum(len(gp['params']) for gp in opts['backbone'].param_groups)
    assert n_model == sum(1 for n, p in params.items() if p.requires_grad and 'detector' not in n)
    assert n_back == sum(1 for n, p in params.items() if p.requires_grad and 'detector' in n)
    assert opts['model'].param_groups[0]['lr'] == 1e-4 and opts['backbone'].param_groups[0]['lr'] == 1e-5
    assert all(gp['weight_decay'] == 0 for gp in opts['model'].param_groups + opts['backbone'].param_groups)


def test_one_self_critical_step_matches_reference():
    """Next-row N2: train_sc_step (reference engine/caption_engine.py:421-449) -- beam search WITH gradient, out_size =
    beam, reward injected -- against fixture G9 made by the reference model: same beams, same log-probs, same loss, same
    gradients; the step then updates both optimizers."""
    from grit_amd.engine.caption_engine import build_optimizers, train_sc_step
    from grit_amd.utils.misc import NestedTensor
    g = load("sc_g9.npz")
    ref = json.load(open(os.path.join(GOLDEN, "sc_g9.json")))
    model, cfg = build_model(3, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train()
    disable_drop_path(model)
    B, beam, T = g["tokens"].shape
    cfg.model.beam_size, cfg.model.beam_len = beam, T
    opts = build_optimizers(model, cfg, mode='sc')
    seen = {}

    def reward_fn(tokens, batch):
        seen['tokens'] = tokens.clone()
        return t(g["reward"])

    before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    images = t(g["images"])
    batch = {'samples': NestedTensor(images, torch.zeros(images.shape[0], *images.shape[-2:], dtype=torch.bool))}
    with oracle_ops():
        loss, reward, baseline = train_sc_step(model, batch, opts, reward_fn, cfg)
    assert np.array_equal(seen['tokens'].numpy(), g["tokens"])  # identical beams, all `beam` of them per image
    assert abs(loss.item() - ref["loss"]) < 2e-4 * abs(ref["loss"]) + 1e-7
    assert abs(reward.item() - g["reward"].mean()) < 1e-6 and abs(baseline.item() - g["reward"].mean()) < 1e-6
    params = dict(model.named_parameters())
    norms = {}
    for n, p in params.items():
        if p.requires_grad and p.grad is not None:
            top = '.'.join(n.split('.')[:2]) if n.startswith('detector') else n.split('.')[0]
            norms[top] = norms.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    for k, v in ref["grad_norms"].items():
        assert abs(norms[k]**0.5 - v) < 3e-3 * v, (k, norms[k]**0.5, v)
    for key in g.files:
        if key.startswith("grad:"):
            got = params[key[5:]].grad.flatten()[:64].numpy()
            np.testing.assert_allclose(got, g[key], rtol=3e-3, atol=1e-6 + 3e-3 * np.abs(g[key]).max())
    changed = sum(1 for n, p in params.items() if n in before and p.grad is not None and not torch.equal(p.detach(), before[n]))
    assert changed > 100  # both Adams stepped


def test_caption_strings_match_reference_decode(tmp_path):
    """BASELINE config 1, last hop: token ids -> caption string.  inference_caption.decode (cut at the first <eos>, join
    with spaces) against the reference's TextField.decode on the G7 beams and on edge rows (eos first, <unk>/<pad>/<bos>
    inside a caption), with the reference's word list (fixture G10)."""
    from inference_caption import decode
    g10 = json.load(open(os.path.join(GOLDEN, "vocab_g10.json")))
    g7 = load("model_g7.npz")
    vocab = tmp_path / "vocab.json"
    vocab.write_text(json.dumps({"itos": g10["itos"]}))
    assert len(g10["itos"]) == 10201 and g10["itos"][3] == g10["eos_token"] == "<eos>"
    assert decode(t(g7["beam1_tokens"]), str(vocab)) == g10["beam1"]
    assert decode(t(g7["beam5_tokens"]), str(vocab)) == g10["beam5"]
    assert decode(torch.tensor(g10["extra_tokens"]), str(vocab)) == g10["extra"]


def test_teacher_forcing_and_xe_step_on_cached_features(g7_model):
    """Next-row N3, model side: the cached-feature ("freezing") training mode (transformer.py:64-67 with
    model.cached_features = True; features as tools/extract_features.py stores them: gri_feat [N, fh*fw, 1024] f32,
    gri_mask [N, 1, 1, fh*fw] bool, reg_feat [N, 150, 512], reg_mask [N, 1, 1, 150]).  From the reference's own visual
    features the decoder reproduces the reference's teacher-forcing log-probs, and an XE step in this mode trains the
    grid net + caption generator only."""
    from grit_amd.engine.caption_engine import build_optimizers, train_xe_step
    model, cfg = g7_model
    g = load("model_g7.npz")
    vis = {k: t(g[k]) for k in ("gri_feat", "gri_mask", "reg_feat", "reg_mask")}
    before = {n: p.detach().clone() for n, p in model.named_parameters()}  # the fixture model is shared: restored below
    model.cached_features = True
    try:
        with oracle_ops(), torch.no_grad():
            lp = model(vis, t(g["seq"]))
        top = lp.topk(16, -1)
        np.testing.assert_array_equal(top.indices[..., :4].numpy(), g["tf_top_idx"][..., :4])
        np.testing.assert_allclose(top.values.numpy(), g["tf_top_val"], rtol=1e-4, atol=1e-4)
        model.train()
        opts = build_optimizers(model, cfg, mode='xe')
        with oracle_ops():
            loss = train_xe_step(model, {'samples': vis, 'captions': t(g["seq"])}, opts, torch.nn.NLLLoss(ignore_index=1))
        assert torch.isfinite(loss)
        moved = {n for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n])}
        assert moved and all(not n.startswith('detector.') for n in moved)
        assert any(n.startswith('grid_net.') for n in moved) and any(n.startswith('cap_generator.') for n in moved)
    finally:
        model.cached_features = False
        model.eval()
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.copy_(before[n])
