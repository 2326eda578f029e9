"""Host-side modules + the PyTorch oracle ops against fixtures made from the imported reference modules
(G3 MSDeformAttn module, G4/G5 WindowAttention + BasicLayer, G6 ParallelAttentionLayer, state-dict surface)."""
import json
import os

import numpy as np
import torch

from tests.helpers import GOLDEN, build_model, deterministic_fill_, load, oracle_ops, t


def test_state_dict_surface_matches_reference():
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    for n in (3, 2):
        model, _ = build_model(n, fill=False)
        mine = {k: list(v.shape) for k, v in model.state_dict().items()}
        assert mine == ref[str(n)], set(mine) ^ set(ref[str(n)])
        assert sorted(k for k, p in model.named_parameters() if p.requires_grad) == ref[f"{n}_trainable"]
    assert len(ref["3"]) == 761 and len(ref["2"]) == 719  # SURVEY 8b probe


def test_msdeformattn_module_2d_and_4d_refs():
    from grit_amd.models.ops.modules import MSDeformAttn
    g = load("msda_module_g3.npz")
    mod = deterministic_fill_(MSDeformAttn(d_model=128, n_levels=3, n_heads=4, n_points=4), "g3.").double()
    sh, lsi = t(g["shapes"]), t(g["lsi"])
    with oracle_ops(), torch.no_grad():
        out2 = mod(t(g["query"]).double(), t(g["ref2"]).double(), t(g["src"]).double(), sh, lsi, None)
        out4 = mod(t(g["query"]).double(), t(g["ref4"]).double(), t(g["src"]).double(), sh, lsi, t(g["pad"]))
    np.testing.assert_allclose(out2.numpy(), g["out2"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out4.numpy(), g["out4"], rtol=1e-5, atol=1e-5)


def _basic_layer():
    from grit_amd.models.common.swin_model import BasicLayer, PatchMerging
    layer = BasicLayer(dim=128, depth=2, num_heads=4, window_size=12, drop_path=[0.0, 0.1], downsample=PatchMerging)
    return deterministic_fill_(layer, "g4.").eval()


def test_window_attention_reference_call_form():
    """WindowAttention.forward(x_windows, mask) (swin_model.py:155-186), without and with the shift mask."""
    g = load("win_g4.npz")
    layer = _basic_layer()
    attn = layer.blocks[1].attn
    with oracle_ops(), torch.no_grad():
        o0 = attn(t(g["xw"]), None)
        o1 = attn(t(g["xw"]), t(g["attn_mask"]))
    np.testing.assert_allclose(o0.numpy(), g["o_nomask"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o1.numpy(), g["o_mask"], rtol=1e-4, atol=1e-5)
    # the materialised mask helper equals the reference's BasicLayer mask
    np.testing.assert_array_equal(layer.attention_mask(20, 20).numpy(), g["attn_mask"])


def test_basic_layer_shift_pad_crop_merge():
    """Two blocks (no shift / shift 6) on a 20x20 map padded to 24x24, then PatchMerging: pins the token-order
    formulation (qkv on the un-partitioned map, pad tokens = qkv bias, analytic shift mask) against the reference's
    pad/roll/partition pipeline (swin_model.py:244-300, 414-456, 324-349)."""
    g = load("win_g4.npz")
    layer = _basic_layer()
    with oracle_ops(), torch.no_grad():
        x_out, H, W, x_down, Wh, Ww = layer(t(g["x"]), 20, 20)
    assert [H, W, Wh, Ww] == g["dims"].tolist()
    np.testing.assert_allclose(x_out.numpy(), g["x_out"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(x_down.numpy(), g["x_down"], rtol=1e-4, atol=2e-5)


def test_parallel_attention_layer_with_pad_tokens():
    """cap_generator.py:40-56 incl. the fc_alpha1-twice quirk; attention.py:166-184; pos_embed.py:44-48."""
    from grit_amd.models.caption.cap_generator import ParallelAttentionLayer
    g = load("attn_g6.npz")
    layer = deterministic_fill_(ParallelAttentionLayer(512, 8, 2048, dropout=0.1), "g6.").eval()
    with oracle_ops(), torch.no_grad():
        out = layer(t(g["x"]), t(g["y1"]), t(g["y2"]), t(g["mask_pad"]), t(g["mask_x"]), t(g["mask_y1"]), t(g["mask_y2"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-4, atol=1e-5)
    # fc_alpha2 really is dead: perturbing it changes nothing
    with oracle_ops(), torch.no_grad():
        layer.fc_alpha2.weight.add_(1.0)
        out2 = layer(t(g["x"]), t(g["y1"]), t(g["y2"]), t(g["mask_pad"]), t(g["mask_x"]), t(g["mask_y1"]), t(g["mask_y2"]))
    assert torch.equal(out, out2)


def test_get_seq_inputs_masks():
    from grit_amd.models.caption.cap_generator import CaptionGenerator
    g = load("attn_g6.npz")
    gen = CaptionGenerator(vocab_size=50, max_len=10, n_layers=1, pad_idx=1)
    x, mask_x, mask_pad = gen.get_seq_inputs(t(g["tokens"]))
    np.testing.assert_array_equal(mask_x.numpy(), g["mask_x"])
    np.testing.assert_array_equal(mask_pad.numpy(), g["mask_pad"])
    # positions 1..T, 0 on PAD -> the sinusoid row 0 is all zeros
    assert torch.equal(x[1, 4], gen.word_emb(torch.tensor(1)) + 0)


def test_modules_refuse_cpu_without_injection():
    """No silent fallback: on CPU tensors the product path raises like the reference's CPU stub."""
    import pytest
    layer = _basic_layer()
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        layer(torch.zeros(1, 400, 128), 20, 20)


def test_nested_tensor_contract():
    from grit_amd.utils.misc import nested_tensor_from_tensor_list, inverse_sigmoid
    a, b = torch.ones(3, 4, 6), torch.ones(3, 5, 2)
    nt = nested_tensor_from_tensor_list([a, b])
    assert nt.tensors.shape == (2, 3, 5, 6) and nt.mask.shape == (2, 5, 6)
    assert not nt.mask[0, :4, :6].any() and nt.mask[0, 4:].all() and nt.mask[1, :, 2:].all() and not nt.mask[1, :5, :2].any()
    assert nt.tensors[1, :, :, 2:].abs().sum() == 0
    x = torch.tensor([0.0, 0.25, 1.0, 2.0])
    np.testing.assert_allclose(inverse_sigmoid(x).numpy(), np.log(np.array([1e-5, 0.25, 1.0, 1.0]) / np.array([1.0, 0.75, 1e-5, 1e-5])), rtol=1e-5)


def test_cosine_scheduler_closed_form():
    """cap_scheduler.py:28-59 evaluated by hand for a 2-epoch x 10-it schedule."""
    import math
    from grit_amd.utils.cap_scheduler import CosineLRScheduler
    opt = torch.optim.Adam(torch.nn.Linear(2, 2).parameters(), lr=5e-4)
    s = CosineLRScheduler(opt, num_epochs=2, num_its_per_epoch=10, init_lr=5e-4, min_lr=1e-4, warmup_init_lr=1e-5)
    for k in range(1, 21):
        lr = s.step()
        if k < 10:
            a = k / 10
            want = (5e-4 - 1e-5) * (0.1 * (1 - a) + a) + 1e-5
        else:
            want = max(1e-4, (5e-4 - 1e-4) * (1 + math.cos(math.pi * k / 20)) / 2 + 1e-4)
        assert abs(lr - want) < 1e-12 and opt.param_groups[0]['lr'] == lr
    s2 = CosineLRScheduler(opt, 2, 10)
    s2.load_state_dict(s.state_dict())
    assert s2.global_steps == 20 and s2.init_lr == 5e-4


def test_weight_gradient_slab_policy():
    """split_k (grit_amd/ops/linear.py): slabs divide M, powers of two on the GRIT shapes, 4 slabs for Swin stage 3."""
    from grit_amd.ops.linear import split_k
    assert split_k(51200) == 16 and split_k(204800) == 64 and split_k(272000) == 64 and split_k(819200) == 64
    assert split_k(12800) == 4 and split_k(6400) == 4 and split_k(4800) == 1 and split_k(25600) == 8
    for M in (25600, 38400, 51200, 61440, 100000, 204800, 272000):
        assert M % split_k(M) == 0 and 1 <= split_k(M) <= 64
