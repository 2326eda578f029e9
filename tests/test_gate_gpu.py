"""grit_gate_pack / grit_gate_fuse (grit_amd/ops/gate.py) against the composed arithmetic of the reference's
ParallelAttentionLayer.forward (models/caption/cap_generator.py:44-56): bit-exact in bf16 (every intermediate rounded where the
composed form rounds it), to the last place or two in fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _composed(self_att, enc1, enc2, mask_pad, fc):
    enc1 = enc1 * mask_pad
    enc2 = enc2 * mask_pad
    gate1 = torch.sigmoid(fc(torch.cat([self_att, enc1], -1)))
    gate2 = torch.sigmoid(fc(torch.cat([self_att, enc2], -1)))
    fused = (enc1 * gate1 + enc2 * gate2) / np.sqrt(2)
    return fused * mask_pad


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,T", [(320, 1), (64, 1), (7, 20), (1, 1)])
def test_gated_merge_is_the_composed_form(dtype, rows, T):
    from grit_amd.ops import gate
    torch.manual_seed(rows + T)
    d = 512
    fc = torch.nn.Linear(2 * d, d).cuda().to(dtype)
    self_att = torch.randn(rows, T, d, device='cuda').to(dtype)
    enc1 = (torch.randn(rows, T, d, device='cuda') * 2).to(dtype)
    enc2 = (torch.randn(rows, T, d, device='cuda') * 2).to(dtype)
    mask_pad = (torch.rand(rows, T, 1, device='cuda') > 0.2).to(dtype)
    it = torch.int16 if dtype == torch.bfloat16 else torch.int32
    with torch.no_grad():
        assert gate.supported(self_att, enc1, enc2, mask_pad, fc)
        e1, e2 = enc1 * mask_pad, enc2 * mask_pad
        # pack == the two concatenations, stacked
        X = gate.pack(self_att, enc1, enc2, mask_pad)
        want_X = torch.cat([torch.cat([self_att, e1], -1), torch.cat([self_att, e2], -1)], 0).view(2 * rows * T, 2 * d)
        assert torch.equal(X.view(it), want_X.view(it))
        # fuse == the composed element-wise chain on the SAME gate pre-activations, bit for bit (signed zeros included)
        G = fc(X)
        got = gate.fuse(enc1, enc2, G, mask_pad)
        Gv = G.view(2, rows, T, d)
        want = ((e1 * torch.sigmoid(Gv[0]) + e2 * torch.sigmoid(Gv[1])) / np.sqrt(2)) * mask_pad
        if dtype == torch.bfloat16:
            assert torch.equal(got.view(it), want.view(it))
        else:  # fp32: the device exp / divide of this library and of torch's kernels may differ in the last place
            assert torch.allclose(got, want, rtol=2e-6, atol=1e-6)
        # end to end against the reference's form (two GEMMs of R rows instead of one of 2R: the library may choose another
        # kernel / reduction order, which moves a gate pre-activation by an ulp and, where the two products cancel, more)
        full = gate.gated_merge(self_att, enc1, enc2, mask_pad, fc)
        ref = _composed(self_att, enc1, enc2, mask_pad, fc)
        tol = 6e-2 if dtype == torch.bfloat16 else 1e-5
        assert (full.float() - ref.float()).abs().max().item() <= tol
    with torch.enable_grad():
        assert not gate.supported(self_att, enc1, enc2, mask_pad, fc)  # training keeps the differentiable composed form


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,cur,beam,t_old", [(64, 5, 5, 7), (64, 1, 5, 1), (3, 5, 5, 19), (2, 3, 3, 4), (5, 1, 1, 0), (4, 5, 5, 0)])
def test_kv_append_is_gather_plus_cat(dtype, B, cur, beam, t_old):
    """grit_kv_append (grit_amd/ops/kv_cache.py) == re-gather of the cache by the surviving beam (transformer.py:229) followed by
    the append of the new key / value (attention.py:166-181); byte copies, so exact."""
    from grit_amd.ops import kv_cache
    g = torch.Generator(device='cuda').manual_seed(B * 100 + t_old)
    d = 512
    qkv = torch.randn(B * beam, 1, 3 * d, device='cuda', generator=g).to(dtype)      # new k / v are slices of the fused projection
    new_k, new_v = qkv[..., d:2 * d], qkv[..., 2 * d:]
    if t_old == 0:
        got_k, got_v = kv_cache.append(None, None, None, new_k, new_v, beam=1 if cur == beam == 1 else beam)
        assert torch.equal(got_k, new_k) and torch.equal(got_v, new_v)
        return
    old_k = torch.randn(B * cur, t_old, d, device='cuda', generator=g).to(dtype)
    old_v = torch.randn(B * cur, t_old, d, device='cuda', generator=g).to(dtype)
    src = torch.randint(0, cur, (B, beam), device='cuda', generator=g)
    got_k, got_v = kv_cache.append(old_k, old_v, src, new_k, new_v, beam=beam)
    idx = src.view(B, beam, 1, 1).expand(B, beam, t_old, d)
    want_k = torch.cat([torch.gather(old_k.view(B, cur, t_old, d), 1, idx).view(B * beam, t_old, d), new_k], 1)
    want_v = torch.cat([torch.gather(old_v.view(B, cur, t_old, d), 1, idx).view(B * beam, t_old, d), new_v], 1)
    assert torch.equal(got_k, want_k) and torch.equal(got_v, want_v)
    if cur == beam:  # no index: every beam continues itself
        same_k, same_v = kv_cache.append(old_k, old_v, None, new_k, new_v, beam=1)
        assert torch.equal(same_k, torch.cat([old_k, new_k], 1)) and torch.equal(same_v, torch.cat([old_v, new_v], 1))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_decode_step_inputs_match_get_seq_inputs(dtype):
    """grit_decode_step_inputs (CaptionGenerator._step_inputs) against the reference-order get_seq_inputs + embedding sum in stateful
    mode, four consecutive steps with padding tokens in the stream: bit-exact embeddings, masks, step counters."""
    import grit_amd.models.caption.cap_generator as CG
    torch.manual_seed(3)
    gen = CG.CaptionGenerator(vocab_size=1000, max_len=54, n_layers=1, pad_idx=1).cuda().to(dtype).eval()
    ref = CG.CaptionGenerator(vocab_size=1000, max_len=54, n_layers=1, pad_idx=1).cuda().to(dtype).eval()
    ref.load_state_dict(gen.state_dict())
    R = 40
    with torch.no_grad(), gen.statefulness(R), ref.statefulness(R):
        for step in range(4):
            tokens = torch.randint(0, 6, (R, 1), device='cuda')  # a sixth of the tokens are <pad> (index 1)
            fused = gen._step_inputs(tokens)
            assert fused is not None
            x, mask_x, mask_pad = fused
            wx, wmask_x, wmask_pad = ref.get_seq_inputs(tokens)
            assert torch.equal(x, wx)
            assert mask_x.dtype == torch.bool and torch.equal(mask_x, wmask_x) and mask_x.shape == (R, 1, 1, step + 1)
            assert torch.equal(mask_pad.float(), wmask_pad.float())
            assert torch.equal(gen.running_seq, ref.running_seq) and int(gen.running_seq[0]) == step + 1
            # what Transformer.iter does between steps: every state re-gathered by the surviving beam (here: a permutation)
            perm = torch.randperm(R, device='cuda')
            for m in (gen, ref):
                m.apply_to_states(lambda t: t[perm] if t.shape[0] == R else t)
                for layer in m.layers:  # the step counter the fused path checks against the position table
                    layer.self_att.timestep += 1
