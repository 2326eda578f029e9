"""grit_amd/ops/glue.py (fused element-wise glue of the decoders' training step) against the composed torch forms they replace,
which are the reference's own lines: ms_deform_attn.py:97-113, det_module.py:40-53, det_module.py:302-304 / pos_embed.py:44-48,
cap_generator.py:44-56.  Values AND gradients; fp32 at 1e-6, bf16 at bf16 resolution."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("ref_dim", [4, 2])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L,P", [(4, 4), (3, 2), (4, 8)])
def test_sampling_geometry_matches_module_arithmetic(ref_dim, dtype, L, P):
    from grit_amd.ops.glue import sampling_geometry
    g = torch.Generator().manual_seed(L * 10 + P)
    N, Lq, M = 3, 37, 8
    shapes = torch.tensor([[40, 30], [20, 15], [10, 8], [5, 4]][:L], device=DEV)
    off = torch.randn(N, Lq, M * L * P * 2, generator=g).to(DEV, dtype).requires_grad_(True)
    log = torch.randn(N, Lq, M * L * P, generator=g).to(DEV, dtype).requires_grad_(True)
    ref = torch.rand(N, Lq, L, ref_dim, generator=g).to(DEV)
    c_loc = torch.randn(N, Lq, M, L, P, 2, generator=g).to(DEV)
    c_aw = torch.randn(N, Lq, M, L, P, generator=g).to(DEV)
    loc, aw = sampling_geometry(off, log, ref, shapes, M, L, P)
    assert loc.dtype == aw.dtype == torch.float32
    ((loc * c_loc).sum() + (aw * c_aw).sum()).backward()
    got = (loc.detach(), aw.detach(), off.grad.clone(), log.grad.clone())
    off.grad = log.grad = None
    # the module's composed form (fp32 on the same inputs)
    o = off.view(N, Lq, M, L, P, 2).float()
    w = F.softmax(log.view(N, Lq, M, L * P).float(), -1).view(N, Lq, M, L, P)
    if ref_dim == 2:
        wh = torch.stack([shapes[..., 1], shapes[..., 0]], -1).float()
        rl = ref[:, :, None, :, None, :] + o / wh[None, None, None, :, None, :]
    else:
        rl = ref[:, :, None, :, None, :2] + o / P * ref[:, :, None, :, None, 2:] * 0.5
    ((rl * c_loc).sum() + (w * c_aw).sum()).backward()
    torch.testing.assert_close(got[0], rl.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got[1], w.detach(), rtol=1e-5, atol=1e-6)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-3)
    torch.testing.assert_close(got[2].float(), off.grad.float(), **tol)
    torch.testing.assert_close(got[3].float(), log.grad.float(), **tol)


@pytest.mark.parametrize("ref_dim", [4, 2])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_box_refine_matches_reference_arithmetic(ref_dim, dtype):
    from grit_amd.ops.glue import box_refine
    from grit_amd.utils.misc import inverse_sigmoid
    g = torch.Generator().manual_seed(ref_dim)
    delta = (3 * torch.randn(5, 150, 4, generator=g)).to(DEV, dtype)
    ref = torch.rand(5, 150, ref_dim, generator=g).to(DEV)
    ref[0, :4] = torch.tensor([0.0, 1.0, 1e-7, 1 - 1e-7][:ref_dim] + [0.5] * max(0, ref_dim - 4), device=DEV)[:ref_dim]  # clamp branches
    got = box_refine(delta, ref)
    d = delta.float()
    want = (d + inverse_sigmoid(ref)).sigmoid() if ref_dim == 4 else torch.cat([d[..., :2] + inverse_sigmoid(ref), d[..., 2:]], -1).sigmoid()
    assert got.dtype == torch.float32
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_relu_dropout_value_gradient_and_mask(dtype, p):
    from grit_amd.ops.glue import relu_dropout
    g = torch.Generator().manual_seed(3)
    x = torch.randn(32, 150, 1024, generator=g).to(DEV, dtype).requires_grad_(True)
    cot = torch.randn(32, 150, 1024, generator=g).to(DEV, dtype)
    y = relu_dropout(x, p, True)
    y.backward(cot)
    pos = x.detach() > 0
    keep = (y.detach() != 0) | ~pos  # where x > 0 a zero output means "dropped"
    if p == 0:
        assert torch.equal(y.detach(), F.relu(x.detach()))
        assert torch.equal(x.grad, torch.where(pos, cot, torch.zeros_like(cot)))
    else:
        rate = 1.0 - keep[pos].float().mean().item()
        assert abs(rate - p) < 5e-3, rate
        scale = 1.0 / (1.0 - p)
        want = torch.where(pos & keep, x.detach().float() * scale, torch.zeros_like(x, dtype=torch.float32))
        torch.testing.assert_close(y.detach().float(), want, rtol=1e-2 if dtype == torch.bfloat16 else 1e-6, atol=1e-6)
        wantg = torch.where(pos & keep, cot.float() * scale, torch.zeros_like(cot, dtype=torch.float32))
        torch.testing.assert_close(x.grad.float(), wantg, rtol=1e-2 if dtype == torch.bfloat16 else 1e-6, atol=1e-6)
        y2 = relu_dropout(x, p, True)  # another call draws another mask
        assert not torch.equal(y2.detach() != 0, y.detach() != 0)
    assert torch.equal(relu_dropout(x.detach(), p, False), F.relu(x.detach()))  # eval: plain ReLU


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gated_merge_training_node_matches_composed_form(dtype):
    """Both gates through fc_alpha1 (the reference's quirk, cap_generator.py:48-49), PAD rows masked."""
    from grit_amd.ops.glue import gated_merge_train
    from grit_amd.ops.linear import Linear
    g = torch.Generator().manual_seed(7)
    B, T, d = 6, 20, 512
    fc = Linear(2 * d, d).to(DEV, dtype)
    mk = lambda: torch.randn(B, T, d, generator=g).to(DEV, dtype).requires_grad_(True)
    s, e1, e2 = mk(), mk(), mk()
    mask = (torch.rand(B, T, 1, generator=g) > 0.25).to(DEV, dtype)
    cot = torch.randn(B, T, d, generator=g).to(DEV, dtype)

    def grads():
        out = [t.grad.clone() for t in (s, e1, e2, fc.weight, fc.bias)]
        for t in (s, e1, e2, fc.weight, fc.bias):
            t.grad = None
        return out

    got = gated_merge_train(s, e1, e2, mask, fc)
    got.backward(cot)
    g_got = grads()
    a, b = e1 * mask, e2 * mask
    g1 = torch.sigmoid(fc(torch.cat([s, a], -1)))
    g2 = torch.sigmoid(fc(torch.cat([s, b], -1)))
    want = ((a * g1 + b * g2) / np.sqrt(2)) * mask
    want.backward(cot)
    g_want = grads()
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(got.detach().float(), want.detach().float(), **tol)
    for name, x, y in zip(("d_self", "d_enc1", "d_enc2", "dW", "db"), g_got, g_want):
        scale = y.float().abs().max().item()
        err = (x.float() - y.float()).abs().max().item()
        assert err <= (1e-4 if dtype == torch.float32 else 2e-2) * scale + 1e-6, (name, err, scale)
    assert float(got.detach()[(mask == 0).expand_as(got)].abs().max()) == 0.0
