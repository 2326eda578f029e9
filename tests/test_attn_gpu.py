"""HIP attention core (grit_attn_*; fp32 arithmetic) against the PyTorch oracle and the reference fixture G6."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from tests.helpers import deterministic_fill_, load, t

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, Tq, Nk, H=8, seed=0, mask_kind=None, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Tq, H, 64, generator=g)
    k = torch.randn(B, Nk, H, 64, generator=g)
    v = torch.randn(B, Nk, H, 64, generator=g)
    mask = None
    if mask_kind == "causal":  # [B,1,T,T] causal | pad, like CaptionGenerator.get_seq_inputs
        pad = torch.zeros(B, Nk, dtype=torch.bool)
        pad[-1, Nk - 3:] = True
        mask = torch.triu(torch.ones(Tq, Nk, dtype=torch.bool), 1)[None, None] | pad[:, None, None, :]
    elif mask_kind == "key":  # [B,1,1,Nk] like gri_mask
        mask = torch.zeros(B, 1, 1, Nk, dtype=torch.bool)
        mask[0, ..., Nk // 2:] = True
    elif mask_kind == "shared":  # [1,1,Tq,Nk]
        mask = (torch.rand(1, 1, Tq, Nk, generator=g) < 0.3)
        mask[..., 0] = False
    return q.to(dtype), k.to(dtype), v.to(dtype), mask


def _grads(fn, q, k, v, cot):
    q, k, v = (x.clone().requires_grad_(True) for x in (q, k, v))
    out = fn(q, k, v)
    out.backward(cot)
    return out.detach(), q.grad, k.grad, v.grad


@pytest.mark.parametrize("B,Tq,Nk,mask_kind", [(2, 20, 20, "causal"), (3, 20, 100, "key"), (2, 20, 150, None),
                                               (2, 150, 150, None), (1, 1, 7, None), (2, 54, 54, "causal"),
                                               (1, 33, 256, "shared"), (2, 5, 65, "key")])
def test_forward_backward_vs_oracle_fp32(B, Tq, Nk, mask_kind):
    """north_star: decoder within 1e-4 fp32."""
    from grit_amd.ops.attention import attention
    q, k, v, mask = _case(B, Tq, Nk, mask_kind=mask_kind)
    cot = torch.randn(B, Tq, 512, generator=torch.Generator().manual_seed(9))
    ref = _grads(lambda a, b, c: torch_ref.attention(a, b, c, mask), q.double(), k.double(), v.double(), cot.double())
    dm = None if mask is None else mask.to(DEV)
    got = _grads(lambda a, b, c: attention(a, b, c, dm), q.to(DEV), k.to(DEV), v.to(DEV), cot.to(DEV))
    for name, r, o in zip(("out", "dq", "dk", "dv"), ref, got):
        np.testing.assert_allclose(o.cpu().numpy(), r.float().numpy(), rtol=1e-4, atol=1e-4, err_msg=name)


def test_strided_projection_slices_and_bf16():
    """q and k as slices of one packed projection (det_module query_self_attention), bf16 storage."""
    from grit_amd.ops.attention import attention
    g = torch.Generator().manual_seed(4)
    qk = torch.randn(2, 150, 1024, generator=g).to(DEV)
    vv = torch.randn(2, 150, 512, generator=g).to(DEV)
    q, k = qk[..., :512].view(2, 150, 8, 64), qk[..., 512:].view(2, 150, 8, 64)
    out = attention(q, k, vv.view(2, 150, 8, 64))
    ref = torch_ref.attention(q.cpu().double(), k.cpu().double(), vv.cpu().double().view(2, 150, 8, 64))
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), rtol=1e-4, atol=1e-4)
    o16 = attention(q.bfloat16(), k.bfloat16(), vv.bfloat16().view(2, 150, 8, 64))
    assert o16.dtype == torch.bfloat16
    ref16 = torch_ref.attention(q.bfloat16().cpu().double(), k.bfloat16().cpu().double(),
                                vv.bfloat16().cpu().double().view(2, 150, 8, 64))
    np.testing.assert_allclose(o16.float().cpu().numpy(), ref16.float().numpy(), rtol=2e-2, atol=2e-2)


def test_dropout_is_a_scaled_bernoulli_mask_and_backward_matches():
    """V = I (Nk = 64): the output IS the dropped P.  Same seed -> same mask; gradient is that of the masked graph."""
    from grit_amd.ops.attention import _AttentionFn
    B, Tq, Nk, H, p = 2, 40, 64, 8, 0.25
    q, k, _, _ = _case(B, Tq, Nk)
    v = torch.eye(64).view(1, 64, 1, 64).expand(B, 64, H, 64).contiguous()
    q, k, v = q.to(DEV), k.to(DEV), v.to(DEV)
    scale = 0.125
    pd = _AttentionFn.apply(q, k, v, None, scale, p, 1234, None).view(B, Tq, H, 64)
    pd2 = _AttentionFn.apply(q, k, v, None, scale, p, 1234, None).view(B, Tq, H, 64)
    pd3 = _AttentionFn.apply(q, k, v, None, scale, p, 99, None).view(B, Tq, H, 64)
    assert torch.equal(pd, pd2) and not torch.equal(pd, pd3)
    P = torch.softmax(torch.einsum("bqhd,bkhd->bqhk", q, k) * scale, -1)
    kept = pd != 0
    frac = 1 - kept.float().mean().item()
    assert abs(frac - p) < 0.01
    np.testing.assert_allclose(pd[kept].cpu().numpy(), (P / (1 - p))[kept].cpu().numpy(), rtol=1e-4, atol=1e-6)
    # backward against autograd through the same mask
    M = kept.float() / (1 - p)
    cot = torch.randn(B, Tq, 512, device=DEV)
    gq, gk, gv = (x.clone().requires_grad_(True) for x in (q, k, v))
    _AttentionFn.apply(gq, gk, gv, None, scale, p, 1234, None).backward(cot)
    rq, rk, rv = (x.clone().requires_grad_(True) for x in (q, k, v))
    Pm = torch.softmax(torch.einsum("bqhd,bkhd->bqhk", rq, rk) * scale, -1) * M
    torch.einsum("bqhk,bkhd->bqhd", Pm, rv).reshape(B, Tq, 512).backward(cot)
    for a, b_ in ((gq.grad, rq.grad), (gk.grad, rk.grad), (gv.grad, rv.grad)):
        np.testing.assert_allclose(a.cpu().numpy(), b_.cpu().numpy(), rtol=1e-3, atol=1e-4)


def test_parallel_attention_layer_fixture_on_gpu():
    """G6 through the HIP path: ParallelAttentionLayer (3 attentions with causal|pad / key masks) vs the reference."""
    from grit_amd.models.caption.cap_generator import ParallelAttentionLayer
    g = load("attn_g6.npz")
    layer = deterministic_fill_(ParallelAttentionLayer(512, 8, 2048, dropout=0.1), "g6.").eval().to(DEV)
    with torch.no_grad():
        out = layer(*(t(g[k], device=DEV) for k in ("x", "y1", "y2", "mask_pad", "mask_x", "mask_y1", "mask_y2")))
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-4)


def test_fully_masked_row_is_nan_like_reference_and_errors():
    from grit_amd.ops.attention import attention
    q, k, v, _ = _case(1, 2, 5)
    mask = torch.zeros(1, 1, 2, 5, dtype=torch.bool)
    mask[0, 0, 1] = True
    out = attention(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV))
    assert torch.isnan(out[0, 1]).all() and not torch.isnan(out[0, 0]).any()
    with pytest.raises(RuntimeError):
        attention(q.to(DEV)[..., :32], k.to(DEV)[..., :32], v.to(DEV)[..., :32])
    big = torch.randn(1, 300, 8, 64, device=DEV)
    with pytest.raises(RuntimeError, match="not supported"):
        attention(q.to(DEV), big, big)


# ---------------------------------------------------------------------------------------------------------------
# bf16 storage -> matrix-core kernels (attn_mfma.hip); oracle evaluated in fp64 on the same bf16-rounded inputs
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Tq,Nk,mask_kind", [(2, 20, 20, "causal"), (3, 20, 100, "key"), (2, 20, 150, None),
                                               (2, 150, 150, None), (1, 1, 7, None), (2, 54, 54, "causal"),
                                               (1, 33, 160, "shared"), (2, 5, 65, "key"), (1, 160, 17, None)])
def test_bf16_mfma_forward_backward_vs_oracle(B, Tq, Nk, mask_kind):
    from grit_amd.ops.attention import attention
    q, k, v, mask = _case(B, Tq, Nk, mask_kind=mask_kind, dtype=torch.bfloat16)
    cot = torch.randn(B, Tq, 512, generator=torch.Generator().manual_seed(9)).bfloat16()
    ref = _grads(lambda a, b, c: torch_ref.attention(a, b, c, mask), q.double(), k.double(), v.double(), cot.double())
    dm = None if mask is None else mask.to(DEV)
    got = _grads(lambda a, b, c: attention(a, b, c, dm), q.to(DEV), k.to(DEV), v.to(DEV), cot.to(DEV))
    for name, r, o in zip(("out", "dq", "dk", "dv"), ref, got):
        assert o.dtype == torch.bfloat16
        r, o = r.float(), o.float().cpu()
        scale = r.abs().max().item() + 1e-6
        assert (o - r).abs().max().item() < 3e-2 * scale, (name, (o - r).abs().max().item(), scale)
        assert (o - r).abs().mean().item() < 4e-3 * scale, name


def test_bf16_mfma_dropout_consistency():
    """MFMA path: the keep-mask of forward and backward is the same function of the seed."""
    from grit_amd.ops.attention import _AttentionFn
    B, Tq, Nk, H, p = 2, 48, 64, 8, 0.2
    q, k, _, _ = _case(B, Tq, Nk, dtype=torch.bfloat16)
    v = torch.eye(64).view(1, 64, 1, 64).expand(B, 64, H, 64).contiguous().bfloat16()
    q, k, v = q.to(DEV), k.to(DEV), v.to(DEV)
    pd = _AttentionFn.apply(q, k, v, None, 0.125, p, 0, torch.tensor([77], device=DEV)).view(B, Tq, H, 64).float()
    P = torch.softmax(torch.einsum("bqhd,bkhd->bqhk", q.float(), k.float()) * 0.125, -1)
    kept = pd != 0
    dropped_frac = 1 - (kept | (P < 1e-3)).float().mean().item()  # tiny P may round to 0 in bf16
    assert abs(dropped_frac - p) < 0.02
    big = kept & (P > 1e-2)
    np.testing.assert_allclose(pd[big].cpu().numpy(), (P / (1 - p))[big].cpu().numpy(), rtol=2e-2)
    M = kept.float() / (1 - p)
    cot = torch.randn(B, Tq, 512, device=DEV).bfloat16()
    gq, gk, gv = (x.clone().requires_grad_(True) for x in (q, k, v))
    _AttentionFn.apply(gq, gk, gv, None, 0.125, p, 0, torch.tensor([77], device=DEV)).backward(cot)
    rq, rk, rv = (x.float().clone().requires_grad_(True) for x in (q, k, v))
    Pm = torch.softmax(torch.einsum("bqhd,bkhd->bqhk", rq, rk) * 0.125, -1) * M
    torch.einsum("bqhk,bkhd->bqhd", Pm, rv).reshape(B, Tq, 512).backward(cot.float())
    for a, b_ in ((gq.grad, rq.grad), (gk.grad, rk.grad), (gv.grad, rv.grad)):
        scale = b_.abs().max().item()
        assert (a.float() - b_).abs().max().item() < 4e-2 * scale
