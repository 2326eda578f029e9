"""grit_gemm_bf16_nt (fused-epilogue bf16 MFMA GEMM) and the Mlp node built on it, through the C ABI.

Reference: the same contraction in float32 on the bf16 inputs (torch on the device), the reference's exact GELU
(torch.nn.functional.gelu, models/common/swin_model.py:31-37) and its autograd derivative.  Outputs are bf16: the tolerance is
half a bf16 ulp of the result (2^-8 relative) plus accumulation-order noise."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2.0 ** -7  # of the tensor's max magnitude (bf16 output rounding 2^-9 relative, the rest is margin)


def _close(got, ref, tol=TOL):
    scale = ref.abs().max().item()
    err = (got.float() - ref).abs().max().item()
    assert err <= tol * scale, (err, scale)


def _inputs(M, N, K, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(M, K, device='cuda', generator=g).bfloat16()
    w = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device='cuda', generator=g).bfloat16()
    return x, w, b


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1000, 512, 128), (4096 + 17, 256, 512), (37, 1024, 192), (70000, 768, 96)])
def test_gemm_epilogues(M, N, K, variant):
    from grit_amd.ops import gemm as G
    from grit_amd.ops.linear import slab_sum
    if variant in (2, 4) and K % 64:
        pytest.skip("BK = 64 configuration")
    if variant in (4, 5) and N % 256:
        pytest.skip("256-column tiles")
    x, w, b = _inputs(M, N, K)
    ref = x.float() @ w.float().t()
    _close(G.gemm_nt(x, w, G.NONE, variant=variant), ref)
    _close(G.gemm_nt(x, w, G.BIAS, bias=b, variant=variant), ref + b.float())
    pre = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
    act = G.gemm_nt(x, w, G.BIAS_GELU, bias=b, aux=pre, variant=variant)
    _close(pre, ref + b.float())
    _close(act, F.gelu(ref + b.float()))
    _close(G.gemm_nt(x, w, G.BIAS_GELU, bias=b, variant=variant), F.gelu(ref + b.float()))  # no pre-activation kept
    # GELU' epilogue + column sums (rows past M must not leak into the sums)
    aux = torch.randn(M, N, device='cuda').bfloat16()
    part = torch.full((-(-M // 128), N), float('nan'), device='cuda')
    d = G.gemm_nt(x, w, G.DGELU, aux=aux, colsum=part, variant=variant)
    a32 = aux.float().requires_grad_(True)
    F.gelu(a32).backward(ref)
    _close(d, a32.grad)
    db = slab_sum(part.unsqueeze(0), torch.float32)[0]
    assert torch.isfinite(part).all()
    ref_db = a32.grad.sum(0)
    assert (db - ref_db).abs().max().item() <= 2e-3 * ref_db.abs().max().item() + 1e-3


def test_gemm_full_size_property():
    """BASELINE shape of Swin stage 2 (M = 32 * 40 * 40, 512 -> 2048): linearity in the bias and exact row independence --
    a row's result does not depend on which tile / workgroup computed it."""
    from grit_amd.ops import gemm as G
    M, N, K = 51200, 2048, 512
    x, w, b = _inputs(M, N, K, seed=1)
    full = G.gemm_nt(x, w, G.BIAS, bias=b)
    rows = torch.tensor([0, 255, 256, 12345, 51199], device='cuda')
    part = G.gemm_nt(x[rows].contiguous(), w, G.BIAS, bias=b)
    assert torch.equal(full[rows], part)
    _close(full[rows], x[rows].float() @ w.float().t() + b.float())


def test_gemm_rejects_bad_shapes():
    from grit_amd.ops import gemm as G
    from grit_amd.lib import GritHipError
    x, w, b = _inputs(64, 96, 64)  # N % 128 != 0
    with pytest.raises(GritHipError):
        G.gemm_nt(x, w, G.NONE)
    x, w, b = _inputs(64, 128, 48)  # K % 32 != 0
    with pytest.raises(GritHipError):
        G.gemm_nt(x, w, G.NONE)


@pytest.mark.parametrize("with_norm", [False, True])
def test_fused_mlp_matches_module(with_norm):
    """grit_amd.ops.mlp nodes against the unfused Mlp module (library GEMMs + torch GELU), forward and all gradients."""
    from grit_amd.models.common.swin_model import Mlp
    from grit_amd.ops.layer_norm import LayerNorm
    from grit_amd.ops import mlp as M
    torch.manual_seed(0)
    B, T, C = 3, 1000, 256
    mod = Mlp(C, 4 * C).cuda().bfloat16()
    norm = LayerNorm(C).cuda().bfloat16()
    x = torch.randn(B, T, C, device='cuda').bfloat16().requires_grad_(True)
    sc = torch.randn(B, T, C, device='cuda').bfloat16().requires_grad_(True)
    scale = torch.tensor([1.25, 0.0, 1.25], device='cuda')
    cot = torch.randn(B, T, C, device='cuda').bfloat16()

    def run(fused):
        for p in list(mod.parameters()) + list(norm.parameters()) + [x, sc]:
            p.grad = None
        if with_norm:
            if fused:
                out, y = M.mlp_add_layer_norm(x, mod, sc, scale, norm)
            else:
                out = torch.addcmul(sc, mod(x), scale.view(-1, 1, 1).bfloat16())
                y = norm(out)
            (out * cot + y * cot.flip(0)).sum().backward()
            res = [out, y]
        else:
            out = M.mlp(x, mod) if fused else mod(x)
            (out * cot).sum().backward()
            res = [out]
        return [r.detach().float() for r in res], [p.grad.float().clone() for p in list(mod.parameters()) + [x]]

    outs_f, grads_f = run(True)
    outs_r, grads_r = run(False)
    for a, b in zip(outs_f, outs_r):
        _close(a, b, 2.0 ** -6)
    for a, b in zip(grads_f, grads_r):
        _close(a, b, 2.0 ** -5)
