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


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 10, 11, 12, 13])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1000, 512, 128), (4096 + 17, 256, 512), (37, 1024, 192), (70000, 768, 96)])
def test_gemm_epilogues(M, N, K, variant):
    """(variants 10-13, round 6: the same kernel template on 64 x 128 / 128 x 128 / 64 x 64 tiles for the decoders' short maps -- two or
    three workgroups per CU; first instantiated with fewer token blocks per wave than DMA pieces per stage: the pieces past the
    block loop were never issued -- every epilogue of every tile shape runs here.)"""
    from grit_amd.ops import gemm as G
    from grit_amd.ops.linear import slab_sum
    if variant in (2, 4, 10, 11, 12) and K % 64:
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
    if variant >= 10:
        return  # (the short-map tiles have no GELU' epilogue: its column sums are laid out per 128-row wave block)
    # GELU' epilogue + column sums (rows past M must not leak into the sums)
    aux = torch.randn(M, N, device='cuda').bfloat16()
    # the partials sit between two canary rows: a wave whose rows all lie past M must not write a slab (round 3: a 37-row
    # problem wrote 4 KB of zeros behind its one slab, into whatever the allocator had placed there)
    slabs = -(-M // 128)
    fenced = torch.full((slabs + 2, N), float('nan'), device='cuda')
    part = fenced[1:1 + slabs]
    d = G.gemm_nt(x, w, G.DGELU, aux=aux, colsum=part, variant=variant)
    assert torch.isnan(fenced[0]).all() and torch.isnan(fenced[-1]).all()
    a32 = aux.float().requires_grad_(True)
    F.gelu(a32).backward(ref)
    _close(d, a32.grad)
    db = slab_sum(part.unsqueeze(0), torch.float32)[0]
    assert torch.isfinite(part).all()
    ref_db = a32.grad.sum(0)
    assert (db - ref_db).abs().max().item() <= 2e-3 * ref_db.abs().max().item() + 1e-3


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 256, 128), (4096 + 17, 512, 512), (70000, 768, 192), (51200, 1536, 512),
                                   (12800, 1024, 1024)])
def test_gemm_four_wave_persistent_kernel(M, N, K):
    """Variant 7 (grit_amd/csrc/gemm_w4.hip: four waves, 128 x 128 wave tiles, persistent stream of K steps over a workgroup's
    tiles): every epilogue against the float32 contraction of the same bf16 inputs; bias epilogue bit for bit equal to the
    per-tile kernel (same products, same k order per accumulator); the last row tile is shifted back to end at row M (odd M), its
    shared rows must not be counted twice in the GELU' column sums; nothing written outside the [2 ceil(M / 256), N] partials."""
    from grit_amd.ops import gemm as G
    x, w, b = _inputs(M, N, K, seed=M % 97)
    ref = x.float() @ w.float().t()
    _close(G.gemm_nt(x, w, G.NONE, variant=7), ref)
    own = G.gemm_nt(x, w, G.BIAS, bias=b, variant=7)
    _close(own, ref + b.float())
    if K % 64 == 0:
        assert torch.equal(own, G.gemm_nt(x, w, G.BIAS, bias=b, variant=4))
    pre = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
    act = G.gemm_nt(x, w, G.BIAS_GELU, bias=b, aux=pre, variant=7)
    _close(pre, ref + b.float())
    _close(act, F.gelu(ref + b.float()))
    assert torch.equal(act, G.gemm_nt(x, w, G.BIAS_GELU, bias=b, variant=7))  # no pre-activation kept: same activation
    aux = torch.randn(M, N, device='cuda').bfloat16()
    slabs = 2 * (-(-M // 256))
    fenced = torch.full((slabs + 2, N), float('nan'), device='cuda')
    part = fenced[1:1 + slabs]
    part.zero_()
    d = G.gemm_nt(x, w, G.DGELU, aux=aux, colsum=part, variant=7)
    assert torch.isnan(fenced[0]).all() and torch.isnan(fenced[-1]).all()
    a32 = aux.float().requires_grad_(True)
    F.gelu(a32).backward(ref)
    _close(d, a32.grad)
    db = part.sum(0)
    ref_db = a32.grad.sum(0)
    assert (db - ref_db).abs().max().item() <= 4e-3 * ref_db.abs().max().item() + 1e-2, (db - ref_db).abs().max().item()


@pytest.mark.parametrize("M,N,K", [(51200, 512, 2048), (12800, 1024, 1024), (204800, 256, 768), (4096 + 17, 512, 512), (300, 256, 128),
                                   (12800, 4096, 1024)])
def test_gemm_four_wave_224_row_tiles(M, N, K):
    """Variant 9 = variant 7 with the tile height chosen by shape (gemm_w4.hip, MI = 7: 224-row tiles where they need fewer block rows
    per CU).  Same products in the same k order per accumulator: EVERY output of every epilogue bit for bit equal to variant 7; the
    GELU' column sums (2 rows per tile row, hence a different partition of the rows) equal after the sum; nothing written outside them."""
    from grit_amd.ops import gemm as G
    x, w, b = _inputs(M, N, K, seed=(M + N) % 83)
    rows = G.w4_tile_rows(M, N)
    assert rows in (224, 256)
    if (M, N) in ((51200, 512), (12800, 1024), (204800, 256)):
        assert rows == 224  # the N = C products of the three trainable Swin stages on a 256-CU device
    assert torch.equal(G.gemm_nt(x, w, G.NONE, variant=9), G.gemm_nt(x, w, G.NONE, variant=7))
    assert torch.equal(G.gemm_nt(x, w, G.BIAS, bias=b, variant=9), G.gemm_nt(x, w, G.BIAS, bias=b, variant=7))
    pre7 = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
    pre9 = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
    act7 = G.gemm_nt(x, w, G.BIAS_GELU, bias=b, aux=pre7, variant=7)
    act9 = G.gemm_nt(x, w, G.BIAS_GELU, bias=b, aux=pre9, variant=9)
    assert torch.equal(act7, act9) and torch.equal(pre7, pre9)
    assert torch.equal(act7, G.gemm_nt(x, w, G.BIAS_GELU, bias=b, variant=9))  # (no pre-activation kept)
    aux = torch.randn(M, N, device='cuda').bfloat16()
    part7 = torch.zeros((2 * (-(-M // 256)), N), device='cuda')
    slabs = 2 * (-(-M // rows))
    fenced = torch.full((slabs + 2, N), float('nan'), device='cuda')
    part9 = fenced[1:1 + slabs]
    part9.zero_()
    d7 = G.gemm_nt(x, w, G.DGELU, aux=aux, colsum=part7, variant=7)
    d9 = G.gemm_nt(x, w, G.DGELU, aux=aux, colsum=part9, variant=9)
    assert torch.equal(d7, d9)
    assert torch.isnan(fenced[0]).all() and torch.isnan(fenced[-1]).all() and torch.isfinite(part9).all()
    s7, s9 = part7.sum(0), part9.sum(0)
    assert float((s7 - s9).abs().max()) <= 1e-4 * float(s7.abs().max()) + 1e-3


@pytest.mark.parametrize("M,N,K,per", [(51200, 512, 512, 1600), (51200, 512, 2048, 1600), (12800, 1024, 4096, 400), (204800, 256, 256, 6400),
                                       (4096 + 17, 512, 512, 0), (2560, 256, 128, 256),
                                       (102400, 128, 128, 25600), (102400, 128, 512, 25600), (1000 + 7, 128, 96, 0), (5120, 128, 64, 512)])
def test_gemm_residual_epilogue(M, N, K, per):
    """grit_gemm_bf16_nt_res: x = shortcut + factor[sample] * (inp W^T + b) in one launch, against the two-launch form it replaces --
    the branch stored in bf16 by variant 7, then grit_add_layernorm_fwd's sum -- bit for bit (same roundings in the same order), with
    drop-path factors (zeros included: x == shortcut there, whatever the branch) and without; and against the float32 arithmetic."""
    from grit_amd.ops import gemm as G
    from grit_amd.ops.layer_norm import add_layer_norm
    x, w, b = _inputs(M, N, K, seed=(M + K) % 71)
    g = torch.Generator(device='cuda').manual_seed(5)
    res = torch.randn(M, N, device='cuda', generator=g).bfloat16()
    # (N = 128 / 384, the stage-0 map's narrow outputs: the per-tile kernel's 256 x 128 tiles carry the epilogue there)
    branch = G.gemm_nt(x, w, G.BIAS, bias=b, variant=7 if N % 256 == 0 and K % 64 == 0 else 1)
    if per:
        B = M // per
        scale = torch.full((B,), 1.0 / 0.9, device='cuda')
        scale[torch.arange(B, device='cuda') % 5 == 1] = 0.0
    else:
        B, scale = 1, None
    got = G.gemm_nt_residual(x, w, b, res, scale, per)
    ln_w, ln_b = torch.ones(N, device='cuda').bfloat16(), torch.zeros(N, device='cuda').bfloat16()
    want, _ = add_layer_norm(res.view(B, M // B, N), branch.view(B, M // B, N), scale, ln_w, ln_b)
    assert torch.equal(got, want.reshape(M, N))
    ref = res.float() + (1.0 if scale is None else scale.repeat_interleave(per)[:, None]) * (x.float() @ w.float().t() + b.float())
    _close(got, ref)
    if scale is not None:
        dropped = (scale == 0).repeat_interleave(per)
        assert torch.equal(got[dropped], res[dropped])


def test_residual_epilogue_inside_the_block_tail_nodes(monkeypatch):
    """_LinearAddLayerNormFn (attn.proj + residual + norm2) and _MlpAddLayerNormFn (Mlp + residual + next norm1) with the residual
    connection in the GEMM epilogue (GRIT_GEMM_RESIDUAL, default) against the same nodes with the stored branch + grit_add_layernorm_fwd:
    x, LayerNorm(x) and every gradient bit for bit (the epilogue reproduces the add kernel's roundings; the backward is the same code)."""
    from grit_amd.ops import gemm as G
    from grit_amd.ops import transposed
    from grit_amd.ops.layer_norm import LayerNorm, linear_add_layer_norm
    from grit_amd.ops.linear import Linear
    from grit_amd.ops.mlp import mlp_add_layer_norm
    torch.manual_seed(3)
    B, T, C = 8, 1600, 512
    proj = Linear(C, C).cuda().bfloat16()
    norm = LayerNorm(C).cuda().bfloat16()
    with torch.no_grad():
        norm.weight.normal_(1.0, 0.2)
        norm.bias.normal_(0.0, 0.2)

    class Mlp(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1, self.act, self.fc2, self.drop = Linear(C, 4 * C), torch.nn.GELU(), Linear(4 * C, C), torch.nn.Dropout(0.0)
    mlp = Mlp().cuda().bfloat16()
    inp = torch.randn(B, T, C, device='cuda').bfloat16()
    shortcut = torch.randn(B, T, C, device='cuda').bfloat16()
    scale = torch.full((B,), 1.0 / 0.85, device='cuda')
    scale[2] = 0.0
    cot = [torch.randn(B, T, C, device='cuda').bfloat16() for _ in range(2)]
    params = list(proj.parameters()) + list(norm.parameters()) + list(mlp.parameters())
    results = []
    for fused in (True, False):
        monkeypatch.setattr(G, "RESIDUAL", fused)
        transposed.refresh([proj.weight, mlp.fc1.weight, mlp.fc2.weight])
        out = []
        for which in ("proj", "mlp"):
            for p in params:
                p.grad = None
            a, s_ = inp.clone().requires_grad_(True), shortcut.clone().requires_grad_(True)
            if which == "proj":
                x, y = linear_add_layer_norm(a, proj, s_, scale, norm.weight, norm.bias, norm.eps)
            else:
                x, y = mlp_add_layer_norm(a, mlp, s_, scale, norm)
            (x * cot[0] * scale.view(-1, 1, 1).to(x.dtype) + y * cot[1]).sum().backward()
            out += [x.detach(), y.detach(), a.grad, s_.grad] + [p.grad.clone() for p in params if p.grad is not None]
        results.append(out)
    assert len(results[0]) == len(results[1])
    for i, (f, u) in enumerate(zip(*results)):
        assert torch.equal(f, u), i
    # without autograd (inference, the frozen stage 0 of a training step) the same epilogue replaces linear + grit_add_layernorm_fwd
    nograd = []
    for fused in (True, False):
        monkeypatch.setattr(G, "RESIDUAL", fused)
        with torch.no_grad():
            nograd.append([t for sc in (None, scale) for t in linear_add_layer_norm(inp, proj, shortcut, sc, norm.weight, norm.bias, norm.eps)])
    for i, (f, u) in enumerate(zip(*nograd)):
        assert torch.equal(f, u), ("no_grad", i)
    assert torch.equal(nograd[0][2], results[0][0]) and torch.equal(nograd[0][3], results[0][1])  # == the training node's x, LayerNorm(x)


def test_own_long_map_policy_and_linear_nodes():
    """grit_amd.ops.gemm.prefers_own / long_linear / long_input_grad and the Linear node on top: the own kernel takes the long-map
    shapes the policy names (values and gradients equal to the library path within bf16 rounding), everything else runs the library."""
    from grit_amd.ops import gemm as G
    from grit_amd.ops import transposed
    from grit_amd.ops.linear import linear
    assert G.prefers_own(51200, 1536, 512) and G.prefers_own(51200, 512, 512) and G.prefers_own(12800, 3072, 1024)
    assert G.prefers_own(51200, 512, 2048) and G.prefers_own(12800, 1024, 4096) and G.prefers_own(204800, 256, 1024)  # (round 6: 224-row tiles)
    assert not G.prefers_own(4800, 512, 512) and not G.prefers_own(51200, 384, 512)
    assert G.prefers_own(272000, 3072, 512)  # 12 756 tiles: level with the library inside the step (profiles/r06/ab_own_max_tiles.txt)
    torch.manual_seed(0)
    M, N, K = 16384, 512, 256
    x = torch.randn(M, K, device='cuda').bfloat16().requires_grad_(True)
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16().requires_grad_(True)
    b = torch.randn(N, device='cuda').bfloat16().requires_grad_(True)
    cot = torch.randn(M, N, device='cuda').bfloat16()
    assert G.long_linear(x.detach(), w.detach(), b.detach()) is not None
    transposed.refresh([w])
    assert G.long_input_grad(cot, w) is not None
    y = linear(x, w, b)
    (y * cot).sum().backward()
    got = [y.detach().float(), x.grad.float().clone(), w.grad.float().clone(), b.grad.float().clone()]
    for t in (x, w, b):
        t.grad = None
    yr = F.linear(x, w, b)
    (yr * cot).sum().backward()
    ref = [yr.detach().float(), x.grad.float(), w.grad.float(), b.grad.float()]
    for g_, r_ in zip(got, ref):
        _close(g_, r_)


def test_short_map_policy_linear_and_packed_in_projection_nodes(monkeypatch):
    """grit_amd.ops.gemm.prefers_own_short (round 6): the Linears of the decoders' 640 .. 4 800-row maps run the 64 x 64 x 64 tiles (variant 12)
    in the forward and -- on the transposed copies -- in the input gradient; the Linear node and the packed in-projection node with the
    policy on against the same nodes on the library (GRIT_GEMM_OWN_SHORT=0) and against float32: values and every gradient."""
    from grit_amd.ops import gemm as G
    from grit_amd.ops import transposed
    from grit_amd.ops.linear import linear, packed_in_proj
    assert G.prefers_own_short(4800, 512, 512) and G.prefers_own_short(640, 2048, 512) and G.prefers_own_short(4800, 512, 1024)
    assert not G.prefers_own_short(640, 512, 1024) and not G.prefers_own_short(640, 512, 2048) and not G.prefers_own_short(4800, 4, 512)
    assert not G.prefers_own_short(9600, 512, 512) and not G.prefers_own_short(32, 512, 512)  # long-map policy / a handful of rows: not these tiles
    assert G.prefers_own_short(320, 512, 512)  # the beam-search steps' 64 .. 320 rows (profiles/r06/decode_short_min_rows.txt)
    torch.manual_seed(1)
    for M, N, K in ((4800, 512, 512), (640, 2048, 512), (4800 + 37, 256, 1024)):
        x = torch.randn(M, K, device='cuda').bfloat16().requires_grad_(True)
        w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16().requires_grad_(True)
        b = torch.randn(N, device='cuda').bfloat16().requires_grad_(True)
        cot = torch.randn(M, N, device='cuda').bfloat16()
        out = []
        for on in (True, False):
            monkeypatch.setattr(G, "OWN_SHORT", on)
            transposed.refresh([w])
            assert (G.long_linear(x.detach(), w.detach(), b.detach()) is not None) == on
            assert (G.long_input_grad(cot, w) is not None) == (on and G.prefers_own_short(M, K, N))  # (as an NT product: N and K swap)
            for t in (x, w, b):
                t.grad = None
            y = linear(x, w, b)
            (y * cot).sum().backward()
            out.append([y.detach().float(), x.grad.float().clone(), w.grad.float().clone(), b.grad.float().clone()])
        x32, w32, b32 = (t.detach().float().requires_grad_(True) for t in (x, w, b))
        y32 = F.linear(x32, w32, b32)
        (y32 * cot.float()).sum().backward()
        for g_, l_, r_ in zip(out[0], out[1], [y32.detach(), x32.grad, w32.grad, b32.grad]):
            _close(g_, r_)
            _close(g_, l_)
    # the packed in-projection: q = k from one input, v from another, two row ranges of one [3E, E] parameter
    E, B, Lq = 256, 32, 150
    qk_in = torch.randn(B, Lq, E, device='cuda').bfloat16().requires_grad_(True)
    v_in = torch.randn(B, Lq, E, device='cuda').bfloat16().requires_grad_(True)
    w = (torch.randn(3 * E, E, device='cuda') * E ** -0.5).bfloat16().requires_grad_(True)
    b = torch.randn(3 * E, device='cuda').bfloat16().requires_grad_(True)
    cots = torch.randn(B, Lq, 2 * E, device='cuda').bfloat16(), torch.randn(B, Lq, E, device='cuda').bfloat16()
    out = []
    for on in (True, False):
        monkeypatch.setattr(G, "OWN_SHORT", on)
        transposed.refresh([w])
        for t in (qk_in, v_in, w, b):
            t.grad = None
        qk, v = packed_in_proj(qk_in, v_in, w, b)
        ((qk * cots[0]).sum() + (v * cots[1]).sum()).backward()
        out.append([qk.detach().float(), v.detach().float()] + [t.grad.float().clone() for t in (qk_in, v_in, w, b)])
    for g_, l_ in zip(*out):
        _close(g_, l_)
    # (the two paths can agree to the bit -- same MFMA, same k order per accumulator -- so the choice of path is checked directly)
    from grit_amd.ops.linear import _own_linear, _short_transposed
    monkeypatch.setattr(G, "OWN_SHORT", True)
    transposed.refresh([w])
    assert _short_transposed(w, cots[0].reshape(-1, 2 * E)) is not None and _own_linear(qk_in.detach(), w[:2 * E].detach(), b[:2 * E].detach()) is not None
    monkeypatch.setattr(G, "OWN_SHORT", False)
    assert _short_transposed(w, cots[0].reshape(-1, 2 * E)) is None and _own_linear(qk_in.detach(), w[:2 * E].detach(), b[:2 * E].detach()) is None


@pytest.mark.gpu
def test_narrow_output_policy_for_the_stage0_maps():
    """grit_amd.ops.gemm.prefers_own_narrow: the 128- / 384-column products of the 819 200-token stage-0 map run the eight-wave own
    kernel (variant 0) in the forward and -- on the transposed weight -- in the input gradient; values against fp32 torch."""
    from grit_amd.ops import gemm as G
    from grit_amd.ops import transposed
    assert G.prefers_own_narrow(819200, 384, 128) and G.prefers_own_narrow(819200, 128, 512) and G.prefers_own_narrow(819200, 128, 384)
    assert not G.prefers_own_narrow(204800, 128, 128) and not G.prefers_own_narrow(819200, 256, 128) and not G.prefers_own_narrow(819200, 128, 1024)
    torch.manual_seed(3)
    M = 262144 + 64
    for N, K in ((384, 128), (128, 512)):
        x = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
        b = torch.randn(N, device='cuda').bfloat16()
        y = G.long_linear(x, w, b)
        assert y is not None
        rows = torch.tensor([0, 255, 256, 99999, M - 1], device='cuda')
        _close(y[rows], x[rows].float() @ w.float().t() + b.float())
        cot = torch.randn(M, N, device='cuda').bfloat16()
        transposed.refresh([w])
        dx = G.long_input_grad(cot, w)
        assert dx is not None and dx.shape == (M, K)
        _close(dx[rows], cot[rows].float() @ w.float())


@pytest.mark.gpu
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


def test_fused_mlp_node_with_and_without_the_dropped_sample_shortcut(monkeypatch):
    """The Mlp + residual + LayerNorm node with drop-path factors: not computing the fc1 tiles of dropped samples (forward) and the
    GELU' tiles (backward) leaves every output and every gradient of the node bit-identical (GRIT_GEMM_ROW_SKIP on / off)."""
    from grit_amd.models.common.swin_model import Mlp
    from grit_amd.ops.layer_norm import LayerNorm
    from grit_amd.ops import gemm as G
    from grit_amd.ops import mlp as M
    torch.manual_seed(1)
    B, T, C = 4, 1600, 256
    mod = Mlp(C, 4 * C).cuda().bfloat16()
    norm = LayerNorm(C).cuda().bfloat16()
    x = torch.randn(B, T, C, device='cuda').bfloat16().requires_grad_(True)
    sc = torch.randn(B, T, C, device='cuda').bfloat16().requires_grad_(True)
    scale = torch.tensor([0.0, 1.25, 0.0, 1.25], device='cuda')
    cot = torch.randn(B, T, C, device='cuda').bfloat16()

    def run(skip):
        monkeypatch.setattr(G, "ROW_SKIP", skip)
        for p in list(mod.parameters()) + list(norm.parameters()) + [x, sc]:
            p.grad = None
        out, y = M.mlp_add_layer_norm(x, mod, sc, scale, norm)
        (out * cot + y * cot.flip(0)).sum().backward()
        return [out.detach().clone(), y.detach().clone()] + [p.grad.clone() for p in list(mod.parameters()) + list(norm.parameters()) + [x, sc]]
    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("M,N,K", [(4800, 512, 512), (4800, 128, 512), (4800, 1024, 512), (640, 512, 512), (3200, 2048, 512),
                                   (777, 64, 192), (33, 128, 64), (16000, 256, 1024), (4801, 64, 64)])
@pytest.mark.parametrize("strided", [False, True])
def test_small_map_weight_and_bias_gradient(M, N, K, strided, monkeypatch):
    """grit_wgrad_small (dW = dY^T X and db = colsum(dY) of a short map in one launch: finished bf16 gradients up to 4 800 rows,
    f32 split partials + the grouped reduction beyond), against the same contraction in float64 on the bf16 inputs; ragged M
    (tail rows zero-filled), operands that are column slices of wider tensors (leading dimension > width), canary rows around
    both outputs."""
    import ctypes
    from grit_amd import lib as _lib
    from grit_amd.ops import linear as L
    from grit_amd.ops.linear import SlabGroup, small_weight_bias_grad
    monkeypatch.setattr(L, "WGRAD_SMALL", True)  # opt-in path (GRIT_WGRAD_SMALL=1): slower inside the step, see linear.py
    g = torch.Generator(device='cuda').manual_seed(M + N)
    if strided:
        dy = torch.randn(M, N + 64, device='cuda', generator=g).bfloat16()[:, 32:32 + N]
        x = torch.randn(M, K + 8, device='cuda', generator=g).bfloat16()[:, 8:]
    else:
        dy = torch.randn(M, N, device='cuda', generator=g).bfloat16()
        x = torch.randn(M, K, device='cuda', generator=g).bfloat16()
    ref_w = (dy.double().t() @ x.double()).float()
    ref_b = dy.double().sum(0).float()
    dw, db = small_weight_bias_grad(dy, x, True, torch.bfloat16)
    assert dw.dtype == db.dtype == torch.bfloat16 and dw.shape == (N, K) and db.shape == (N,)
    _close(dw, ref_w)
    _close(db, ref_b)
    group = SlabGroup()  # through a caller's group, weight gradient alone
    dw2, none = small_weight_bias_grad(dy, x, False, torch.bfloat16, group)
    group.run()
    assert none is None and torch.equal(dw2, dw)
    # the raw entry point between canaries: nothing is written outside the outputs
    lib = _lib.load()
    S = lib.grit_wgrad_small_splits(M, N, K)
    assert S >= 1
    wdt = torch.bfloat16 if S == 1 else torch.float32
    wfence = torch.full((S + 2, N, K), float('nan'), device='cuda', dtype=wdt)
    bfence = torch.full((S + 2, N), float('nan'), device='cuda', dtype=wdt)
    st = lib.grit_wgrad_small(ctypes.c_void_p(dy.data_ptr()), dy.stride(0), ctypes.c_void_p(x.data_ptr()), x.stride(0), M, N, K, S,
                              ctypes.c_void_p(wfence[1].data_ptr()), ctypes.c_void_p(bfence[1].data_ptr()), _lib.current_stream_ptr())
    assert st == 0
    assert torch.isnan(wfence[0]).all() and torch.isnan(wfence[-1]).all() and torch.isfinite(wfence[1:-1]).all()
    assert torch.isnan(bfence[0]).all() and torch.isnan(bfence[-1]).all() and torch.isfinite(bfence[1:-1]).all()
    if S > 1:  # f32 partials: their sum is the contraction to fp32 accuracy
        scale = ref_w.abs().max().item()
        assert (wfence[1:-1].double().sum(0) - ref_w.double()).abs().max().item() <= 1e-5 * scale + 1e-4
        assert (bfence[1:-1].double().sum(0) - ref_b.double()).abs().max().item() <= 1e-5 * ref_b.abs().max().item() + 1e-4
    else:
        assert torch.equal(wfence[1], dw) and torch.equal(bfence[1], db)


def test_small_map_gradient_rejects_what_it_does_not_cover(monkeypatch):
    from grit_amd import lib as _lib
    from grit_amd.ops import linear as L
    monkeypatch.setattr(L, "WGRAD_SMALL", True)
    lib = _lib.load()
    assert lib.grit_wgrad_small_splits(4800, 10201, 512) == 0  # vocabulary projection: N % 64 != 0 -> the library GEMM
    assert lib.grit_wgrad_small_splits(4800, 512, 100) == 0
    from grit_amd.ops.linear import small_weight_bias_grad
    dy = torch.randn(100, 10201, device='cuda').bfloat16()
    x = torch.randn(100, 512, device='cuda').bfloat16()
    assert small_weight_bias_grad(dy, x, True, torch.bfloat16) is None
    dy = torch.randn(100, 128, device='cuda')  # fp32 gradients: the parity path stays on the library
    assert small_weight_bias_grad(dy, x, True, torch.float32) is None


@pytest.mark.parametrize("M,per,N,K", [(51200, 1600, 2048, 512), (4096 + 300, 550, 256, 128), (12800, 400, 1024, 256)])
def test_dropped_samples_are_skipped_not_computed(M, per, N, K):
    """grit_gemm_bf16_nt_rows: with the per-sample drop-path factors at hand, the GELU' GEMM writes the tiles of dropped samples as
    zeros without a K loop.  Bit-identical to the dense kernel on the same (zeroed) rows -- result and column sums -- whatever the
    alignment of samples and 256-row tiles; a NaN planted in the pre-activation of a skipped tile is never read."""
    from grit_amd.ops import gemm as G
    torch.manual_seed(M + N)
    B = -(-M // per)
    scale = (torch.rand(B, device='cuda') > 0.4).float() / 0.6
    scale[0] = 0.0
    dy = torch.randn(M, K, device='cuda').bfloat16()
    rows = torch.arange(M, device='cuda') // per
    dy = (dy.float() * scale[rows].unsqueeze(1)).bfloat16()          # dbranch = scale * dx: exact zero rows for dropped samples
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
    pre = torch.randn(M, N, device='cuda').bfloat16()
    dense, part_dense = G.input_grad_dgelu(dy, w, pre)
    pre_nan = pre.clone()
    if per >= 512:
        pre_nan[256:512] = float('nan')  # rows of sample 0 (dropped), a tile that lies inside it
    skip, part_skip = G.input_grad_dgelu(dy, w, pre_nan, scale, per)
    assert torch.equal(skip, dense)
    assert torch.equal(part_skip, part_dense)
    assert float(skip[:per].abs().max()) == 0.0


def test_fused_gelu_epilogues_against_erf_gelu_over_the_whole_range():
    """the BIAS_GELU / DGELU epilogues evaluate GELU as x * sigmoid(x * P(x^2)) fitted to the reference's exact erf
    form (nn.GELU, models/common/swin_model.py:19-37).  Bound the deviation where it can be seen: an identity GEMM (K = N = 128,
    weight = I) feeds every bf16 value of [-9, 9] through both epilogues; forward against torch's erf GELU and backward against
    its autograd derivative, both evaluated in fp32 on the same bf16 inputs.  Tolerance: the fit's stated 2.6e-5 / 1.1e-4 absolute
    plus the bf16 rounding of the stored result (2^-9 relative)."""
    from grit_amd.ops import gemm as G
    N = 128
    vals = torch.linspace(-9, 9, 256 * 1024, device='cuda').bfloat16().unique()
    M = (vals.numel() + N - 1) // N * N
    x = torch.zeros(M, device='cuda', dtype=torch.bfloat16)
    x[:vals.numel()] = vals
    x = x.view(-1, N).contiguous()
    eye = torch.eye(N, device='cuda', dtype=torch.bfloat16)
    zero_b = torch.zeros(N, device='cuda', dtype=torch.bfloat16)
    pre = torch.empty_like(x)
    act = G.gemm_nt(x, eye, G.BIAS_GELU, bias=zero_b, aux=pre)
    assert torch.equal(pre, x)
    ref = F.gelu(x.float())
    err = (act.float() - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -8).max()) <= 3e-5, float(err.max())
    # derivative: d = (1 . I) * gelu'(x) with an all-ones accumulator operand
    ones = torch.ones_like(x)
    part = torch.empty((-(-x.shape[0] // 128), N), device='cuda')
    d = G.gemm_nt(ones, eye, G.DGELU, aux=x, colsum=part)
    xr = x.float().requires_grad_(True)
    F.gelu(xr).sum().backward()
    derr = (d.float() - xr.grad).abs()
    assert float((derr - xr.grad.abs() * 2.0 ** -8).max()) <= 1.5e-4, float(derr.max())


DEV = "cuda"


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,pad", [(51200, 2048, 512, 0), (51200, 512, 512, 0), (51200, 1536, 512, 0), (12800, 1024, 4096, 0),
                                       (2464 * 3 + 32, 256, 256, 8), (64, 256, 512, 0), (204800, 768, 256, 0),
                                       (51232, 512, 512, 0), (96, 256, 256, 0), (128, 512, 256, 8), (204800, 256, 1024, 0)])
def test_long_map_weight_gradient_tn(M, N, K, pad):
    """grit_wgrad_tn: dW = dY^T X over row slices (fp32 partials) on the Swin shapes -- the four-wave kernel (rows a multiple of
    64: 64-row steps) and the eight-wave one (M % 64 == 32: 32-row steps), even and odd numbers of steps per slice, one and two
    steps in all, a last slice shorter than the others, fewer workgroups than CUs, operands with a leading dimension larger than
    their width -- against an fp64 reference of the same bf16 inputs; shapes outside the contract answer 0 slices."""
    import ctypes
    from grit_amd import lib as _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    dy_full = torch.randn(M, N + pad, device=DEV, generator=g).bfloat16()
    x_full = torch.randn(M, K + pad, device=DEV, generator=g).bfloat16()
    dy, x = dy_full[:, :N], x_full[:, :K]
    S = lib.grit_wgrad_tn_splits(M, N, K)
    assert S >= 1 and (N // 256) * (K // 256) * S <= 256
    part = torch.full((S, N, K), float("nan"), dtype=torch.float32, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    bpart = torch.full((S, N), float("nan"), dtype=torch.float32, device=DEV)
    st = lib.grit_wgrad_tn(p(dy), dy.stride(0), p(x), x.stride(0), M, N, K, S, p(part), p(bpart), _lib.current_stream_ptr())
    assert st == 0
    refb = dy.double().sum(0)  # the bias gradient as a by-product of the same launch
    assert float((bpart.double().sum(0) - refb).abs().max()) <= 1e-5 * float(refb.abs().max()) + 1e-3
    got = part.double().sum(0)
    ref = dy.double().t() @ x.double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale * max(1.0, M / 51200) + 1e-4
    assert lib.grit_wgrad_tn_splits(51200, 384, 128) == 0 and lib.grit_wgrad_tn_splits(51201, 512, 512) == 0
    assert lib.grit_wgrad_tn(p(dy), dy.stride(0), p(x), x.stride(0), M, N, K, S + 1, p(part), None, _lib.current_stream_ptr()) != 0


@pytest.mark.gpu
def test_weight_grad_routes_long_maps_to_the_own_kernel(monkeypatch):
    """ops.linear.weight_grad: the long-map path (own kernel + grouped slab sum) and the library path (GRIT_WGRAD_TN=0) agree
    within the bf16 rounding of the result."""
    from grit_amd.ops import linear as L
    g = torch.Generator(device=DEV).manual_seed(5)
    dy = torch.randn(51200, 512, device=DEV, generator=g).bfloat16()
    x = torch.randn(51200, 512, device=DEV, generator=g).bfloat16()
    monkeypatch.setattr(L, "WGRAD_TN", True)
    assert L.long_weight_grad_partials(dy, x) is not None
    own = L.weight_grad(dy, x)
    monkeypatch.setattr(L, "WGRAD_TN", False)
    assert L.long_weight_grad_partials(dy, x) is None
    lib_path = L.weight_grad(dy, x)
    assert own.dtype == torch.bfloat16 and own.shape == (512, 512)
    torch.testing.assert_close(own.float(), lib_path.float(), rtol=1e-2, atol=1e-2 * float(lib_path.float().abs().max()))


@pytest.mark.gpu
def test_grouped_long_kernel_and_grouped_column_sums():
    """grit_wgrad_tn_grouped: several short-map problems (different shapes, row counts, slice counts, one with padded leading
    dimensions) in one launch of the 256 x 256-tile kernel; grit_colsum_grouped: their column sums in one launch.  Against fp64."""
    import ctypes
    from grit_amd import lib as _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(11)
    problems = [(4800, 512, 512, 2, 0), (4800, 1024, 512, 3, 0), (640, 512, 2048, 1, 0), (4800, 256, 512, 2, 8), (3200, 512, 256, 5, 0)]
    assert all(lib.grit_wgrad_tn_group_ok(M, N, K) for M, N, K, _, _ in problems) and not lib.grit_wgrad_tn_group_ok(4800, 128, 512)
    table = (_lib.WgradJob * len(problems))()
    ctable = (_lib.ColsumJob * len(problems))()
    keep, keep_b = [], []
    for t, (M, N, K, S, pad) in enumerate(problems):
        dy = torch.randn(M, N + pad, device=DEV, generator=g).bfloat16()[:, :N]
        x = torch.randn(M, K + pad, device=DEV, generator=g).bfloat16()[:, :K]
        part = torch.full((S, N, K), float("nan"), dtype=torch.float32, device=DEV)
        slabs = max(1, M // 64)
        bpart = torch.full((slabs, N), float("nan"), dtype=torch.float32, device=DEV)
        bslice = torch.full((S, N), float("nan"), dtype=torch.float32, device=DEV)
        table[t] = _lib.WgradJob(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, N, K, S, part.data_ptr(),
                                 bslice.data_ptr() if t % 2 == 0 else None)
        keep_b.append(bslice if t % 2 == 0 else None)
        ctable[t] = _lib.ColsumJob(dy.data_ptr(), dy.stride(0), M, N, slabs, bpart.data_ptr())
        keep.append((dy, x, part, bpart))
    assert lib.grit_wgrad_tn_grouped(table, len(problems), _lib.current_stream_ptr()) == 0
    assert lib.grit_colsum_grouped(ctable, len(problems), _lib.current_stream_ptr()) == 0
    torch.cuda.synchronize()
    for (dy, x, part, bpart), bslice in zip(keep, keep_b):
        if bslice is not None:  # the by-product of the GEMM launch itself
            refb = dy.double().sum(0)
            assert float((bslice.double().sum(0) - refb).abs().max()) <= 1e-5 * float(refb.abs().max()) + 1e-3
        ref = dy.double().t() @ x.double()
        assert float((part.double().sum(0) - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-4
        refb = dy.double().sum(0)
        assert float((bpart.double().sum(0) - refb).abs().max()) <= 1e-5 * float(refb.abs().max()) + 1e-4
    # a slice count that would leave a slice without a step is refused
    table[2] = _lib.WgradJob(keep[2][0].data_ptr(), keep[2][0].stride(0), keep[2][1].data_ptr(), keep[2][1].stride(0), 640, 512, 2048, 21,
                             keep[2][2].data_ptr(), None)
    assert lib.grit_wgrad_tn_grouped(table, len(problems), _lib.current_stream_ptr()) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,per,N,K,drop", [(32, 1600, 512, 2048, (0, 3, 4, 17, 31)), (32, 1600, 2048, 512, (5,)), (8, 6400, 256, 1024, (1, 2, 6)),
                                            (32, 1600, 512, 512, tuple(range(1, 32))), (32, 1600, 512, 512, ()), (16, 400, 1024, 1024, (2, 9))])
def test_weight_gradient_skips_the_rows_of_dropped_samples(B, per, N, K, drop):
    """grit_wgrad_tn_rows / the row_scale fields of grit_wgrad_tn_grouped (round 5): with the drop-path factors of the branch at hand the
    64-row steps inside dropped samples (exact zero rows of dY) are never loaded -- NaNs planted in X there are not seen -- and the slices
    share the live steps: the SUM of the slice partials equals the plain kernel's up to fp32 summation order, the bias by-product too.
    Nothing dropped: the very same slices, bit for bit.  rows_per_sample % 64 != 0 (stage 3: 400 rows): every row is processed."""
    import ctypes
    from grit_amd import lib as _lib
    lib = _lib.load()
    M = B * per
    g = torch.Generator(device=DEV).manual_seed(B + N + len(drop))
    dy = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    scale = torch.full((B,), 1.25, device=DEV)
    for b in drop:
        scale[b] = 0.0
        dy[b * per:(b + 1) * per] = 0
    skipping = per % 64 == 0
    x_nan = x.clone()
    if skipping and 0 < len(drop) and (B - len(drop)) * (per // 64) >= lib.grit_wgrad_tn_splits(M, N, K):
        for b in drop:
            x_nan[b * per:(b + 1) * per] = float("nan")
    S = lib.grit_wgrad_tn_splits(M, N, K)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    plain = torch.empty((S, N, K), dtype=torch.float32, device=DEV)
    bplain = torch.empty((S, N), dtype=torch.float32, device=DEV)
    assert lib.grit_wgrad_tn(p(dy), dy.stride(0), p(x), x.stride(0), M, N, K, S, p(plain), p(bplain), _lib.current_stream_ptr()) == 0
    part = torch.full((S, N, K), float("nan"), dtype=torch.float32, device=DEV)
    bpart = torch.full((S, N), float("nan"), dtype=torch.float32, device=DEV)
    assert lib.grit_wgrad_tn_rows(p(dy), dy.stride(0), p(x_nan), x.stride(0), M, N, K, S, p(part), p(bpart), p(scale), per,
                                  _lib.current_stream_ptr()) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(part).all()) and bool(torch.isfinite(bpart).all())
    if not drop or not skipping:
        assert torch.equal(part, plain) and torch.equal(bpart, bplain)
    ref = dy.double().t() @ x.double()
    tol = 2e-5 * float(ref.abs().max()) + 1e-4
    assert float((part.double().sum(0) - ref).abs().max()) <= tol
    assert float((part.double().sum(0) - plain.double().sum(0)).abs().max()) <= tol
    refb = dy.double().sum(0)
    assert float((bpart.double().sum(0) - refb).abs().max()) <= 1e-5 * float(refb.abs().max()) + 1e-3
    # the same through the grouped launch, next to a job without factors
    table = (_lib.WgradJob * 2)()
    gpart = torch.full((S, N, K), float("nan"), dtype=torch.float32, device=DEV)
    other = torch.full((S, N, K), float("nan"), dtype=torch.float32, device=DEV)
    table[0] = _lib.WgradJob(dy.data_ptr(), dy.stride(0), x_nan.data_ptr(), x.stride(0), M, N, K, S, gpart.data_ptr(), None, scale.data_ptr(), per)
    table[1] = _lib.WgradJob(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, N, K, S, other.data_ptr(), None, None, 0)
    assert lib.grit_wgrad_tn_grouped(table, 2, _lib.current_stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(gpart, part) and torch.equal(other, plain)
    assert lib.grit_wgrad_tn_rows(p(dy), dy.stride(0), p(x), x.stride(0), M, N, K, S, p(part), None, p(scale), 0, _lib.current_stream_ptr()) != 0


@pytest.mark.parametrize("B,T,d_model,d_ff,p", [(32, 150, 256, 1024, 0.1), (32, 20, 512, 2048, 0.1), (16, 150, 256, 1024, 0.0)])
def test_ffn_relu_dropout_in_the_gemm_epilogues(monkeypatch, B, T, d_model, d_ff, p):
    """Round 6: dropout(relu(fc1(x))) of the decoders' position-wise FFNs as the epilogue of fc1's GEMM (GRIT_GEMM_BIAS_RELU_DROP) and its
    backward as the epilogue of fc2's input-gradient GEMM (GRIT_GEMM_DRELU), against the same block with grit_relu_dropout_{fwd,bwd} as
    launches of their own (GRIT_FFN_RELU_EPILOGUE=0): output and EVERY gradient bit for bit (same hash over the same element index, same
    roundings in the same order), with and without dropout; and without a transposed copy of fc2's weight at hand (the library's product
    followed by the mask kernel: the contract between the two nodes holds on that path too)."""
    from grit_amd.models.common.pos_embed import FeedForward
    from grit_amd.ops import backend
    from grit_amd.ops import gemm as G
    from grit_amd.ops import linear as L
    from grit_amd.ops import transposed
    torch.manual_seed(11)
    ffn = FeedForward(d_model, d_ff, dropout=p).cuda().bfloat16().train()
    x = torch.randn(B, T, d_model, device='cuda').bfloat16()
    cot = torch.randn(B, T, d_model, device='cuda').bfloat16()
    calls = []
    real = G.gemm_nt_relu
    monkeypatch.setattr(G, "gemm_nt_relu", lambda *a, **k: (calls.append(a[2]), real(*a, **k))[1])

    def run(on, with_copies=True):
        monkeypatch.setattr(L, "RELU_EPILOGUE", on)
        for q in ffn.parameters():
            q.grad = None
            q.__dict__.pop("_grit_transposed", None)
        if with_copies:
            transposed.refresh_linears(ffn)
        backend._seeds.buf = None  # (the dropout seeds come in blocks from the device generator: a fresh block ...
        torch.manual_seed(5)       #  ... from the same generator state in every arm)
        del calls[:]
        xi = x.clone().requires_grad_(True)
        y = ffn(xi)
        (y * cot).sum().backward()
        return [y.detach(), xi.grad] + [q.grad.clone() for q in ffn.parameters()], list(calls)

    fused, c_on = run(True)
    plain, c_off = run(False)
    assert c_on == [G.BIAS_RELU_DROP, G.DRELU] and c_off == []
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert torch.equal(a, b), i
    assert fused[1].abs().max().item() > 0
    no_copy, c_nc = run(True, with_copies=False)
    assert c_nc == [G.BIAS_RELU_DROP]  # fc2's input gradient fell back to the library + the mask kernel
    for i, (a, b) in enumerate(zip(no_copy, plain)):
        _close(a.float(), b.float())
