"""HIP window attention against the PyTorch oracle and the reference fixture G4 (WindowAttention call form + a whole
BasicLayer with shift / pad / crop / merge): the bf16 MFMA kernels on the same bf16-rounded inputs at bf16 tolerances, the
fp32 kernels (fp32 tensors: the reference's precision) at 1e-4."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from tests.helpers import deterministic_fill_, load, t

pytestmark = pytest.mark.gpu
DEV = "cuda"
# bf16 storage of q/k/v, P and the output: ~3 significant digits
RTOL, ATOL = 3e-2, 3e-2


def _inputs(B, H, W, nH, seed=0, scale_in=1.0):
    g = torch.Generator().manual_seed(seed)
    C = 32 * nH
    qkv = (torch.randn(B, H * W, 3 * C, generator=g) * scale_in).bfloat16()
    bias = torch.randn(nH, 144, 144, generator=g) * 0.5
    pad = (torch.randn(3 * C, generator=g) * 0.3).bfloat16()
    return qkv, bias, pad


def _oracle(qkv, bias, pad, H, W, nH, shift, mask=None):
    return torch_ref.window_attention(qkv.float(), bias, pad.float(), H, W, nH, 12, shift, 32**-0.5, mask=mask)


@pytest.mark.parametrize("B,H,W,nH,shift", [(2, 20, 20, 4, 0), (2, 20, 20, 4, 6), (1, 24, 36, 8, 6), (3, 13, 30, 2, 6),
                                            (1, 12, 12, 1, 0), (2, 40, 40, 16, 6), (1, 7, 5, 4, 6)])
def test_forward_vs_oracle(B, H, W, nH, shift):
    from grit_amd.ops.window_attention import window_attention
    qkv, bias, pad = _inputs(B, H, W, nH, seed=H * W + shift)
    ref = _oracle(qkv, bias, pad, H, W, nH, shift)
    out = window_attention(qkv.to(DEV), bias.to(DEV), pad.to(DEV), H, W, nH, 12, shift, 32**-0.5)
    assert out.dtype == torch.bfloat16 and out.shape == (B, H * W, 32 * nH)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)
    # tighter statistical bound: mean error two orders below the data scale
    assert (out.float().cpu() - ref).abs().mean() < 4e-3


@pytest.mark.parametrize("B,H,W,nH,shift", [(2, 20, 20, 4, 6), (1, 24, 36, 8, 0), (2, 13, 30, 2, 6), (1, 40, 40, 16, 6)])
def test_backward_vs_oracle(B, H, W, nH, shift):
    from grit_amd.ops.window_attention import window_attention
    qkv, bias, pad = _inputs(B, H, W, nH, seed=7 + shift)
    cot = torch.randn(B, H * W, 32 * nH, generator=torch.Generator().manual_seed(1)).bfloat16()
    a, b_, c = qkv.float().requires_grad_(True), bias.clone().requires_grad_(True), pad.float().requires_grad_(True)
    _oracle(a, b_, c, H, W, nH, shift).backward(cot.float())
    x, y, z = qkv.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True), pad.to(DEV).requires_grad_(True)
    window_attention(x, y, z, H, W, nH, 12, shift, 32**-0.5).backward(cot.to(DEV))
    for name, got, ref in (("dqkv", x.grad, a.grad), ("dbias", y.grad, b_.grad), ("dpad", z.grad, c.grad)):
        if ref is None:  # no window padding at this geometry: the oracle never touches pad_qkv
            assert not got.any()
            continue
        got, ref = got.float().cpu(), ref.float()
        scale = ref.abs().max().item() + 1e-6
        err = (got - ref).abs().max().item()
        assert err < 4e-2 * scale, (name, err, scale)
        assert (got - ref).abs().mean().item() < 6e-3 * scale, name


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_explicit_mask_call_form_and_fixture(golden_dir, dtype):
    """WindowAttention.forward(x_windows, mask) (swin_model.py:155-186) on the HIP path vs the reference output: bf16 weights
    -> MFMA kernels (bf16 tolerances), fp32 weights -> fp32 kernels (1e-4, north_star's fp32 bar)."""
    from grit_amd.models.common.swin_model import BasicLayer, PatchMerging
    g = load("win_g4.npz")
    layer = BasicLayer(dim=128, depth=2, num_heads=4, window_size=12, drop_path=[0.0, 0.1], downsample=PatchMerging)
    layer = deterministic_fill_(layer, "g4.").eval().to(DEV, dtype)
    attn = layer.blocks[1].attn
    with torch.no_grad():
        o0 = attn(t(g["xw"], device=DEV).to(dtype), None)
        o1 = attn(t(g["xw"], device=DEV).to(dtype), t(g["attn_mask"], device=DEV))
        x_out, H, W, x_down, Wh, Ww = layer(t(g["x"], device=DEV).to(dtype), 20, 20)
    rtol, atol, rl, al = (RTOL, ATOL, 5e-2, 5e-2) if dtype == torch.bfloat16 else (1e-4, 1e-4, 1e-4, 1e-4)
    np.testing.assert_allclose(o0.float().cpu().numpy(), g["o_nomask"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(o1.float().cpu().numpy(), g["o_mask"], rtol=rtol, atol=atol)
    # two blocks + merge (bf16: accumulated rounding on O(1) activations)
    np.testing.assert_allclose(x_out.float().cpu().numpy(), g["x_out"], rtol=rl, atol=al)
    np.testing.assert_allclose(x_down.float().cpu().numpy(), g["x_down"], rtol=rl, atol=al)


@pytest.mark.parametrize("B,H,W,nH,shift", [(2, 20, 20, 4, 0), (2, 20, 20, 4, 6), (1, 24, 36, 8, 6), (3, 13, 30, 2, 6),
                                            (1, 12, 12, 1, 0), (1, 7, 5, 4, 6)])
def test_fp32_kernels_vs_oracle(B, H, W, nH, shift):
    """fp32 storage + fp32 arithmetic: forward and all three gradients within 1e-4 of the oracle (fp32 on the CPU)."""
    from grit_amd.ops.window_attention import window_attention
    qkv, bias, pad = _inputs(B, H, W, nH, seed=H * W + shift)
    qkv, pad = qkv.float(), pad.float()
    cot = torch.randn(B, H * W, 32 * nH, generator=torch.Generator().manual_seed(1))
    a, b_, c = qkv.clone().requires_grad_(True), bias.clone().requires_grad_(True), pad.clone().requires_grad_(True)
    ref = _oracle(a, b_, c, H, W, nH, shift)
    ref.backward(cot)
    x, y, z = qkv.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True), pad.to(DEV).requires_grad_(True)
    out = window_attention(x, y, z, H, W, nH, 12, shift, 32**-0.5)
    assert out.dtype == torch.float32
    out.backward(cot.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    for name, got, want in (("dqkv", x.grad, a.grad), ("dbias", y.grad, b_.grad), ("dpad", z.grad, c.grad)):
        if want is None:
            assert not got.any()
            continue
        scale = want.abs().max().item() + 1e-6
        assert (got.cpu() - want).abs().max().item() < 1e-4 * scale, (name, (got.cpu() - want).abs().max().item(), scale)


def test_properties_at_benchmark_size():
    """Stage-0 geometry of the 640x640 benchmark (160x160 map -> 14x14 windows, 4 heads), B = 4:
    (i) softmax rows sum to one: V = const -> out = const for every real token;
    (ii) permutation invariance inside a window is respected by the shift: shift and un-shift give the same
         result on a map whose content is periodic with the window size."""
    from grit_amd.ops.window_attention import window_attention
    B, H, W, nH = 4, 160, 160, 4
    qkv, bias, pad = _inputs(B, H, W, nH, seed=3)
    C = 32 * nH
    qkv[..., 2 * C:] = 0.75
    pad[2 * C:] = 0.75
    out = window_attention(qkv.to(DEV), bias.to(DEV), pad.to(DEV), H, W, nH, 12, 6, 32**-0.5)
    assert torch.allclose(out.float(), torch.full_like(out.float(), 0.75), atol=1e-2)
    assert torch.isfinite(out.float()).all()


def test_rejects_unsupported_geometry():
    from grit_amd.ops.window_attention import window_attention
    qkv, bias, pad = _inputs(1, 12, 12, 1)
    with pytest.raises(RuntimeError):
        window_attention(qkv.to(DEV), bias.to(DEV), pad.to(DEV), 12, 12, 1, 7, 0, 1.0)
    with pytest.raises(RuntimeError, match="CPU"):
        window_attention(qkv, bias, pad, 12, 12, 1, 12, 0, 1.0)


@pytest.mark.parametrize("nH,dtype", [(4, torch.float32), (16, torch.float32), (32, torch.bfloat16), (3, torch.bfloat16)])
def test_relative_position_bias_kernels(nH, dtype):
    """grit_relbias_{fwd,bwd} against the reference formulation table[index.view(-1)].view(N, N, nH).permute(2, 0, 1)
    (swin_model.py:168-171) and its autograd gradient."""
    from grit_amd.models.common.swin_model import _relative_position_index
    from grit_amd.ops.rel_bias import relative_position_bias
    g = torch.Generator().manual_seed(nH)
    index = _relative_position_index(12, 12).to(DEV)
    table = torch.randn(529, nH, generator=g).to(dtype).to(DEV).requires_grad_(True)
    cot = torch.randn(nH, 144, 144, generator=g).to(DEV)
    out = relative_position_bias(table, index)
    assert out.dtype == torch.float32 and out.shape == (nH, 144, 144)
    ref_t = table.detach().clone().requires_grad_(True)
    ref = ref_t[index.view(-1)].view(144, 144, nH).permute(2, 0, 1).contiguous().float()
    assert torch.equal(out, ref)
    out.backward(cot)
    ref.backward(cot)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(table.grad.float().cpu().numpy(), ref_t.grad.float().cpu().numpy(), rtol=tol, atol=tol * 10)


def test_grouped_relative_position_gather():
    """grit_relbias_fwd_grouped (round 6: the gathers of all Swin blocks in one launch): every slab bit-equal to the per-module kernel's,
    mixed head counts and table dtypes; handed to relative_position_bias(.., given=) the autograd node yields the same table gradient."""
    from grit_amd.models.common.swin_model import _relative_position_index
    from grit_amd.ops.rel_bias import relative_position_bias, relative_position_bias_grouped
    g = torch.Generator().manual_seed(7)
    index = _relative_position_index(12, 12).to(DEV)
    specs = [(4, torch.bfloat16), (4, torch.bfloat16), (8, torch.float32), (16, torch.bfloat16), (32, torch.bfloat16), (3, torch.float32)]
    tables = [torch.randn(529, nH, generator=g).to(dt).to(DEV).requires_grad_(True) for nH, dt in specs]
    with torch.no_grad():
        slabs = relative_position_bias_grouped(tables, [index] * len(tables))
    assert slabs is not None and len(slabs) == len(tables)
    for t_, slab in zip(tables, slabs):
        one = relative_position_bias(t_, index)
        assert not slab.requires_grad and torch.equal(slab, one.detach())
        cot = torch.randn(slab.shape, generator=g).to(DEV)
        one.backward(cot)
        want, t_.grad = t_.grad.clone(), None
        via = relative_position_bias(t_, index, given=slab)
        assert via.requires_grad and torch.equal(via.detach(), slab)
        via.backward(cot)
        assert torch.equal(t_.grad, want)
    with torch.no_grad():
        assert relative_position_bias(tables[0], index, given=slabs[0]) is slabs[0]  # no autograd: the slab itself
        plain = relative_position_bias_grouped(tables, [index] * len(tables))
    assert all(not q.requires_grad and torch.equal(q, s_.detach()) for q, s_ in zip(plain, slabs))
    # with autograd the slabs are the outputs of ONE node: one backward launch for all tables (grit_relbias_bwd_grouped); a frozen
    # table's slab does not require a gradient
    tables[1].requires_grad_(False)
    for t_ in tables:
        t_.grad = None
    cots = [torch.randn(s_.shape, generator=g).to(DEV) for s_ in slabs]
    want = []
    for t_, c_ in zip(tables, cots):
        if t_.requires_grad:
            relative_position_bias(t_, index).backward(c_)
            want.append(t_.grad.clone())
            t_.grad = None
        else:
            want.append(None)
    outs = relative_position_bias_grouped(tables, [index] * len(tables))
    assert [o.requires_grad for o in outs] == [t_.requires_grad for t_ in tables]
    assert all(relative_position_bias(t_, index, given=o) is o for t_, o in zip(tables, outs) if o.requires_grad)
    torch.autograd.backward([o for o in outs if o.requires_grad], [c_ for o, c_ in zip(outs, cots) if o.requires_grad])
    for t_, w_ in zip(tables, want):
        assert (t_.grad is None) if w_ is None else torch.equal(t_.grad, w_)


def test_backward_conservation_at_benchmark_size():
    """Stage-0 geometry of the 640x640 benchmark (160x160 map padded to 14x14 windows, 4 heads, shift 6), B = 4 -- too big
    for the oracle, checked through a size-independent property of the backward: softmax rows sum to one, so
    sum_j dV_j = sum_i dO_i per head channel, where j runs over the real tokens AND the window-padding tokens (whose share
    lands in d(pad_qkv)); dO of padded queries is zero (they are cropped), and d(bias) sums to zero over keys for every
    query row (d softmax is orthogonal to the all-ones direction)."""
    from grit_amd.ops.window_attention import window_attention
    B, H, W, nH = 4, 160, 160, 4
    C = 32 * nH
    qkv, bias, pad = _inputs(B, H, W, nH, seed=11)
    cot = torch.randn(B, H * W, C, generator=torch.Generator().manual_seed(12)).bfloat16()
    x, y, z = qkv.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True), pad.to(DEV).requires_grad_(True)
    window_attention(x, y, z, H, W, nH, 12, 6, 32**-0.5).backward(cot.to(DEV))
    dv_real = x.grad[..., 2 * C:].float().sum((0, 1))                 # [C]
    dv_pad = z.grad[2 * C:].float()                                   # [C]
    do_sum = cot.float().sum((0, 1)).to(DEV)
    scale = do_sum.abs().max().item() + cot.float().abs().sum().item() * 2e-3 / C
    assert (dv_real + dv_pad - do_sum).abs().max().item() < 2e-2 * scale + 0.5
    row = y.grad.sum(-1)                                              # [nH, 144]: sum over keys of d(bias)
    assert row.abs().max().item() < 2e-2 * y.grad.abs().sum(-1).max().item() + 1e-3
    assert torch.isfinite(x.grad.float()).all() and torch.isfinite(y.grad).all()


def test_register_staged_backward_variant_passes_the_same_tests():
    """The default backward is winattn_bwd_dma (operands of the next window DMA'd into a second LDS tile buffer, bias slab as bf16);
    GRIT_WINATTN_BWD_DMA=0 selects the register-staged winattn_bwd.  The library reads the knob once per process, so the backward
    tests of this file are re-run in a child process with the knob set (the child never recurses into this test)."""
    import os
    import subprocess
    import sys
    if os.environ.get("GRIT_WINATTN_BWD_DMA") == "0" or os.environ.get("GRIT_WINATTN_FWD_DMA") == "0":
        pytest.skip("already a child run")
    env = dict(os.environ, GRIT_WINATTN_BWD_DMA="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k",
                        "backward or explicit_mask or properties"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_register_staged_forward_variant_passes_the_same_tests():
    """The default forward is winattn_fwd_dma (Q / K / V tiles of the next window DMA'd into double-buffered LDS tiles, chunks permuted on
    the source address); GRIT_WINATTN_FWD_DMA=0 selects the register-staged winattn_fwd (also the kernel of explicit-mask calls).  The
    library reads the knob once per process, so the forward tests of this file are re-run in a child process with the knob set."""
    import os
    import subprocess
    import sys
    if os.environ.get("GRIT_WINATTN_FWD_DMA") == "0" or os.environ.get("GRIT_WINATTN_BWD_DMA") == "0":
        pytest.skip("already a child run")
    env = dict(os.environ, GRIT_WINATTN_FWD_DMA="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k",
                        "forward or explicit_mask or properties"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout


def test_dma_and_register_staged_kernels_agree_bit_for_bit_on_random_geometries():
    """36 random geometries (B 1-3, maps 5-50 on a side, 1-16 heads, shift 0 / 6; bf16-representable bias, as the training step's is):
    the DMA-staged forward / backward kernels (chunk permutations, token tables from the loader waves, one barrier per window) and the
    register-staged ones give the same output and the same dqkv BIT FOR BIT, and the same sum |d(bias)|.  Two child processes: the
    library reads the knobs once."""
    import os
    import subprocess
    import sys
    if os.environ.get("GRIT_WINATTN_FWD_DMA") == "0" or os.environ.get("GRIT_WINATTN_BWD_DMA") == "0":
        pytest.skip("a child run")
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "micro", "fuzz_winattn_variants.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = []
    for knobs in ({}, {"GRIT_WINATTN_FWD_DMA": "0", "GRIT_WINATTN_BWD_DMA": "0"}):
        r = subprocess.run([sys.executable, script], env=dict(os.environ, **knobs), cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append([l for l in r.stdout.split("\n") if " fwd " in l])
    assert len(runs[0]) == len(runs[1]) == 36
    for a, b in zip(runs[0], runs[1]):
        fa, fb = a.split(" fwd ")[1].split()[0], b.split(" fwd ")[1].split()[0]
        qa, qb = a.split(" dqkv ")[1].split()[0], b.split(" dqkv ")[1].split()[0]
        assert fa == fb and qa == qb, (a, b)
        da, db = float(a.split(" dbias ")[1].split()[0]), float(b.split(" dbias ")[1].split()[0])
        assert abs(da - db) <= 1e-5 * abs(db) + 1e-9, (a, b)


@pytest.mark.parametrize("B,H,W,nH,shift,drop", [(8, 20, 20, 4, 6, (1, 5)), (4, 24, 36, 8, 0, (0,)), (6, 40, 40, 16, 6, (0, 2, 3, 5)),
                                                 (5, 13, 30, 2, 6, (0, 1, 2, 3, 4)), (3, 20, 20, 4, 6, ())])
def test_backward_skips_the_windows_of_dropped_images(B, H, W, nH, shift, drop):
    """Drop path (round 5): with the per-image factors of the attention branch neither direction computes the windows of images whose
    factor is 0 (the caller multiplies the branch by the factors, so their dO is zero).  Against the plain backward on the same dO: dq / dk / dv of the kept
    images bit for bit, exact zeros for the dropped ones, d(bias) / d(pad) equal up to the order of the float atomics; the operands of
    dropped images are never read (NaNs planted in their q / k / v, O and log-sum-exp rows do not show)."""
    from grit_amd.ops.window_attention import _WindowAttentionFn
    qkv, bias, pad = _inputs(B, H, W, nH, seed=B + H)
    cot = torch.randn(B, H * W, 32 * nH, generator=torch.Generator().manual_seed(2)).bfloat16()
    scale = torch.full((B,), 1.0 / 0.8)
    for b in drop:
        scale[b] = 0.0
        cot[b] = 0
    args = (None, H, W, nH, 12, shift, 32**-0.5)

    def run(row_scale, poison):
        x, y, z = qkv.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True), pad.to(DEV).requires_grad_(True)
        out = _WindowAttentionFn.apply(x, y, z, *args, row_scale)
        result = out.detach().clone()
        if poison:  # what the backward would read for the dropped images: saved q / k / v, O and the log-sum-exps
            with torch.no_grad():
                saved = out.grad_fn.saved_tensors
                for b in drop:  # (.data: no version bump -- autograd must not notice)
                    saved[0].data[b] = float("nan")
                    saved[4].data[b] = float("nan")
                    nw = saved[5].shape[0] // B
                    saved[5].data[b * nw:(b + 1) * nw] = float("nan")
        out.backward(cot.to(DEV))
        return x.grad, y.grad, z.grad, result

    import os
    skipping = os.environ.get("GRIT_WINATTN_BWD_DMA") != "0" and os.environ.get("GRIT_WINATTN_ROW_SKIP") != "0"  # (else: every window is computed)
    plain = run(None, False)
    skipped = run(scale.to(DEV), len(drop) > 0 and skipping)
    assert all(bool(torch.isfinite(t_).all()) for t_ in skipped)
    assert torch.equal(skipped[0], plain[0])
    for b in drop:
        assert not bool(skipped[0][b].any())
    # the forward with the factors: the kept images' outputs bit for bit, zeros for the dropped ones (when the skipping kernels run)
    kept = [b for b in range(B) if b not in drop]
    assert torch.equal(skipped[3][kept], plain[3][kept])
    if skipping:
        for b in drop:
            assert not bool(skipped[3][b].any())
    for got, ref in zip(skipped[1:3], plain[1:3]):
        assert float((got.float() - ref.float()).abs().max()) <= 1e-4 * float(ref.float().abs().max()) + 1e-5


def test_row_skip_check_knob_catches_a_caller_that_breaks_the_promise(monkeypatch):
    """GRIT_ROW_SKIP_CHECK=1 (grit_amd/ops/backend.py check_dropped_rows): `row_scale` is the caller's promise that the gradient of the
    dropped images is zero.  A caller that passes the factors but does not multiply the branch by them gets wrong gradients silently;
    with the knob the backward stops with an error instead, and a caller that keeps the promise passes."""
    from grit_amd.ops import backend
    from grit_amd.ops.window_attention import _WindowAttentionFn
    B, H, W, nH = 4, 20, 20, 4
    qkv, bias, pad = _inputs(B, H, W, nH, seed=3)
    scale = torch.tensor([1.25, 0.0, 1.25, 1.25])
    cot = torch.randn(B, H * W, 32 * nH, generator=torch.Generator().manual_seed(5)).bfloat16()
    monkeypatch.setattr(backend, "ROW_SKIP_CHECK", True)

    def run(cotangent):
        x = qkv.to(DEV).requires_grad_(True)
        out = _WindowAttentionFn.apply(x, bias.to(DEV), pad.to(DEV), None, H, W, nH, 12, 6, 32**-0.5, scale.to(DEV))
        out.backward(cotangent.to(DEV))
        return x.grad

    with pytest.raises(RuntimeError, match="promises zero gradient rows"):
        run(cot)  # image 1 has factor 0 and a non-zero dO
    kept = cot.clone()
    kept[1] = 0
    assert bool(torch.isfinite(run(kept)).all())


def test_pad_gradient_inside_the_qkv_bias_sum_and_one_zero_fill(monkeypatch):
    """Round 6: the q / k / v rows of the window-padding tokens are the qkv bias, so d(pad_qkv) is a second gradient of that bias.  With
    GRIT_WINATTN_PAD_VIA_BIAS the attention's backward leaves its float32 d(pad) for the qkv Linear's backward, which adds it inside the
    bias gradient's slab sum (grit_slab_job.extra) instead of a cast + an autograd add; with GRIT_WINATTN_ZERO_ARENA the d(bias) | d(pad)
    accumulators of all blocks come zeroed from one fill.  Knobs on against knobs off: every gradient of a Swin block, the bias gradient
    within one bf16 rounding of the float32 sum of its two terms; nothing left behind on the parameter or the module."""
    from grit_amd.models.common import swin_model as S
    torch.manual_seed(2)
    C, nH, H, W, B = 256, 8, 30, 26, 3  # (30 x 26 tokens: padded to 36 x 36, so d(pad) is not zero)
    blk = S.SwinTransformerBlock(dim=C, num_heads=nH, window_size=12, shift_size=6, drop_path=0.0).cuda().bfloat16().train()
    blk.H, blk.W = H, W
    x = torch.randn(B, H * W, C, device='cuda').bfloat16()
    cot = torch.randn(B, H * W, C, device='cuda').bfloat16()
    params = [p for p in blk.parameters()]

    class Backbone(torch.nn.Module):  # (_hand_out_backward_workspaces walks self.layers[*].blocks)
        def __init__(self):
            super().__init__()
            stage = torch.nn.Module()
            stage.blocks = torch.nn.ModuleList([blk])
            self.layers = torch.nn.ModuleList([stage])
    bb = Backbone()
    out = []
    for on in (True, False):
        monkeypatch.setattr(S, "_PAD_GRAD_VIA_BIAS", on)
        monkeypatch.setattr(S, "_ZERO_ARENA", on)
        for p in params:
            p.grad = None
        S.SwinTransformer._hand_out_backward_workspaces(bb, x.device)
        assert ("_grit_acc" in blk.attn.__dict__) == on
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        assert "_grit_acc" not in blk.attn.__dict__  # taken by the attention node
        (y * cot).sum().backward()
        assert not hasattr(blk.attn.qkv.bias, "_grit_bias_extra")  # taken by the qkv Linear's backward
        out.append([y.detach().float(), xi.grad.float()] + [p.grad.float().clone() for p in params])
    names = ["y", "dx"] + [n for n, _ in blk.named_parameters()]
    for n, a, b in zip(names, *out):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2.0 ** -7 * scale + 1e-6, n
    i = names.index("attn.qkv.bias")
    assert out[0][i].abs().max().item() > 0 and not torch.equal(out[0][i], out[1][i])  # (one rounding instead of three somewhere)
