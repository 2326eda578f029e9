"""HIP MSDeformAttn (through the C ABI) against the golden vectors and the C oracle.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

from oracle import msda as omsda
from tests.test_msda_oracle import kink_mask

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def _run(value, shapes, lsi, loc, aw, cot=None):
    from grit_amd.ops.msda import MSDeformAttnFunction
    v, l, a = value.clone().requires_grad_(True), loc.clone().requires_grad_(True), aw.clone().requires_grad_(True)
    out = MSDeformAttnFunction.apply(v, shapes, lsi, l, a, 64)
    if cot is None:
        return out.detach()
    out.backward(cot)
    return out.detach(), v.grad, l.grad, a.grad


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_reference_forward_checks(golden_dir):
    """models/ops/test.py:31-60: double with allclose defaults, float with rtol 1e-2 / atol 1e-3 (we hold 1e-6)."""
    g = _load(golden_dir, "msda_g1.npz")
    sh, lsi = _t(g["shapes"]), _t(g["lsi"])
    out = _run(_t(g["dbl_value"], torch.float64), sh, lsi, _t(g["dbl_loc"], torch.float64), _t(g["dbl_aw"], torch.float64))
    assert torch.allclose(out.cpu(), torch.from_numpy(g["dbl_out"]))
    out = _run(_t(g["flt_value"]), sh, lsi, _t(g["flt_loc"]), _t(g["flt_aw"]))
    np.testing.assert_allclose(out.cpu().numpy(), g["flt_out"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("d", [30, 32, 64, 71])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 1e-4)])
def test_reference_gradient_cases(golden_dir, d, dtype, tol):
    """models/ops/test.py:63-86 channel counts (four different backward kernels in the reference)."""
    g = _load(golden_dir, "msda_g1.npz")
    sh, lsi = _t(g["shapes"]), _t(g["lsi"])
    out, gv, gl, ga = _run(_t(g[f"g{d}_value"], dtype), sh, lsi, _t(g[f"g{d}_loc"], dtype), _t(g[f"g{d}_aw"], dtype),
                           _t(g[f"g{d}_cot"], dtype))
    rt = tol
    np.testing.assert_allclose(out.cpu().numpy(), g[f"g{d}_out"], rtol=rt, atol=tol * 1e-2)
    np.testing.assert_allclose(gv.cpu().numpy(), g[f"g{d}_gv"], rtol=rt, atol=tol)
    np.testing.assert_allclose(gl.cpu().numpy(), g[f"g{d}_gl"], rtol=rt, atol=tol)
    np.testing.assert_allclose(ga.cpu().numpy(), g[f"g{d}_ga"], rtol=rt, atol=tol)


def test_grit_shape_border_points_fp32(golden_dir):
    """north_star tolerance: 1e-4 fp32 against the reference op."""
    g = _load(golden_dir, "msda_g2.npz")
    out, gv, gl, ga = _run(_t(g["value"]), _t(g["shapes"]), _t(g["lsi"]), _t(g["loc"]), _t(g["aw"]), _t(g["cot"]))
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gv.cpu().numpy(), g["gv"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ga.cpu().numpy(), g["ga"], rtol=1e-4, atol=1e-4)
    keep = ~kink_mask(g["loc"], g["shapes"])
    np.testing.assert_allclose(gl.cpu().numpy()[keep], g["gl"][keep], rtol=1e-4, atol=1e-3)
    # on the kinks the HIP kernel must agree with the CUDA rule, i.e. with the C oracle
    ogv, ogl, oga = omsda.msda_backward(g["value"], g["shapes"], g["lsi"], g["loc"], g["aw"], g["cot"])
    np.testing.assert_allclose(gl.cpu().numpy(), ogl, rtol=1e-4, atol=1e-3)


def _config2(B, seed=0, Lq=150):
    """SURVEY 8(d) config 2 inputs: levels 80/40/20/10, M=8, D=64, L=P=4."""
    gen = torch.Generator().manual_seed(seed)
    shapes = torch.tensor([[80, 80], [40, 40], [20, 20], [10, 10]])
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, M, D, L, P = int(shapes.prod(1).sum()), 8, 64, 4, 4
    value = torch.randn(B, S, M, D, generator=gen)
    ref = torch.rand(B, Lq, 1, 1, 1, 2, generator=gen)
    loc = (ref + 0.05 * torch.randn(B, Lq, M, L, P, 2, generator=gen)).clamp(-0.05, 1.05)
    aw = torch.softmax(torch.randn(B, Lq, M, L * P, generator=gen), -1).view(B, Lq, M, L, P)
    return value, shapes, lsi, loc, aw


def test_config2_forward_backward_vs_oracle():
    value, shapes, lsi, loc, aw = _config2(B=2)
    cot = torch.randn(2, 150, 512, generator=torch.Generator().manual_seed(5))
    out, gv, gl, ga = _run(value.to(DEV), shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV), cot.to(DEV))
    ref = omsda.msda_forward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy())
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    ogv, ogl, oga = omsda.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), cot.numpy())
    np.testing.assert_allclose(gv.cpu().numpy(), ogv, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ga.cpu().numpy(), oga, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gl.cpu().numpy(), ogl, rtol=1e-3, atol=2e-3)


def test_config2_full_size_properties():
    """BASELINE config 2 at full size (B=8): exact check vs the oracle plus size-independent properties."""
    value, shapes, lsi, loc, aw = _config2(B=8, seed=1)
    dv, ds, dl, dloc, daw = (t.to(DEV) for t in (value, shapes, lsi, loc, aw))
    out = _run(dv, ds, dl, dloc, daw)
    ref = omsda.msda_forward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy())
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    # linearity in value
    v2 = torch.randn_like(dv)
    lhs = _run(2.5 * dv + v2, ds, dl, dloc, daw)
    rhs = 2.5 * out + _run(v2, ds, dl, dloc, daw)
    assert torch.allclose(lhs, rhs, rtol=1e-4, atol=1e-4)
    # constant map + interior points: out = sum of weights = 1
    ones = torch.ones_like(dv)
    inner = dloc.clamp(0.06, 0.94)
    o1 = _run(ones, ds, dl, inner, daw)
    assert torch.allclose(o1, torch.ones_like(o1), atol=1e-5)
    # every point outside the map -> exact zeros
    assert not _run(dv, ds, dl, dloc + 5.0, daw).any()
    # sum(grad_value) == sum_q sum(cot * weights-inside): conservation of the scatter
    cot = torch.ones(8, 150, 512, device=DEV)
    _, gv, gl, ga = _run(ones, ds, dl, inner, daw, cot)
    assert abs(gv.sum().item() - cot.numel()) / cot.numel() < 1e-4


def test_ragged_and_odd_shapes():
    """non power-of-two D, L*P not a multiple of 4, 1x1 level, batch not a multiple of anything."""
    gen = torch.Generator().manual_seed(3)
    for (B, M, D, Lq, shapes_l, P) in [(3, 3, 7, 5, [(5, 3), (1, 1)], 3), (1, 1, 1, 1, [(1, 1)], 1),
                                       (2, 4, 32, 9, [(9, 11), (4, 6), (2, 3)], 4), (5, 2, 130, 3, [(3, 3)], 5)]:
        shapes = torch.tensor(shapes_l)
        lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
        S, L = int(shapes.prod(1).sum()), len(shapes_l)
        value = torch.randn(B, S, M, D, generator=gen)
        loc = torch.rand(B, Lq, M, L, P, 2, generator=gen) * 1.4 - 0.2
        aw = torch.rand(B, Lq, M, L, P, generator=gen)
        cot = torch.randn(B, Lq, M * D, generator=gen)
        out, gv, gl, ga = _run(value.to(DEV), shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV), cot.to(DEV))
        ref = omsda.msda_forward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy())
        ogv, ogl, oga = omsda.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), cot.numpy())
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gv.cpu().numpy(), ogv, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(ga.cpu().numpy(), oga, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(gl.cpu().numpy(), ogl, rtol=1e-3, atol=1e-3)


def test_bf16_io_and_errors():
    from grit_amd.ops.msda import ms_deform_attn_forward
    value, shapes, lsi, loc, aw = _config2(B=1, Lq=10)
    dv, ds, dl, dloc, daw = (t.to(DEV) for t in (value, shapes, lsi, loc, aw))
    o32 = ms_deform_attn_forward(dv, ds, dl, dloc, daw, 64)
    o16 = ms_deform_attn_forward(dv.bfloat16(), ds, dl, dloc, daw, 64)
    assert o16.dtype == torch.bfloat16
    assert torch.allclose(o16.float(), o32, rtol=5e-2, atol=5e-2)
    with pytest.raises(RuntimeError, match="contiguous"):
        ms_deform_attn_forward(dv.transpose(1, 2), ds, dl, dloc, daw, 64)
    with pytest.raises(RuntimeError, match="CPU"):
        ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)


def test_shim_module_name():
    """`import MultiScaleDeformableAttention as MSDA` (ms_deform_attn_func.py:18) resolves to the HIP op."""
    import MultiScaleDeformableAttention as MSDA
    value, shapes, lsi, loc, aw = (t.to(DEV) for t in _config2(B=1, Lq=4))
    out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, torch.ones_like(out), 64)
    assert out.shape == (1, 4, 512) and gv.shape == value.shape and gl.shape == loc.shape and ga.shape == aw.shape


def _set_accumulation(msda_op, mode, monkeypatch):
    """mode: False (packed bf16 atomics), "sorted" (gather form, the default) or "staged" (f32 atomics + flush)."""
    monkeypatch.setattr(msda_op, "F32_ACCUMULATE", bool(mode))
    if mode:
        monkeypatch.setattr(msda_op, "F32_METHOD", mode)


@pytest.mark.parametrize("f32_accumulate", [False, "sorted", "staged"])
def test_bf16_value_maps_forward_backward(f32_accumulate, monkeypatch):
    """Training path: value / grad_out in bf16, oracle in fp32 on the same rounded inputs.  grad_value is accumulated
    in f32 with one final rounding (grit_msda_bwd_bf16_staged, the default: the reference's atomicAdd precision; relative L2
    error ~2e-3 = the bf16 rounding of the result) or, opt-in (GRIT_MSDA_BWD_F32ACC=0), in bf16 by packed atomics
    (grit_msda_bwd_bf16acc: rounding of torch's own bf16 scatter backward, ~4e-3)."""
    from grit_amd.ops import msda as msda_op
    value, shapes, lsi, loc, aw = _config2(B=2)
    v16 = value.bfloat16()
    cot = torch.randn(2, 150, 512, generator=torch.Generator().manual_seed(5)).bfloat16()
    _set_accumulation(msda_op, f32_accumulate, monkeypatch)
    out, gv, gl, ga = _run(v16.to(DEV), shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV), cot.to(DEV))
    assert out.dtype == torch.bfloat16 and gv.dtype == torch.bfloat16
    vr, cr = v16.float().numpy(), cot.float().numpy()
    ref = omsda.msda_forward(vr, shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy())
    ogv, ogl, oga = omsda.msda_backward(vr, shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), cr)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)          # bf16 output rounding
    got = gv.float().cpu().numpy()
    rel = np.linalg.norm(got - ogv) / np.linalg.norm(ogv)
    assert rel < (3e-3 if f32_accumulate else 8e-3), rel
    assert np.abs(got - ogv).max() < (1e-2 if f32_accumulate else 3e-2) * np.abs(ogv).max()
    assert (got[ogv == 0] == 0).all()                                                          # untouched pixels stay exactly zero
    np.testing.assert_allclose(ga.cpu().numpy(), oga, rtol=1e-4, atol=1e-4)                       # fp32 accumulations
    np.testing.assert_allclose(gl.cpu().numpy(), ogl, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("f32_accumulate", [False, "sorted", "staged"])
def test_bf16_backward_merges_points_that_share_a_cell(f32_accumulate, monkeypatch):
    """Freshly initialised model: the P points of a level sit in one pixel cell (tiny offsets).  The bf16-accumulating
    backward sums their bilinear weights and issues one update per corner; mixed here with levels whose points differ,
    cells on the border (dead corners) and points outside the map.  Same tolerances as the spread case."""
    value, shapes, lsi, loc, aw = _config2(B=2)
    g = torch.Generator().manual_seed(9)
    ref = torch.rand(2, 150, 1, 1, 1, 2, generator=g)
    hw = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()[None, None, None, :, None, :]  # (W, H) per level
    # cell-centred reference + offsets well inside the cell for levels 0, 2, 3; level 1 keeps spread points
    centre = (torch.floor(ref * hw) + 1.0) / hw  # pixel coordinate k + 0.5: the middle of cell k
    tight = centre + (torch.rand(2, 150, 8, 4, 4, 2, generator=g) - 0.5) * 0.6 / hw
    loc = loc.clone()
    for l in (0, 2, 3):
        loc[:, :, :, l] = tight[:, :, :, l]
    loc[0, :10] = -0.2            # entirely outside: no update at all
    loc[1, :10, :, 3] = 0.999      # last cell of the coarsest level: right / bottom corners dead
    from grit_amd.ops import msda as msda_op
    _set_accumulation(msda_op, f32_accumulate, monkeypatch)
    v16 = value.bfloat16()
    cot = torch.randn(2, 150, 512, generator=g).bfloat16()
    out, gv, gl, ga = _run(v16.to(DEV), shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV), cot.to(DEV))
    vr, cr = v16.float().numpy(), cot.float().numpy()
    ogv, ogl, oga = omsda.msda_backward(vr, shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), cr)
    got = gv.float().cpu().numpy()
    rel = np.linalg.norm(got - ogv) / np.linalg.norm(ogv)
    assert rel < (3e-3 if f32_accumulate else 8e-3), rel
    assert np.abs(got - ogv).max() < (1e-2 if f32_accumulate else 3e-2) * np.abs(ogv).max()
    assert (got[ogv == 0] == 0).all()
    np.testing.assert_allclose(ga.cpu().numpy(), oga, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gl.cpu().numpy(), ogl, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("B", [1, 4, 32])
def test_d64_backward_batches_and_level_layouts(B):
    """msda_bwd_d64 (geometry once per row + butterfly reductions) over several batch sizes and level layouts,
    including 2- and 3-level maps (L*P = 8, 12 < 16 lanes of geometry)."""
    gen = torch.Generator().manual_seed(B)
    for shapes_l in ([(40, 40), (20, 20), (10, 10), (5, 5)], [(9, 7), (5, 4), (3, 2)], [(30, 30), (28, 28)]):
        shapes = torch.tensor(shapes_l)
        lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
        S, L, M, D, Lq, P = int(shapes.prod(1).sum()), len(shapes_l), 8, 64, 40, 4
        value = torch.randn(B, S, M, D, generator=gen)
        loc = torch.rand(B, Lq, M, L, P, 2, generator=gen) * 1.2 - 0.1
        aw = torch.rand(B, Lq, M, L, P, generator=gen)
        cot = torch.randn(B, Lq, M * D, generator=gen)
        out, gv, gl, ga = _run(value.to(DEV), shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV), cot.to(DEV))
        ogv, ogl, oga = omsda.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), cot.numpy())
        np.testing.assert_allclose(gv.cpu().numpy(), ogv, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(ga.cpu().numpy(), oga, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(gl.cpu().numpy(), ogl, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("shapes_l,M,Lq,P,B", [
    ([(40, 40), (20, 20), (10, 10), (5, 5)], 8, 37, 4, 3),   # L*P = 16: four-rows-per-wave kernel, 888 rows (not a multiple of 16)
    ([(9, 7), (5, 4), (3, 2)], 3, 5, 4, 1),                   # L*P = 12, 15 rows: tail wave with dead rows, M not a power of two
    ([(30, 30), (28, 28)], 8, 150, 4, 2),                     # L*P = 8: second half-row idle
    ([(16, 16), (8, 8)], 4, 9, 2, 2),                         # L*P = 4
    ([(40, 40), (20, 20), (10, 10), (5, 5)], 8, 20, 8, 2),    # L*P = 32: wave-per-row kernel
])
def test_bf16_forward_kernels_layouts_and_dead_corners(shapes_l, M, Lq, P, B):
    """grit_msda_fwd_bf16 across its two kernels.  Pixel (0, 0) of every level is NaN and no live corner touches it:
    points are either well inside the map or entirely outside it (dead: clamped address = pixel (0, 0)), so any leak of
    a zero-weight corner into the sum shows up as NaN -- the reference never reads such corners
    (ms_deform_im2col_cuda.cuh:259-262)."""
    from grit_amd.ops.msda import ms_deform_attn_forward
    gen = torch.Generator().manual_seed(len(shapes_l) * 100 + P)
    shapes = torch.tensor(shapes_l)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, L, D = int(shapes.prod(1).sum()), len(shapes_l), 64
    value = torch.randn(B, S, M, D, generator=gen).bfloat16()
    value[:, lsi] = float("nan")
    # inside points: pixel coordinates in [1.6, side - 1.1] never reach row / column 0
    hw = shapes.flip(1).float().view(1, 1, 1, L, 1, 2)  # (W, H) per level
    inside = (1.6 + torch.rand(B, Lq, M, L, P, 2, generator=gen) * (hw - 2.7) + 0.5) / hw
    outside = -(1.5 + torch.rand(B, Lq, M, L, P, 2, generator=gen)) / hw
    dead = torch.rand(B, Lq, M, L, P, 1, generator=gen) < 0.25
    loc = torch.where(dead, outside, inside).contiguous()
    aw = torch.rand(B, Lq, M, L, P, generator=gen)
    out = ms_deform_attn_forward(value.to(DEV), shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV), 64)
    assert out.dtype == torch.bfloat16 and torch.isfinite(out.float()).all()
    ref = omsda.msda_forward(torch.nan_to_num(value.float()).numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy())
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)


def test_staged_f32_accumulation_rounds_once_and_leaves_scratch_zeroed(monkeypatch):
    """grit_msda_bwd_bf16_staged at the benchmark's batch (B = 32, config-2 points): (1) against the dense f32-accumulating kernel
    grit_msda_bwd_bf16 followed by ONE rounding to bf16, the staged result differs by at most the last bf16 bit of a sum whose
    f32 atomics arrived in another order; (2) cells no sampling point reaches stay exactly zero; (3) the staging map and the
    cell flags are all zero again after the call (the invariant the next layer's launch relies on); (4) a second call on the
    same scratch gives the same answer."""
    import ctypes
    from grit_amd import lib as _lib
    from grit_amd.ops import msda as msda_op
    value, shapes, lsi, loc, aw = _config2(B=32)
    v16 = value.bfloat16().to(DEV)
    shapes, lsi, loc, aw = shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV)
    cot = torch.randn(32, 150, 512, generator=torch.Generator().manual_seed(5)).bfloat16().to(DEV)
    assert msda_op.F32_ACCUMULATE  # the default
    monkeypatch.setattr(msda_op, "F32_METHOD", "staged")
    gv, gl, ga = msda_op.ms_deform_attn_backward(v16, shapes, lsi, loc, aw, cot)
    assert gv.dtype == torch.bfloat16
    B, S, M, D = v16.shape
    dense = torch.zeros(v16.shape, dtype=torch.float32, device=DEV)
    gl2, ga2 = torch.empty_like(loc), torch.empty_like(aw)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = _lib.load().grit_msda_bwd_bf16(p(v16), p(shapes), p(lsi), p(loc), p(aw), p(cot), B, S, M, D, 4, 150, 4, p(dense), p(gl2),
                                        p(ga2), _lib.current_stream_ptr())
    assert st == 0
    want = dense.to(torch.bfloat16)
    diff = (gv.float() - want.float()).abs()
    ulp = want.float().abs() * 2.0 ** -7 + 1e-6
    assert bool((diff <= ulp).all()), float((diff / ulp).max())
    assert float((gv != want).float().mean()) < 1e-2   # order-dependent last-bit flips are rare
    # cells no sampling point reaches stay exactly zero (per CELL: inside a touched cell an element may cancel to exactly 0.0 in
    # one summation order and to 2^-27 in another -- seen on hardware)
    untouched = dense.abs().sum(-1) == 0
    assert bool(untouched.any()) and bool((gv[untouched] == 0).all())
    torch.testing.assert_close(gl, gl2, rtol=1e-3, atol=2e-4)  # two kernels, two summation orders over the 64 channels
    torch.testing.assert_close(ga, ga2, rtol=1e-3, atol=2e-4)
    ent = msda_op._STAGE[(str(v16.device), B, S, M)]
    assert not ent[2] and int(ent[0].count_nonzero()) == 0 and int(ent[1].count_nonzero()) == 0
    gv_again, _, _ = msda_op.ms_deform_attn_backward(v16, shapes, lsi, loc, aw, cot)
    assert float((gv_again != gv).float().mean()) < 1e-2 and bool(((gv_again.float() - gv.float()).abs() <= ulp).all())


@pytest.mark.parametrize("points", ["config2", "one_cell"])
def test_sorted_gather_form_backward(points):
    """grit_msda_bwd_bf16_sorted at the benchmark's batch (B = 32), called through the C ABI on a STRIDED gradient map (layer 1
    of 3) that arrives full of NaN (the kernel takes no scratch: everything between the inputs and the rows lives in LDS): (1) every cell of the layer's slice is written -- sums where points sampled, exact zeros
    elsewhere -- and the other layers' slices are not touched; (2) against the dense f32-accumulating kernel grit_msda_bwd_bf16
    followed by ONE rounding to bf16 the result differs by at most the last bf16 bit (another order of the same f32 terms);
    (3) grad_loc / grad_attn_w equal the row walk of the other kernels; (4) "one_cell": every query of an image samples the same
    cell of the coarsest level (runs of 600 contributions in four cells), others empty."""
    import ctypes
    from grit_amd import lib as _lib
    from grit_amd.ops import msda as msda_op
    value, shapes, lsi, loc, aw = _config2(B=32)
    if points == "one_cell":
        loc = loc.clone()
        loc[:, :, :, 3] = 0.55 + 0.01 * torch.rand(loc[:, :, :, 3].shape, generator=torch.Generator().manual_seed(2))
    v16 = value.bfloat16().to(DEV)
    shapes, lsi, loc, aw = shapes.to(DEV), lsi.to(DEV), loc.to(DEV).contiguous(), aw.to(DEV).contiguous()
    cot = torch.randn(32, 150, 512, generator=torch.Generator().manual_seed(5)).bfloat16().to(DEV)
    B, S, M, D = v16.shape
    L, Lq, P, n = 4, 150, 4, 3
    assert msda_op.sorted_applies(B, S, M, L, Lq, P)
    stacked_v = torch.zeros(B, S, n, M, D, dtype=torch.bfloat16, device=DEV)
    stacked_v[:, :, 1] = v16
    grad = torch.full((B, S, n, M, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    gl, ga = torch.empty_like(loc), torch.empty_like(aw)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    layer_off = 1 * M * D * 2
    st = _lib.load().grit_msda_bwd_bf16_sorted(ctypes.c_void_p(stacked_v.data_ptr() + layer_off), n * M * D, p(shapes), p(lsi), p(loc),
                                               p(aw), p(cot), B, S, M, D, L, Lq, P,
                                               ctypes.c_void_p(grad.data_ptr() + layer_off), p(gl), p(ga), _lib.current_stream_ptr())
    assert st == 0
    torch.cuda.synchronize()
    assert bool(torch.isnan(grad[:, :, 0]).all()) and bool(torch.isnan(grad[:, :, 2]).all())
    gv = grad[:, :, 1]
    assert not bool(torch.isnan(gv).any())
    dense = torch.zeros(v16.shape, dtype=torch.float32, device=DEV)
    gl2, ga2 = torch.empty_like(loc), torch.empty_like(aw)
    st = _lib.load().grit_msda_bwd_bf16(p(v16), p(shapes), p(lsi), p(loc), p(aw), p(cot), B, S, M, D, L, Lq, P, p(dense), p(gl2),
                                        p(ga2), _lib.current_stream_ptr())
    assert st == 0
    want = dense.to(torch.bfloat16)
    diff = (gv.float() - want.float()).abs()
    # the same f32 terms in another order: one_cell sums 600 terms of mixed sign per element (cancellation), so the bound is
    # relative to the sum of magnitudes there
    ulp = want.float().abs() * 2.0 ** -7 + (1e-6 if points == "config2" else 2e-4 * float(dense.abs().max()))
    assert bool((diff <= ulp).all()), float((diff / ulp).max())
    if points == "config2":
        assert float((gv != want).float().mean()) < 1e-2
    untouched = dense.abs().sum(-1) == 0
    assert bool(untouched.any()) and bool((gv[untouched] == 0).all())
    torch.testing.assert_close(gl, gl2, rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(ga, ga2, rtol=1e-3, atol=2e-4)


def test_sorted_backward_limits_fall_back_to_the_staged_path(monkeypatch):
    """Shapes beyond one workgroup's LDS (S cells) or register budget (Lq*L*P pairs): the workspace query says unsupported and
    the op takes the staged path (same results within the f32-accumulation tolerance)."""
    from grit_amd.ops import msda as msda_op
    assert msda_op.sorted_applies(2, 8500, 8, 4, 150, 4)
    assert not msda_op.sorted_applies(2, 20000, 8, 4, 150, 4)            # 80 KB of counters + 77 KB of records + rows
    assert not msda_op.sorted_applies(2, 8500, 8, 4, 300, 4)             # 4 800 (query, point) pairs
    monkeypatch.setattr(msda_op, "F32_METHOD", "staged")
    assert not msda_op.sorted_applies(2, 8500, 8, 4, 150, 4)
    monkeypatch.setattr(msda_op, "F32_METHOD", "sorted")
    g = torch.Generator().manual_seed(3)
    shapes = torch.tensor([[20, 20], [10, 10]])
    lsi = torch.tensor([0, 400])
    Lq = 300                                                              # 300 * 2 * 8 = 4 800 pairs: falls back
    value = torch.randn(2, 500, 8, 64, generator=g).bfloat16().to(DEV)
    loc = torch.rand(2, Lq, 8, 2, 8, 2, generator=g).to(DEV)
    aw = torch.softmax(torch.randn(2, Lq, 8, 16, generator=g), -1).view(2, Lq, 8, 2, 8).to(DEV)
    cot = torch.randn(2, Lq, 512, generator=g).bfloat16().to(DEV)
    gv, gl, ga = msda_op.ms_deform_attn_backward(value, shapes.to(DEV), lsi.to(DEV), loc, aw, cot)
    monkeypatch.setattr(msda_op, "F32_ACCUMULATE", False)
    gv2, gl2, ga2 = msda_op.ms_deform_attn_backward(value, shapes.to(DEV), lsi.to(DEV), loc, aw, cot)
    assert torch.linalg.norm(gv.float() - gv2.float()) / torch.linalg.norm(gv2.float()) < 1e-2
    torch.testing.assert_close(gl, gl2, rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("f32_accumulate", ["sorted", "staged", False])
def test_stacked_value_maps_equal_per_layer_maps(f32_accumulate, monkeypatch):
    """Strided kernels (grit_msda_*_strided): three layers' value maps interleaved in one [B, S, 3, M, D] tensor.  Forward
    and the per-row gradients are those of the dense kernels bit for bit (same kernels, other pixel stride); the value
    gradients of all layers land in one buffer that only the node running last hands to autograd."""
    from grit_amd.ops.msda import MSDeformAttnFunction, StackedValueMaps, ms_deform_attn_stacked, stacked_fast_path
    from grit_amd.ops import msda as msda_op
    _set_accumulation(msda_op, f32_accumulate, monkeypatch)
    value, shapes, lsi, loc, aw = _config2(B=2)
    g = torch.Generator().manual_seed(17)
    n = 3
    stacked = torch.randn(2, value.shape[1], n, 8, 64, generator=g).bfloat16().to(DEV).requires_grad_(True)
    shapes, lsi = shapes.to(DEV), lsi.to(DEV)
    locs = [(loc + 0.01 * l).clamp(-0.05, 1.05).to(DEV).requires_grad_(True) for l in range(n)]
    aws = [torch.softmax(torch.randn(2, 150, 8, 16, generator=g), -1).view(2, 150, 8, 4, 4).to(DEV).requires_grad_(True) for _ in range(n)]
    cots = [torch.randn(2, 150, 512, generator=g).bfloat16().to(DEV) for _ in range(n)]
    assert stacked_fast_path(stacked, 4, 4)
    maps = StackedValueMaps(stacked.detach().requires_grad_(True), n)
    outs = [ms_deform_attn_stacked(maps, l, shapes, lsi, locs[l], aws[l]) for l in range(n)]
    assert maps.pending == n
    # layers run their backward in reverse order, as in the decoder
    total = sum((o.float() * c.float()).sum() for o, c in zip(outs, cots))
    total.backward()
    assert maps.pending == 0 and maps.grad is None
    got_gv = maps.stacked.grad
    got = [(locs[l].grad.clone(), aws[l].grad.clone()) for l in range(n)]
    for l in range(n):
        locs[l].grad = aws[l].grad = None
        v = stacked.detach()[:, :, l].contiguous().requires_grad_(True)
        out = MSDeformAttnFunction.apply(v, shapes, lsi, locs[l], aws[l], 64)
        assert torch.equal(out, outs[l])
        (out.float() * cots[l].float()).sum().backward()
        assert torch.equal(locs[l].grad, got[l][0]) and torch.equal(aws[l].grad, got[l][1])
        want, have = v.grad.float(), got_gv[:, :, l].float()
        assert torch.linalg.norm(have - want) / torch.linalg.norm(want) < 1e-2  # bf16 atomics: order-dependent rounding
        assert (have - want).abs().max() < 3e-2 * want.abs().max()
        touched = want.abs().sum(-1) > 0  # (image, pixel, head) rows some point reached
        assert (have.abs().sum(-1)[~touched] == 0).all()


@pytest.mark.parametrize("B,M,levels,P,Lq", [(3, 4, [[7, 9], [3, 5]], 3, 37), (1, 8, [[20, 20]], 4, 5), (5, 2, [[6, 6], [5, 4], [3, 3], [1, 2]], 4, 150),
                                             (2, 8, [[80, 80], [40, 40], [20, 20], [10, 10]], 2, 301)])
def test_sorted_backward_odd_shapes(B, M, levels, P, Lq, monkeypatch):
    """Gather-form backward away from the benchmark's shape: row counts that are no multiple of 4, fewer than 16 points per row,
    a single level, a level of two cells, B x M far from the number of CUs, points on and outside the border -- against the C
    oracle on the bf16-rounded inputs, same tolerances as the config-2 test."""
    from grit_amd.ops import msda as msda_op
    g = torch.Generator().manual_seed(B * 1000 + Lq)
    shapes = torch.tensor(levels)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, L = int(shapes.prod(1).sum()), len(levels)
    value = torch.randn(B, S, M, 64, generator=g).bfloat16()
    loc = (torch.rand(B, Lq, M, L, P, 2, generator=g) * 1.2 - 0.1)
    loc[0, 0] = 0.0
    loc[-1, -1] = 1.0
    aw = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P)
    cot = torch.randn(B, Lq, M * 64, generator=g).bfloat16()
    assert msda_op.sorted_applies(B, S, M, L, Lq, P)
    _set_accumulation(msda_op, "sorted", monkeypatch)
    gv, gl, ga = msda_op.ms_deform_attn_backward(value.to(DEV), shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV), cot.to(DEV))
    ogv, ogl, oga = omsda.msda_backward(value.float().numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), cot.float().numpy())
    got = gv.float().cpu().numpy()
    assert np.linalg.norm(got - ogv) / max(np.linalg.norm(ogv), 1e-20) < 3e-3
    assert np.abs(got - ogv).max() < 1e-2 * np.abs(ogv).max()
    assert (got[ogv == 0] == 0).all()
    np.testing.assert_allclose(ga.cpu().numpy(), oga, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gl.cpu().numpy(), ogl, rtol=1e-3, atol=2e-3)
