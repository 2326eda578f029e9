"""A whole training step with stochastic depth ON (timm DropPath as the reference's Swin blocks use it,
reference models/common/swin_model.py:289-298: per-sample keep w.p. 1 - p, kept branches scaled by 1 / (1 - p)).

Every other model-level comparison turns DropPath off (tests/helpers.py disable_drop_path); the step the benchmark times has it on
and takes shortcuts because of it: 256-row tiles inside a dropped sample are not computed by the fused Mlp GEMMs (forward: fc1 + GELU;
backward: the fc2 input gradient x GELU'), the row panels are dealt round-robin over the XCDs, `add_layernorm` multiplies by the
per-sample factor.  Here the per-sample draw is INJECTED (`SwinTransformer.drop_path_uniforms`) so that different runs see the same
keep mask, at BASELINE config 3's size (16 images of 640 x 640):

  * bf16 step with the skip paths  vs  the fp32-kernel step, same mask: loss and picked gradients at config 3's tolerances;
  * skip paths on  vs  off (GRIT_GEMM_ROW_SKIP=0), both on the eight-wave fused-Mlp kernel (GRIT_GEMM_VARIANT=4): the loss bit-equal,
    and with the region branch cut off the backbone (the MSDeformAttn backward sums a cell's terms in LDS-counter order, the only
    run-to-run non-determinism of the step) EVERY picked gradient, backbone included, bit-equal.
    (Why the variant is pinned: without drop-path factors the stage-3 Mlp runs the four-wave kernel, whose GELU epilogue evaluates
    GELU of the bf16-ROUNDED pre-activation -- what an unfused Linear -> GELU pair computes -- while the eight-wave kernel, which the
    skip path always runs, evaluates it on the fp32 sum: two correct roundings, 1.5e-5 apart in the loss.  First run of this test.)
"""
import os

import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import build_model
from tests.test_configs_gpu import DETERMINISTIC, PICKS, _free_port, assert_close_to_fp32_step

pytestmark = pytest.mark.gpu
DEV = "cuda"
BACKBONE_PICKS = tuple(n for n in PICKS if 'backbone' in n) + ('detector.backbone.layers.3.blocks.1.mlp.fc2.weight',
                                                               'detector.backbone.layers.2.blocks.9.mlp.fc2.bias',
                                                               'detector.backbone.layers.2.blocks.4.norm2.weight')


def _worker(rank, port, mode, cut_regions, pin_variant, n_images, ret):
    """mode: 'skip' (bf16, default paths), 'noskip' (bf16, GRIT_GEMM_ROW_SKIP=0), 'fp32' (fp32 weights and kernels)."""
    if mode == 'noskip':
        os.environ["GRIT_GEMM_ROW_SKIP"] = "0"  # read when grit_amd.ops.gemm is imported: this is a fresh process
        os.environ["GRIT_WGRAD_ROW_SKIP"] = "0"  # ... and by libgrit_hip.so / grit_amd.ops.linear: the weight gradients load every row
        os.environ["GRIT_WINATTN_ROW_SKIP"] = "0"  # ... the window-attention backward computes every window
    if pin_variant:
        os.environ["GRIT_GEMM_VARIANT"] = "4"
    from grit_amd.amp import Bf16Compute
    from grit_amd.data import synthetic_batch
    from grit_amd.models.common.swin_model import DropPath
    from grit_amd.ops import gemm as gemm_ops
    assert gemm_ops.ROW_SKIP == (mode != 'noskip')
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    model, cfg = build_model(3, fill=False, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train().to(DEV)
    backbone = model.detector.backbone
    blocks = [blk for stage in backbone.layers for blk in stage.blocks
              if blk.training and isinstance(blk.drop_path, DropPath) and blk.drop_path.drop_prob > 0.]
    assert len(blocks) >= 18  # DropPath is ON in every trainable block
    u = torch.rand(2 * len(blocks), n_images, generator=torch.Generator().manual_seed(77))
    backbone.drop_path_uniforms = u.to(DEV)
    keep = torch.tensor([1.0 - b.drop_path.drop_prob for b in blocks for _ in range(2)])[:, None]
    ret["dropped"] = int((u >= keep).sum())
    ret["dropped_mlp"] = int((u >= keep)[1::2].sum())
    if cut_regions:  # gradients reach the backbone through the grid feature only: every kernel on that path is deterministic
        det = model.detector.det_module
        orig = det.forward
        det.forward = lambda *a, src_flatten=None, **k: orig(*a, src_flatten=src_flatten.detach(), **k)
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    batch = synthetic_batch(n_images, 640, 640, 20, device=DEV, seed=4)
    if mode == 'fp32':
        out = model(batch['samples'], batch['captions'])
        loss = loss_fn(out[:, :-1].reshape(-1, out.shape[-1]), batch['captions'][:, 1:].reshape(-1))
        loss.backward()
    else:
        wrapped = Bf16Compute(model, bucket_mb=64)
        out = wrapped(batch['samples'], batch['captions'])
        loss = loss_fn(out[:, :-1].reshape(-1, out.shape[-1]).float(), batch['captions'][:, 1:].reshape(-1))
        loss.backward()
        wrapped.finish_gradient_sync()
    named = dict(model.named_parameters())
    torch.cuda.synchronize()
    ret["loss"] = float(loss)
    ret["grads"] = {n: named[n].grad.detach().float().cpu() for n in PICKS + BACKBONE_PICKS if named[n].grad is not None}
    ret["finite"] = all(bool(torch.isfinite(p.grad).all()) for p in named.values() if p.grad is not None)


def _run(*args):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(_free_port(),) + args + (ret,), nprocs=1, join=True)
        return dict(ret)


def test_step_with_drop_path_on_against_fp32_kernels():
    B = 16
    skip = _run('skip', False, False, B)
    fp32 = _run('fp32', False, False, B)
    assert skip["finite"] and fp32["finite"]
    assert skip["dropped_mlp"] >= 20 and skip["dropped"] == fp32["dropped"]  # ~16 % of 44 x 16 (branch, sample) pairs
    # same mask, fp32 kernels: config 3's tolerances (tests/test_configs_gpu.py::test_config3_bs16_step_against_the_fp32_kernels)
    # (the measured per-tensor bounds of the DropPath-off comparison, tests/test_configs_gpu.py GRAD_TOL, with 25 % slack: fewer live
    # (sample, branch) pairs carry the same rounding noise)
    assert_close_to_fp32_step(skip, fp32, grad_slack=1.25, loss_tol=5e-4)


def test_skipped_tiles_change_no_bit_of_the_loss_or_of_any_gradient():
    B = 16
    skip = _run('skip', True, True, B)
    noskip = _run('noskip', True, True, B)
    assert skip["finite"] and noskip["finite"] and skip["dropped_mlp"] >= 20
    assert skip["loss"] == noskip["loss"]
    for n in DETERMINISTIC:
        assert torch.equal(skip["grads"][n], noskip["grads"][n]), n
    exact = 0
    for n in BACKBONE_PICKS:  # (the detection module's own parameters still sit behind its MSDeformAttn backward)
        a, b = skip["grads"][n], noskip["grads"][n]
        assert float(a.abs().max()) > 0, n
        if 'norm' in n:  # LayerNorm sums: untouched by any skip path -- bit for bit
            assert torch.equal(a, b), n
            exact += 1
            continue
        # Weight (and, as the kernel's by-product, bias) gradients of the long maps: the skipped GEMM tiles change no bit of their
        # operands, but the weight-gradient kernel shares the LIVE rows among its slices (grit_wgrad_tn_rows), so the fp32 slice sums are
        # taken in another order: equal up to that -- a bf16 ulp on a few elements (measured below), nowhere near a missing sample
        diff = (a - b).abs()
        assert float(torch.linalg.norm(a - b)) <= 2e-3 * float(torch.linalg.norm(b)), n
        assert float((diff > 0).float().mean()) < 0.2 and float(diff.max()) <= 2.0 ** -7 * float(b.abs().max()), n
    assert exact >= 1
