"""BASELINE configs 4 and 5 at their per-rank workloads, with assertions (they ran only inside bench.py), and the
gradient synchronisation of A14 against an oracle ON the HIP path.

config 4 (reference train_caption.py:61,205-215 + engine/caption_engine.py:312-350): one rank's share of the 8-GPU job -- 32 synthetic
  640 x 640 images, T = 20, bf16 compute / fp32 masters -- forward + backward + gradient sync through a ONE-RANK RCCL process group
  with every collective issued: equal to the run without a process group (bit for bit where the kernels are deterministic; the
  MSDeformAttn backward sums a cell's contributions in LDS-counter order), and to the fp32-kernel step within bf16 tolerances.
A14: two gloo ranks sharing cuda:0 run the real model on half a batch each; the reduced gradients and the losses equal ONE process on
  the concatenated batch.
config 3: the bf16 step at 16 images against the fp32-kernel step of the same batch.
config 5 (reference models/caption/transformer.py:75-132): beam 5, 20 steps, batch 64 from synthetic 640 x 640-sized features, fp32
  weights: the batched decode equals 64 single-image decodes and the loop that composes the reference's operations one by one.
"""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import build_model, disable_drop_path

pytestmark = pytest.mark.gpu
DEV = "cuda"

PICKS = ('cap_generator.fc.weight', 'grid_net.fc.weight', 'cap_generator.layers.1.self_att.attention.fc_q.weight',
         'detector.det_module.decoder_layers.5.cross_attn.value_proj.weight',
         'detector.det_module.decoder_layers.0.cross_attn.sampling_offsets.weight',
         'detector.backbone.layers.2.blocks.17.attn.qkv.weight', 'detector.backbone.layers.2.blocks.0.mlp.fc1.weight',
         'detector.input_proj.0.0.weight')
# bf16 step against the fp32-kernel step: relative L2 distance of a picked gradient.  Tolerance = 2 x the largest distance measured on
# MI355X at 16 and 32 images (profiles/r05/tolerances_measured.json: 0.012 .. 0.078, the zero-initialised sampling_offsets 0.27; the
# bf16 step's own run-to-run floor is < 0.003).  Until round 5 the bounds were 0.3 / 0.6 for every tensor.
GRAD_TOL = {'cap_generator.fc.weight': 0.03, 'grid_net.fc.weight': 0.12, 'cap_generator.layers.1.self_att.attention.fc_q.weight': 0.12,
            'detector.det_module.decoder_layers.5.cross_attn.value_proj.weight': 0.125,
            'detector.det_module.decoder_layers.0.cross_attn.sampling_offsets.weight': 0.55,
            'detector.backbone.layers.2.blocks.17.attn.qkv.weight': 0.12, 'detector.backbone.layers.2.blocks.0.mlp.fc1.weight': 0.12,
            'detector.input_proj.0.0.weight': 0.16}
# two gloo ranks (half batches, bf16 sums on the wire) against one process on the whole batch: 2 x measured (profiles/r06/a14_measured.jsonl:
# 0.0099 / 0.072 / 0.062 / 0.057 / 0.149 / 0.067 / 0.063 / 0.061 in the order of PICKS).  Until round 6: 0.15 / 0.30 for everything.
A14_TOL = {'cap_generator.fc.weight': 0.02, 'grid_net.fc.weight': 0.145, 'cap_generator.layers.1.self_att.attention.fc_q.weight': 0.125,
           'detector.det_module.decoder_layers.5.cross_attn.value_proj.weight': 0.115,
           'detector.det_module.decoder_layers.0.cross_attn.sampling_offsets.weight': 0.30,
           'detector.backbone.layers.2.blocks.17.attn.qkv.weight': 0.135, 'detector.backbone.layers.2.blocks.0.mlp.fc1.weight': 0.126,
           'detector.input_proj.0.0.weight': 0.122}
LOSS_TOL = 2e-4  # measured 2.1e-5 (32 images) and 3.8e-5 (16 images)


def assert_close_to_fp32_step(bf16, fp32, grad_slack=1.0, loss_tol=LOSS_TOL):
    assert abs(bf16["loss"] - fp32["loss"]) < loss_tol * fp32["loss"], (bf16["loss"], fp32["loss"])
    rels = {n: float(torch.linalg.norm(bf16["grads"][n] - fp32["grads"][n]) / torch.linalg.norm(fp32["grads"][n])) for n in PICKS}
    for n, rel in rels.items():
        assert rel < grad_slack * GRAD_TOL[n], rels
    return rels


DETERMINISTIC = PICKS[:3]  # downstream of the loss only through the caption decoder / grid net: no atomics, no LDS-counter order


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _config4_worker(rank, port, mode, n_images, ret, fill=False, smooth=0):
    """mode: 'plain' (no process group), 'rccl' (one-rank nccl group, self-collectives), 'fp32' (fp32 weights and kernels).
    fill: the name-keyed deterministic fill (tests/golden/fill.py) instead of the modules' own initialisation."""
    from grit_amd.amp import Bf16Compute
    from grit_amd.data import synthetic_batch
    torch.cuda.set_device(0)
    if mode == 'rccl':
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GRIT_DDP_SELF_COLLECTIVES="1")
        dist.init_process_group("nccl", rank=0, world_size=1)
    torch.manual_seed(0)
    model, cfg = build_model(3, fill=fill, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train().to(DEV)
    disable_drop_path(model)
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    batch = synthetic_batch(n_images, 640, 640, 20, device=DEV, seed=4)
    if smooth:
        # low-frequency images (white noise on a smooth x smooth grid, bicubic to 640 x 640): the feature maps then vary smoothly between
        # neighbouring cells, and the gradient w.r.t. a sampling location -- differences of neighbouring values -- is a well-conditioned
        # function of the location (on white-noise images a 0.1-pixel shift of a sampling point decorrelates it)
        coarse = torch.randn(n_images, 3, smooth, smooth, generator=torch.Generator().manual_seed(11)).to(DEV)
        batch['samples'].tensors.copy_(torch.nn.functional.interpolate(coarse, size=(640, 640), mode='bicubic', align_corners=False))
    issued = []
    if mode == 'fp32':
        out = model(batch['samples'], batch['captions'])
        loss = loss_fn(out[:, :-1].reshape(-1, out.shape[-1]), batch['captions'][:, 1:].reshape(-1))
        loss.backward()
        named = dict(model.named_parameters())
    else:
        wrapped = Bf16Compute(model, bucket_mb=64)
        assert wrapped.flat_optimizer and wrapped.ddp.collective == (mode == 'rccl')
        if mode == 'rccl':
            def spy(*a, _f=dist.all_reduce, **k):
                issued.append(int(a[0].numel()))
                return _f(*a, **k)
            dist.all_reduce = spy
        out = wrapped(batch['samples'], batch['captions'])
        loss = loss_fn(out[:, :-1].reshape(-1, out.shape[-1]).float(), batch['captions'][:, 1:].reshape(-1))
        loss.backward()
        wrapped.finish_gradient_sync()
        named = dict(model.named_parameters())
    torch.cuda.synchronize()
    ret["loss"] = float(loss)
    ret["grads"] = {n: named[n].grad.detach().float().cpu() for n in PICKS}
    ret["finite"] = all(bool(torch.isfinite(p.grad).all()) for p in named.values() if p.grad is not None)
    ret["issued"] = issued
    if mode == 'rccl':
        dist.destroy_process_group()


def _run(worker, *args, extra=()):
    """worker(rank, *args, ret, *extra) in a fresh process; returns what it left in `ret`."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(worker, args=args + (ret,) + tuple(extra), nprocs=1, join=True)
        return dict(ret)


def test_config4_rank_workload_through_rccl_equals_no_group_and_fp32_step():
    plain = _run(_config4_worker, _free_port(), 'plain', 32)
    rccl = _run(_config4_worker, _free_port(), 'rccl', 32)
    fp32 = _run(_config4_worker, _free_port(), 'fp32', 32)
    assert plain["finite"] and rccl["finite"] and fp32["finite"]
    assert 8.0 < plain["loss"] < 10.5  # ~ log(10201) for a randomly initialised decoder
    # every bucket went through an RCCL all-reduce (64 MiB buckets of bf16 + the small tail): >= 147 M gradients in all
    assert len(rccl["issued"]) >= 5 and sum(rccl["issued"]) >= 140e6, rccl["issued"]
    # forward is deterministic: the loss is the same number; gradients that do not pass through the MSDeformAttn backward too
    assert rccl["loss"] == plain["loss"]
    for n in DETERMINISTIC:
        assert torch.equal(rccl["grads"][n], plain["grads"][n]), n
    for n in PICKS:
        a, b = rccl["grads"][n], plain["grads"][n]
        assert float(torch.linalg.norm(a - b)) <= 2e-2 * float(torch.linalg.norm(b)) + 1e-12, n
    # against the fp32 kernels end to end (the G8 parity path at this workload): bf16 storage of ~60 layers of activations
    assert_close_to_fp32_step(plain, fp32)


def test_config3_bs16_step_against_the_fp32_kernels():
    """BASELINE config 3 (full GRIT fwd+bwd, one GPU, 16 images of 640 x 640, bf16) against the fp32-kernel step of the same batch
    (tests/test_model_gpu.py compares the bf16 step with its own micro-batches; this is the independent reference): loss within
    2 %, picked gradients within the bf16 noise floor measured for config 4."""
    plain = _run(_config4_worker, _free_port(), 'plain', 16)
    fp32 = _run(_config4_worker, _free_port(), 'fp32', 16)
    assert plain["finite"] and fp32["finite"] and 8.0 < plain["loss"] < 10.5
    assert_close_to_fp32_step(plain, fp32)


# measured on MI355X, round 6 (profiles/r06/filled_step_measured.jsonl, the `smooth 10` line); bound = 2 x measured
FILLED_TOL = {'cap_generator.fc.weight': 0.022, 'grid_net.fc.weight': 0.107, 'cap_generator.layers.1.self_att.attention.fc_q.weight': 0.146,
              'detector.det_module.decoder_layers.5.cross_attn.value_proj.weight': 0.10,
              'detector.det_module.decoder_layers.0.cross_attn.sampling_offsets.weight': 0.70,
              'detector.backbone.layers.2.blocks.17.attn.qkv.weight': 0.108, 'detector.backbone.layers.2.blocks.0.mlp.fc1.weight': 0.119,
              'detector.input_proj.0.0.weight': 0.131}


def test_filled_decoder_step_against_the_fp32_kernels():
    """The same comparison with the name-keyed fill of the golden fixtures instead of the modules' own initialisation: MSDeformAttn's
    `_reset_parameters` zero-initialises sampling_offsets.weight, so in the two tests above its gradient is a sum of cancelling terms and
    is held at 0.55 only; here the sampling points spread over the maps (offsets weight ~ N(0, 1 / 512), bias ~ N(0, 1)) and the bf16
    MSDeformAttn backward's grad_loc path is checked at model level with a non-degenerate tensor.

    What the measurement showed (round 6, profiles/r06/filled_step_measured.jsonl): that tensor's bf16-vs-fp32 distance is set by the
    CONDITIONING of d(output)/d(location) -- differences of neighbouring cells of bf16-stored value maps -- not by the kernel (which
    matches the oracle to 1e-3 on equal inputs, tests/test_msda_gpu.py).  On white-noise images (neighbouring cells uncorrelated: a
    0.1-pixel shift of a point, e.g. from the bf16 box-refinement MLPs, decorrelates the term) it is 0.90; on images that are smooth
    over 16 / 32 / 64 / 128 pixels 0.57 / 0.39 / 0.35 / 0.33: the floor is the 2^-9 rounding of neighbouring values whose difference
    is a few per cent of their size.  Zero-mean rounding noise, cosine 0.94 with the fp32 gradient; every other picked tensor is at
    0.01-0.07.  The test runs the smooth-64 images (GRIT_TEST_SMOOTH=10 coarse cells) and bounds each tensor at 2 x measured."""
    smooth = int(os.environ.get("GRIT_TEST_SMOOTH", "10"))
    plain = _run(_config4_worker, _free_port(), 'plain', 8, extra=(True, smooth))
    fp32 = _run(_config4_worker, _free_port(), 'fp32', 8, extra=(True, smooth))
    assert plain["finite"] and fp32["finite"]
    rels = {n: float(torch.linalg.norm(plain["grads"][n] - fp32["grads"][n]) / torch.linalg.norm(fp32["grads"][n])) for n in PICKS}
    rel_loss = abs(plain["loss"] - fp32["loss"]) / abs(fp32["loss"])
    if os.environ.get("GRIT_TEST_MEASURE"):
        with open(os.environ["GRIT_TEST_MEASURE"], "a") as f:
            f.write(json.dumps({"test": "filled_decoder_step", "rels": rels, "loss": [plain["loss"], fp32["loss"]]}) + "\n")
    assert rel_loss < LOSS_TOL, (plain["loss"], fp32["loss"])  # measured 4e-5
    for n, rel in rels.items():
        assert rel < FILLED_TOL[n], rels


# ---------------------------------------------------------------------------------------------------------------------------------
def _small_batch(n, seed):
    from grit_amd.data import synthetic_batch
    return synthetic_batch(n, 224, 224, 12, device=DEV, seed=seed)


def _a14_worker(rank, world, port, ret):
    import faulthandler
    faulthandler.enable()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from grit_amd.amp import Bf16Compute
    from grit_amd.engine.caption_engine import build_optimizers, train_xe_step
    from grit_amd.utils.misc import NestedTensor
    model, cfg = build_model(3, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})  # deterministic fill: same on every rank
    model.train().to(DEV)
    disable_drop_path(model)
    wrapped = Bf16Compute(model, bucket_mb=64)
    opts = build_optimizers(wrapped, cfg, mode='xe')
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    full = _small_batch(4, seed=21)
    per = 4 // world
    lo = rank * per
    mine = {'samples': NestedTensor(full['samples'].tensors[lo:lo + per].contiguous(), full['samples'].mask[lo:lo + per].contiguous(),
                                    any_padding=False),
            'captions': full['captions'][lo:lo + per].contiguous()}
    # step 1 by hand, to look at the reduced gradients before the optimizer consumes them
    out = wrapped(mine['samples'], mine['captions'])
    loss = loss_fn(out[:, :-1].reshape(-1, out.shape[-1]).float(), mine['captions'][:, 1:].reshape(-1))
    loss.backward()
    wrapped.finish_gradient_sync()
    named = dict(model.named_parameters())
    grads = {n: (named[n].grad.detach().float() / wrapped.ddp.world).cpu() for n in PICKS}  # buckets hold the SUM over ranks
    opts['model'].step()
    opts['backbone'].step()
    wrapped.after_optimizer_step()
    losses = [float(loss)]
    for _ in range(2):
        losses.append(float(train_xe_step(wrapped, mine, opts, loss_fn)))  # (rank-averaged by gather_result)
    # steps 4-6 through the step graph: with two ranks captured in SEGMENTS around the bucket all-reduces (graph_step.py), in the
    # single process as one graph
    from grit_amd.engine import graph_step
    graph_step.SEGMENTS = True  # (GRIT_STEP_GRAPH_SEGMENTS=1: opt-in since round 6)
    assert graph_step.supported(wrapped, opts)
    # (nothing may keep the autograd graph of an EAGER step alive across the capture: its AccumulateGrad nodes would be reused, and
    # they run on the stream they were created on -- the default stream, which a capture on another stream must not touch)
    del out, loss
    step = graph_step.GraphedXEStep(wrapped, opts, loss_fn, mine, eager_steps=0)
    assert (step.plan is not None) == (world > 1)
    for _ in range(3):
        losses.append(float(step(mine)))
    torch.cuda.synchronize()
    if rank == 0:
        ret["grads"], ret["losses"] = grads, losses
        ret["plan"] = None if step.plan is None else [k for k, _ in step.plan]
        ret["unused"] = len(wrapped.unused_parameters)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_a14_two_ranks_on_the_hip_path_equal_one_process_on_the_whole_batch():
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_a14_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        two = dict(ret)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_a14_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
        one = dict(ret)
    assert two["unused"] == one["unused"] == 80  # the reference's static unused set
    # step-1 loss of rank 0 is its half batch's; steps 2-3 are rank averages = the whole-batch loss of the single process
    for i, (a, b) in enumerate(zip(two["losses"][1:], one["losses"][1:])):
        # (steps 4-6 run through the step graphs; the two trajectories drift apart with every Adam step on bf16-rounded sums:
        # 0.2 % / 0.2 % eager, then 0.3 % / 0.3 % / 0.6 % measured)
        assert abs(a - b) < (5e-3 if i < 2 else 1.5e-2) * abs(b), (two["losses"], one["losses"])
    assert one["losses"][-1] < one["losses"][0] and len(one["losses"]) == 6
    assert two["plan"].count('collective') >= 2 and two["plan"].count('wait') == 1 and one["plan"] is None, two["plan"]
    rels = {n: float(torch.linalg.norm(two["grads"][n] - one["grads"][n]) / torch.linalg.norm(one["grads"][n])) for n in PICKS}
    if os.environ.get("GRIT_TEST_MEASURE"):  # (tools: append the measured distances, the bounds below are 2 x their maximum over runs)
        with open(os.environ["GRIT_TEST_MEASURE"], "a") as f:
            f.write(json.dumps({"test": "a14_two_ranks_vs_one", "rels": rels}) + "\n")
    for n in PICKS:
        rel = rels[n]
        # mean over two ranks of half-batch gradients, summed in bf16 on the wire, against the whole-batch gradient: bf16 rounding
        # of the partial results (2^-9 relative per tensor element) and the batch-size dependent reduction orders
        # (gradients through the deformable attention of a freshly filled decoder -- sampling_offsets starts from zero weights -- are
        # sums of cancelling terms: the same noise floor as tests/test_model_gpu.py::test_config3_*)
        # Bound = 2 x the distance measured on MI355X (round 6, three runs, profiles/r06/a14_measured.jsonl: the runs agree to 1e-3 of
        # the value); a rank whose gradient were dropped or doubled would show ~50-100 %
        assert rel < A14_TOL[n], (n, rel, A14_TOL[n])


# ---------------------------------------------------------------------------------------------------------------------------------
def test_config5_batch64_beam5_equals_single_image_decodes_and_reference_order_loop(monkeypatch):
    import grit_amd.models.caption.cap_generator as CG
    import grit_amd.models.caption.transformer as T
    import grit_amd.models.common.attention as A
    from grit_amd.ops import gate as gate_ops
    model, _ = build_model(3)
    model.eval().to(DEV)
    B = 64
    gen = torch.Generator().manual_seed(64)
    feats = {"gri_feat": torch.randn(B, 100, 1024, generator=gen).to(DEV), "reg_feat": torch.randn(B, 150, 512, generator=gen).to(DEV),
             "gri_mask": torch.zeros(B, 1, 1, 100, dtype=torch.bool, device=DEV),
             "reg_mask": torch.zeros(B, 1, 1, 150, dtype=torch.bool, device=DEV)}

    def decode(batch):
        model.cached_features = True
        try:
            with torch.no_grad():
                return model(batch, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=1)
        finally:
            model.cached_features = False

    tokens, lps = decode(feats)
    assert tokens.shape == (B, 20)
    singles = [decode({k: v[i:i + 1].contiguous() for k, v in feats.items()}) for i in range(B)]
    single_tokens = torch.cat([s[0] for s in singles], 0)
    single_lps = torch.cat([s[1] for s in singles], 0)
    # ... == the loop composed of the reference's operations (every inference restructuring off, eager), which also records the
    # per-step candidate margins the way fixtures G7 / G15 do: the top beam + 1 scores of every selection
    monkeypatch.setattr(T, "_GRAPH_DECODE", False)
    monkeypatch.setattr(T, "_FUSED_BEAM_STEP", False)
    monkeypatch.setattr(A, "_KV_CACHE", False)
    monkeypatch.setattr(A, "_KV_FUSED_APPEND", False)
    monkeypatch.setattr(CG, "_FUSED_STEP_INPUTS", False)
    monkeypatch.setattr(gate_ops, "supported", lambda *a, **k: False)
    record, orig = [], model.select

    def select(step, cand, beam_size, **kw):
        flat = cand.reshape(cand.shape[0], -1)
        record.append(torch.topk(flat, beam_size + 1, -1).values)
        return orig(step, cand, beam_size, **kw)

    monkeypatch.setattr(model, "select", select)
    ref_tokens, ref_lps = decode(feats)
    top = torch.stack(record, 1)  # [B, steps, beam + 1]
    margin = (top[..., :-1] - top[..., 1:]).abs().amin((1, 2))
    # fp32 GEMMs of 320 rows and of 5 rows run different library kernels (summation order): where two candidates are closer than
    # that round-off a beam may legitimately take the other one.  Allowed ONLY for an image whose recorded margin is below 1e-4
    # (G15's pinned margins start at 1.2e-4); every other image must agree bit for bit, with the single-image decodes and with the
    # reference-order loop
    pinned = margin >= 1e-4
    assert int(pinned.sum()) >= B // 2, margin.sort().values[:8]  # the criterion must not be vacuous
    same = (tokens == single_tokens).all(1)
    same_ref = (tokens == ref_tokens).all(1)
    assert bool(same[pinned].all()), (margin[~same], (~same).nonzero().flatten())
    assert bool(same_ref[pinned].all()), (margin[~same_ref], (~same_ref).nonzero().flatten())
    score_b, score_s = lps.sum(-1), single_lps.sum(-1)
    assert torch.allclose(score_b[same], score_s[same], rtol=1e-4, atol=1e-4)
    assert torch.allclose(score_b[~same], score_s[~same], rtol=0, atol=2e-4)  # a flipped near-tie changes the score by < 2e-4
    assert torch.allclose(lps.sum(-1), ref_lps.sum(-1), rtol=1e-4, atol=2e-4)
