"""N>1 path on CPU: two gloo ranks (127.0.0.1) through grit_amd.ddp.BucketedDataParallel -- bucketed all-reduce from
autograd hooks, static unused-parameter discovery, averaged gradients identical to a single-process run on the
concatenated batch, and the engine's scalar gather_result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


class Toy(nn.Module):

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 4)
        self.dead = nn.Linear(3, 3)  # never used in forward: static unused set
        self.frozen = nn.Linear(8, 8)
        for p in self.frozen.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.b(torch.relu(self.a(self.frozen(x))))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, wire_bf16, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grit_amd.ddp import BucketedDataParallel
    from grit_amd.engine.caption_engine import gather_result
    torch.manual_seed(100 + rank)  # different init per rank: the wrapper must broadcast rank 0's parameters
    model = Toy()
    ddp = BucketedDataParallel(model, bucket_mb=0.0005, wire_dtype=torch.bfloat16 if wire_bf16 else None)
    assert len(ddp.buckets) > 1
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world * 4, 8, generator=g)
    target = torch.randn(world * 4, 4, generator=g)
    xs, ys = data[rank * 4:(rank + 1) * 4], target[rank * 4:(rank + 1) * 4]
    grads = []
    for it in range(3):
        loss = ((ddp(xs) - ys)**2).mean()
        loss.backward()
        ddp.finish_gradient_sync()
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    avg = gather_result(loss.detach().clone())
    if rank == 0:
        ret["params"] = {n: p.detach().clone() for n, p in model.named_parameters()}
        ret["grads"] = grads
        ret["unused"] = [n for n, p in model.named_parameters() if any(p is q for q in ddp.unused_parameters)]
        ret["avg_loss"] = avg.item()
        ret["data"], ret["target"] = data, target
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wire_bf16", [False, True])
def test_bucketed_allreduce_world2(wire_bf16):
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, wire_bf16, ret), nprocs=2, join=True)
        ret = dict(ret)
    # single-process reference on the full batch with rank 0's (broadcast) parameters
    ref = Toy()
    ref.load_state_dict(ret["params"])
    loss = 0.5 * (((ref(ret["data"][:4]) - ret["target"][:4])**2).mean() + ((ref(ret["data"][4:]) - ret["target"][4:])**2).mean())
    loss.backward()
    tol = 2e-2 if wire_bf16 else 1e-6
    for it in range(3):
        for n, p in ref.named_parameters():
            if n.startswith(("dead", "frozen")):
                continue
            assert torch.allclose(ret["grads"][it][n], p.grad, rtol=tol, atol=tol), (it, n)
    assert sorted(ret["unused"]) == ["dead.bias", "dead.weight"]
    assert "dead.weight" not in ret["grads"][1]  # excluded after the first iteration
    assert abs(ret["avg_loss"] - loss.item()) < 1e-5


def _amp_worker(rank, world, port, ret, device="cpu", shard=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grit_amd.amp import Bf16Compute
    from grit_amd.config import default_config
    from grit_amd.engine.caption_engine import build_optimizers
    torch.manual_seed(100 + rank)
    model = Toy().to(device)
    wrapped = Bf16Compute(model, bucket_mb=0.0005, shard_optimizer=shard)
    ret["flat_%d" % rank] = bool(wrapped.flat_optimizer)
    cfg = default_config()
    # the engine's optimizer builder must pick up the fp32 masters (names of the module's parameters)
    opts = build_optimizers(wrapped, cfg, mode='xe')
    n_master = sum(len(g['params']) for g in opts['model'].param_groups)
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world * 4, 8, generator=g)
    target = torch.randn(world * 4, 4, generator=g)
    xs, ys = data[rank * 4:(rank + 1) * 4].bfloat16().to(device), target[rank * 4:(rank + 1) * 4].to(device)
    losses = []
    for it in range(4):
        loss = ((wrapped(xs).float() - ys)**2).mean()
        loss.backward()
        wrapped.finish_gradient_sync()
        opts['model'].step()
        opts['backbone'].step()
        wrapped.after_optimizer_step()
        losses.append(loss.item())
    wrapped.consolidate()
    sd = {k: v.cpu() for k, v in wrapped.master_state_dict().items()}
    flat = torch.cat([v.flatten().float() for v in sd.values()] + [v.flatten().float().cpu() for v in model.state_dict().values()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        ret["same_weights"] = all(torch.equal(gathered[0], t) for t in gathered[1:])
        ret["dtypes"] = sorted({str(v.dtype) for v in sd.values()})
        ret["compute_dtype"] = str(next(model.parameters()).dtype)
        ret["losses"] = losses
        ret["n_master"] = n_master
        ret["keys"] = sorted(sd.keys())
        ret["sd"] = sd
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_compute_fp32_masters_world2():
    """grit_amd.amp.Bf16Compute on two gloo ranks: bf16 flat gradient buckets are all-reduced, fp32 masters stepped by
    Adam stay identical on every rank, compute weights are bf16, the exported state dict is fp32 with module key names."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_amp_worker, args=(2, port, ret), nprocs=2, join=True)
        ret = dict(ret)
    assert ret["same_weights"]
    assert ret["dtypes"] == ["torch.float32"] and ret["compute_dtype"] == "torch.bfloat16"
    assert ret["n_master"] == 6  # a.*, b.* and the never-used dead.* (trainable; their gradient stays zero)
    assert ret["keys"] == sorted(Toy().state_dict().keys())
    assert ret["losses"][-1] < ret["losses"][0]


@pytest.mark.gpu
def test_bf16_compute_flat_adam_world2_on_gpu():
    """The same two-rank step on the GPU (both gloo ranks share cuda:0): here Bf16Compute takes the FlatAdam path
    (grit_adam_flat reading the all-reduced bf16 buckets); masters must stay identical across the ranks.  Then the sharded mode
    (reduce-scatter, each rank's grit_adam_flat on its slice, all-gather of the compute weights, consolidate()): the same
    masters bit for bit (two-term bf16 sums do not depend on the order)."""
    rets = []
    for shard in (False, True):
        port = _free_port()
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_amp_worker, args=(2, port, ret, "cuda", shard), nprocs=2, join=True)
            ret = dict(ret)
        assert ret["flat_0"] and ret["flat_1"]
        assert ret["same_weights"]
        assert ret["dtypes"] == ["torch.float32"] and ret["compute_dtype"] == "torch.bfloat16"
        assert ret["losses"][-1] < ret["losses"][0]
        rets.append(ret)
    for k, v in rets[0]["sd"].items():
        assert torch.equal(v, rets[1]["sd"][k]), k


# ---------------------------------------------------------------------------------------------------------------------
# phase changes of the used-parameter set (reference train_caption.py:105-107: cached-feature epochs with the detector
# unused, then model.module.cached_features = False) and the launch-before-backward-returns / small-tail properties
class Phased(nn.Module):
    """`det` is skipped while cached = True (the reference's cached-feature mode) and used afterwards."""

    def __init__(self):
        super().__init__()
        self.det = nn.Sequential(nn.Linear(8, 8), nn.Tanh())
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 4)
        self.cached = True

    def forward(self, x):
        if not self.cached:
            x = self.det(x)
        return self.b(torch.relu(self.a(x)))


PHASES = (True, False, False, True, True, False)


def _phase_worker(rank, world, port, mode, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grit_amd.ddp import BucketedDataParallel
    torch.manual_seed(5)
    model = Phased()
    ddp = BucketedDataParallel(model, bucket_mb=0.0003, tail_mb=0.0001, repack_unused=(mode == "repack"))
    ddp.check_agreement = True  # GRIT_DDP_CHECK_AGREEMENT=1: assert every step that the ranks took the same decision
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world * 4, 8, generator=g)
    target = torch.randn(world * 4, 4, generator=g)
    xs, ys = data[rank * 4:(rank + 1) * 4], target[rank * 4:(rank + 1) * 4]
    grads, launched, tails = [], [], []
    for cached in PHASES:
        model.cached = cached
        loss = ((ddp(xs) - ys)**2).mean()
        loss.backward()
        # collectives already issued when backward returns (a bucket with no live parameter sends nothing)
        launched.append([b.work is not None or b.expected == 0 for b in ddp.buckets])
        tails.append(ddp.buckets[-1].flat.numel() * ddp.buckets[-1].flat.element_size())
        ddp.finish_gradient_sync()
        grads.append({n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()})
    if rank == 0:
        ret["grads"], ret["launched"], ret["tails"] = grads, launched, tails
        ret["data"], ret["target"] = data, target
        ret["state"] = {k: v.clone() for k, v in model.state_dict().items()}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["repack", "keep_layout"])
def test_used_set_changes_between_steps_world2(mode):
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_phase_worker, args=(2, port, mode, ret), nprocs=2, join=True)
        ret = dict(ret)
    ref = Phased()
    ref.load_state_dict(ret["state"])
    for it, cached in enumerate(PHASES):
        ref.cached = cached
        ref.zero_grad(set_to_none=True)
        d, y = ret["data"], ret["target"]
        (0.5 * (((ref(d[:4]) - y[:4])**2).mean() + ((ref(d[4:]) - y[4:])**2).mean())).backward()
        for n, p in ref.named_parameters():
            got = ret["grads"][it][n]
            if p.grad is None:  # unused in this phase: no gradient, or an all-zero slot that the optimizer does not step
                assert got is None or float(got.abs().max()) == 0.0, (it, n)
            else:
                assert got is not None and torch.allclose(got, p.grad, rtol=1e-5, atol=1e-6), (it, n)
    # steady-state steps (same phase as the step before): every bucket's collective was issued inside backward
    for it in range(1, len(PHASES)):
        if PHASES[it] == PHASES[it - 1] and it >= 2:
            assert all(ret["launched"][it]), (it, ret["launched"][it])


def _real_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from grit_amd.data import synthetic_batch
    from grit_amd.ddp import BucketedDataParallel
    from tests.helpers import build_model, disable_drop_path, oracle_ops
    model, cfg = build_model(2, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train()
    disable_drop_path(model)
    ddp = BucketedDataParallel(model, bucket_mb=64, broadcast_parameters=False)
    sizes = [b.flat.numel() * b.flat.element_size() for b in ddp.buckets]
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    batch = synthetic_batch(1, 64, 64, caption_len=6, seed=50 + rank)
    with oracle_ops():
        for it in range(2):
            out = ddp(batch['samples'], batch['captions'])
            loss = loss_fn(out[:, :-1].reshape(-1, out.shape[-1]), batch['captions'][:, 1:].reshape(-1))
            loss.backward()
            launched = [b.work is not None for b in ddp.buckets]
            ddp.finish_gradient_sync()
    picks = ('cap_generator.fc.weight', 'grid_net.fc.weight', 'detector.det_module.decoder_layers.5.cross_attn.value_proj.weight',
             'detector.backbone.layers.2.blocks.17.attn.qkv.weight', 'detector.backbone.layers.1.blocks.0.mlp.fc1.bias',
             'detector.input_proj.0.0.weight')
    params = dict(model.named_parameters())
    if rank == 0:
        ret["grads"] = {n: params[n].grad.clone() for n in picks}
        ret["sizes"], ret["launched"] = sizes, launched
        ret["n_unused"] = len(ddp.unused_parameters)
        ret["final_sizes"] = [b.flat.numel() * b.flat.element_size() for b in ddp.buckets]
    dist.barrier()
    dist.destroy_process_group()


def test_real_model_bucket_layout_world4():
    """Four gloo ranks, the real GRIT model (deterministic fill, 64 x 64 images, oracle ops on CPU), one image per rank: the
    averaged gradients equal the mean of the four single-image gradients computed in one process; the layout has a small
    tail bucket and, from the second step on, every bucket's all-reduce is issued before backward returns."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_real_worker, args=(4, port, ret), nprocs=4, join=True)
        ret = dict(ret)
    from grit_amd.data import synthetic_batch
    from tests.helpers import build_model, disable_drop_path, oracle_ops
    model, cfg = build_model(2, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train()
    disable_drop_path(model)
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    acc = {}
    params = dict(model.named_parameters())
    with oracle_ops():
        for r in range(4):
            model.zero_grad(set_to_none=True)
            batch = synthetic_batch(1, 64, 64, caption_len=6, seed=50 + r)
            out = model(batch['samples'], batch['captions'])
            loss_fn(out[:, :-1].reshape(-1, out.shape[-1]), batch['captions'][:, 1:].reshape(-1)).backward()
            for n in ret["grads"]:
                acc[n] = acc.get(n, 0) + params[n].grad / 4
    for n, g in ret["grads"].items():
        assert torch.allclose(g, acc[n], rtol=1e-4, atol=1e-6 + 1e-4 * float(acc[n].abs().max())), n
    assert ret["n_unused"] == 78  # the static unused set of the reference (80 at three decoder layers: fixture G8)
    assert ret["final_sizes"][-1] <= 8 * 2**20 and max(ret["final_sizes"]) <= 64 * 2**20 and len(ret["final_sizes"]) >= 3
    assert all(ret["launched"])  # second step: nothing left to send when backward returns


# ---------------------------------------------------------------------------------------------------------------------
# reduce-scatter -> shard-local FlatAdam -> all-gather of the bf16 compute weights (Bf16Compute(shard_optimizer=True)) against
# the all-reduce mode and against ONE process on the concatenated batch; FlatAdam's arithmetic on CPU is the oracle's adam_flat
# (the HIP kernel is covered on the GPU: tests/test_stream_kernels_gpu.py), injected through the test seam
class Wide(nn.Module):
    """Several odd-sized parameters so that slices cut through parameters, plus a never-used one and a frozen one."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 24)
        self.b = nn.Linear(24, 13)
        self.c = nn.Linear(13, 4)
        self.dead = nn.Linear(3, 5)
        self.frozen = nn.Linear(8, 8)
        for p in self.frozen.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.relu(self.a(self.frozen(x))))))


def _shard_worker(rank, world, port, shard, steps, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from grit_amd.amp import Bf16Compute
    from grit_amd.config import default_config
    from grit_amd.engine.caption_engine import build_optimizers
    from tests.helpers import oracle_ops
    torch.manual_seed(3)
    model = Wide()
    wrapped = Bf16Compute(model, bucket_mb=0.0007, flat_optimizer=True, shard_optimizer=shard)
    cfg = default_config(**{'optimizer.xe_lr': 1e-2})
    with oracle_ops():
        opts = build_optimizers(wrapped, cfg, mode='xe')
        g = torch.Generator().manual_seed(7)
        total = 8  # the same 8 samples whatever the world size
        data, target = torch.randn(total, 8, generator=g), torch.randn(total, 4, generator=g)
        n = total // world
        xs, ys = data[rank * n:(rank + 1) * n].bfloat16(), target[rank * n:(rank + 1) * n]
        losses = []
        for it in range(steps):
            loss = ((wrapped(xs).float() - ys) ** 2).mean()
            loss.backward()
            wrapped.finish_gradient_sync()
            opts['model'].step()
            opts['backbone'].step()
            wrapped.after_optimizer_step()
            losses.append(loss.item())
        stale_raises = False
        if shard and world > 1:
            try:
                wrapped.master_state_dict()
            except RuntimeError:
                stale_raises = True
        wrapped.consolidate()
        sd = wrapped.master_state_dict()
        osd = opts['model'].state_dict()
    compute = {k: v.clone() for k, v in model.state_dict().items()}
    if world > 1:
        flat = torch.cat([v.flatten().float() for v in sd.values()] + [v.flatten().float() for v in compute.values()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = all(torch.equal(gathered[0], t) for t in gathered[1:])
    else:
        same = True
    if rank == 0:
        ret["sd"] = sd
        ret["compute"] = compute
        ret["same"] = same
        ret["stale_raises"] = stale_raises
        ret["steps_in_state"] = sorted({float(s['step']) for s in osd['state'].values()})
        ret["moments"] = {i: (s['exp_avg'].clone(), s['exp_avg_sq'].clone()) for i, s in osd['state'].items()}
        ret["n_buckets"] = len(wrapped.ddp.buckets)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_shard(world, shard, steps=3):
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        if world == 1:
            _shard_worker(0, 1, port, shard, steps, ret)
        else:
            mp.spawn(_shard_worker, args=(world, port, shard, steps, ret), nprocs=world, join=True)
        return dict(ret)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_equals_allreduce_and_single_process(world):
    """After 3 steps: masters, bf16 compute weights and Adam moments of the sharded mode (reduce-scatter, every rank steps its
    slice, all-gather of the compute weights, consolidate()) equal those of the all-reduce mode and of ONE process that saw the
    whole batch -- up to the bf16 rounding of the gradient sum on the wire (mean of per-rank means vs one mean)."""
    single = _run_shard(1, False)
    allred = _run_shard(world, False)
    shard = _run_shard(world, True)
    assert shard["same"] and allred["same"]            # every rank holds the same model after consolidate()
    assert shard["stale_raises"]                       # exporting before consolidate() is refused, not silently wrong
    assert shard["n_buckets"] > 1
    assert shard["steps_in_state"] == [0.0, 3.0] or shard["steps_in_state"] == [3.0]  # dead.* never stepped
    # world 2: sharded == all-reduce EXACTLY (a two-term bf16 sum does not depend on the order; the same reduced gradients reach
    # the same Adam arithmetic, only on another rank).  world 4: the two collectives add the four bf16 terms in different orders
    # (Adam turns a last-bit difference of a near-zero gradient into a step of up to lr: most elements agree to 2e-3, none
    # differs by more than 3 steps x lr)
    def close(a, b):
        d = (a.float() - b.float()).abs()
        return float((d <= 2e-3 + 2e-2 * b.float().abs()).float().mean()) >= 0.9 and float(d.max()) <= 3.2e-2
    same = (lambda a, b: torch.equal(a, b)) if world == 2 else close
    for k, v in single["sd"].items():
        assert same(shard["sd"][k], allred["sd"][k]), k
        assert same(shard["compute"][k].float(), allred["compute"][k].float()), k
        assert close(shard["sd"][k], v), k
    for i, (m, v) in allred["moments"].items():
        assert same(shard["moments"][i][0], m) and same(shard["moments"][i][1], v), i


# ---------------------------------------------------------------------------------------------------------------------
# the collectives of the gradient sync on RCCL itself: a one-rank nccl group with GRIT_DDP_SELF_COLLECTIVES=1 issues every
# all-reduce / reduce-scatter / all-gather the N-rank run would (they degenerate to device copies), on the process group's
# stream, from the same hooks.  What a 1-GPU box can show of the contract backend.
def _self_collectives_worker(rank, port, shard, selfc, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if selfc:
        os.environ["GRIT_DDP_SELF_COLLECTIVES"] = "1"
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
    from grit_amd.amp import Bf16Compute
    from grit_amd.config import default_config
    from grit_amd.engine.caption_engine import build_optimizers
    torch.manual_seed(3)
    model = Wide().cuda()
    wrapped = Bf16Compute(model, bucket_mb=0.0007, shard_optimizer=shard)
    assert wrapped.flat_optimizer and wrapped.ddp.collective == selfc
    opts = build_optimizers(wrapped, default_config(**{'optimizer.xe_lr': 1e-2}), mode='xe')
    g = torch.Generator().manual_seed(7)
    xs, ys = torch.randn(8, 8, generator=g).cuda().bfloat16(), torch.randn(8, 4, generator=g).cuda()
    issued = []
    if selfc:
        for name in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor"):
            def spy(*a, _f=getattr(dist, name), _n=name, **k):
                issued.append(_n)
                return _f(*a, **k)
            setattr(dist, name, spy)
    losses = []
    for it in range(3):
        loss = ((wrapped(xs).float() - ys) ** 2).mean()
        loss.backward()
        wrapped.finish_gradient_sync()
        opts['model'].step()
        opts['backbone'].step()
        wrapped.after_optimizer_step()
        losses.append(loss.item())
    wrapped.consolidate()
    torch.cuda.synchronize()
    ret["sd"] = {k: v.cpu() for k, v in wrapped.master_state_dict().items()}
    ret["compute"] = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ret["losses"] = losses
    ret["issued"] = sorted(set(issued))
    if selfc:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("shard", [False, True])
def test_gradient_sync_collectives_on_rccl_one_rank(shard):
    """nccl (= RCCL) backend, one rank, every collective issued: same masters / compute weights / losses, bit for bit, as the run
    without a process group; the collectives that ran are the mode's own."""
    out = []
    for selfc in (False, True):
        port = _free_port()
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_self_collectives_worker, args=(port, shard, selfc, ret), nprocs=1, join=True)
            out.append(dict(ret))
    plain, rccl = out
    assert rccl["losses"] == plain["losses"] and rccl["losses"][-1] < rccl["losses"][0]
    for k, v in plain["sd"].items():
        assert torch.equal(v, rccl["sd"][k]), k
    for k, v in plain["compute"].items():
        assert torch.equal(v, rccl["compute"][k]), k
    if shard:
        assert "reduce_scatter_tensor" in rccl["issued"] and "all_gather_into_tensor" in rccl["issued"]
    else:
        assert "all_reduce" in rccl["issued"] and "reduce_scatter_tensor" not in rccl["issued"]
