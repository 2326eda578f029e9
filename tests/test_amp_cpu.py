"""grit_amd.amp.Bf16Compute host logic on CPU (torch Adam on the fp32 masters; the flat HIP optimizer is covered on the GPU):
loading weights AFTER wrapping (reference train_caption.py:131-132 does model.module.load_state_dict before every
self-critical epoch), fp32 export of frozen tensors, and a used-parameter set that changes between steps."""
import torch
from torch import nn


class Net(nn.Module):

    def __init__(self):
        super().__init__()
        self.frozen = nn.Linear(8, 8)
        for p in self.frozen.parameters():
            p.requires_grad = False
        self.det = nn.Linear(8, 8)
        self.head = nn.Linear(8, 4)
        self.cached = False

    def forward(self, x):
        x = self.frozen(x)
        if not self.cached:
            x = torch.tanh(self.det(x))
        return self.head(x)


def _step(wrapped, opt, x, y):
    loss = ((wrapped(x).float() - y) ** 2).mean()
    loss.backward()
    wrapped.finish_gradient_sync()
    opt.step()
    wrapped.after_optimizer_step()
    return loss.item()


def test_load_state_dict_after_wrapping_reaches_the_masters():
    from grit_amd.amp import Bf16Compute
    torch.manual_seed(0)
    model = Net()
    wrapped = Bf16Compute(model, bucket_mb=0.001)
    assert not wrapped.flat_optimizer
    opt = torch.optim.Adam([m for _, m in wrapped.named_master_parameters()], lr=0.0)
    torch.manual_seed(1)
    ckpt = {k: torch.randn_like(v.float()) * 0.37 for k, v in Net().state_dict().items()}
    wrapped.module.load_state_dict(ckpt)  # the reference's call form: through the inner module
    x, y = torch.randn(4, 8).bfloat16(), torch.randn(4, 4)
    _step(wrapped, opt, x, y)  # lr = 0: the masters rewrite the compute copy -- it must still be the loaded weights
    sd = wrapped.master_state_dict()
    for k, v in ckpt.items():
        assert sd[k].dtype == torch.float32
        assert torch.equal(sd[k], v), k  # fp32 exactly, frozen tensors included (not a bf16 round trip)
        assert torch.equal(dict(model.state_dict())[k].float(), v.bfloat16().float()), k
    # the explicit entry point does the same
    ckpt2 = {k: v + 1 for k, v in ckpt.items()}
    wrapped.load_master_state_dict(ckpt2)
    assert all(torch.equal(wrapped.master_state_dict()[k], v) for k, v in ckpt2.items())


def test_frozen_tensors_are_exported_in_full_precision():
    from grit_amd.amp import Bf16Compute
    torch.manual_seed(0)
    model = Net()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    wrapped = Bf16Compute(model, bucket_mb=0.001)
    sd = wrapped.master_state_dict()
    assert torch.equal(sd["frozen.weight"], before["frozen.weight"])
    assert not torch.equal(before["frozen.weight"], before["frozen.weight"].bfloat16().float())  # the test has teeth


def test_parameters_leave_and_rejoin_the_live_set():
    """Step 0 uses everything, steps 1-2 skip `det` (cached features), step 3 uses it again: while skipped, det's masters do
    not move (torch.optim.Adam skips parameters without gradient) and no stale gradient is consumed; when it rejoins its
    gradient is the fresh one."""
    from grit_amd.amp import Bf16Compute
    torch.manual_seed(0)
    model = Net()
    wrapped = Bf16Compute(model, bucket_mb=0.0002)
    masters = dict(wrapped.named_master_parameters())
    opt = torch.optim.Adam(list(masters.values()), lr=1e-2)
    x, y = torch.randn(4, 8).bfloat16(), torch.randn(4, 4)
    _step(wrapped, opt, x, y)
    model.cached = True
    _step(wrapped, opt, x, y)
    det_after_first_cached = masters["det.weight"].detach().clone()
    assert [n for n, p in model.named_parameters() if any(p is q for q in wrapped.unused_parameters)] == ["det.weight", "det.bias"]
    _step(wrapped, opt, x, y)
    assert torch.equal(masters["det.weight"], det_after_first_cached)  # untouched while outside the live set
    model.cached = False
    head_before = masters["head.weight"].detach().clone()
    _step(wrapped, opt, x, y)
    assert wrapped.unused_parameters == []
    assert not torch.equal(masters["det.weight"], det_after_first_cached)  # stepped again, in the very step it came back
    assert not torch.equal(masters["head.weight"], head_before)


def test_flat_adam_step_counts_survive_freeze_save_load_unfreeze():
    """parameters of one optimizer share a state['step'] tensor per age.  `det` sits out two steps (cached features):
    its state['step'] must stay 1 in the state dict while `head` advances, and after load_state_dict + rejoining it continues with
    the bias corrections of ITS age (torch.optim.Adam's per-parameter step counts)."""
    from grit_amd.amp import Bf16Compute
    from tests.helpers import oracle_ops
    torch.manual_seed(0)
    model = Net()
    wrapped = Bf16Compute(model, bucket_mb=0.0002, flat_optimizer=True)
    masters = dict(wrapped.named_master_parameters())
    opt = wrapped.flat_adam(list(masters.values()), lr=1e-2)
    x, y = torch.randn(4, 8).bfloat16(), torch.randn(4, 4)
    with oracle_ops():
        _step(wrapped, opt, x, y)            # everybody: age 1
        model.cached = True
        _step(wrapped, opt, x, y)            # det outside the live set from the end of this step on
        _step(wrapped, opt, x, y)
        _step(wrapped, opt, x, y)
        ages = {n: int(float(opt.state[m]['step'])) for n, m in masters.items()}
        assert ages["head.weight"] == 4 and ages["head.bias"] == 4, ages
        assert ages["det.weight"] == ages["det.bias"] and ages["det.weight"] < 4, ages
        det_age = ages["det.weight"]
        sd = opt.state_dict()
        steps_saved = sorted({int(float(s['step'])) for s in sd['state'].values()})
        assert steps_saved == sorted({det_age, 4}), steps_saved
        opt2 = wrapped.flat_adam(list(masters.values()), lr=1e-2)
        opt2.load_state_dict(sd)
        assert {n: opt2._steps[m] for n, m in masters.items()} == ages
        model.cached = False
        _step(wrapped, opt2, x, y)
        _step(wrapped, opt2, x, y)  # (the step in which det's gradient arrives late does not step it: it rejoins the runs after it)
        after = {n: int(float(opt2.state[m]['step'])) for n, m in masters.items()}
        assert after["head.weight"] == 6 and det_age < after["det.weight"] <= det_age + 2, after


def test_replay_scalars_are_staged_through_a_ring():
    """FlatAdam.prepare_replay (the per-step scalars a captured step reads from device memory): consecutive calls stage through
    DIFFERENT pinned tables, so that a host running ahead of the device cannot overwrite the scalars of a step whose asynchronous
    copy has not executed yet (seen in a 300-step soak of bench.py: the replayed steps read later steps' bias corrections and the
    loss curve left the eager one); the values are those of the step about to be taken."""
    import math
    from grit_amd.amp import Bf16Compute
    from tests.helpers import oracle_ops
    torch.manual_seed(0)
    model = Net()
    wrapped = Bf16Compute(model, bucket_mb=0.0002, flat_optimizer=True)
    masters = dict(wrapped.named_master_parameters())
    opt = wrapped.flat_adam(list(masters.values()), lr=1e-2)
    x, y = torch.randn(4, 8).bfloat16(), torch.randn(4, 4)
    with oracle_ops():
        _step(wrapped, opt, x, y)
    b1, b2 = opt.param_groups[0]['betas']
    seen = []
    for k in range(opt._HYPER_SLOTS + 2):
        opt.prepare_replay()
        slot = (opt._hyper_slot - 1) % opt._HYPER_SLOTS
        t = opt._runs[0][2] + 1
        want = (1e-2 / (1.0 - b1 ** t), 1.0 / math.sqrt(1.0 - b2 ** t))
        got = opt._hyper_host[slot, 0]
        assert abs(float(got[0]) - want[0]) < 1e-6 * want[0] and abs(float(got[1]) - want[1]) < 1e-6 * want[1]
        assert torch.equal(opt._hyper_dev[0], got)
        seen.append((slot, opt._hyper_host[slot].clone()))
        if k >= 1:  # the previous call's table is untouched by this one
            ps, pv = seen[k - 1]
            assert ps != slot and torch.equal(opt._hyper_host[ps], pv)
        opt.advance()
    assert len({s for s, _ in seen}) == opt._HYPER_SLOTS
