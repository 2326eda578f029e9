"""Cross-entropy training / evaluation loops of the captioner (reference engine/caption_engine.py).

Kept: build_optimizers :18-73 (two Adams split on 'detector' in the parameter name; the groups carry
`weight_decay_rate`, which torch.optim.Adam ignores -> effective weight decay 0), gather_result :76-80,
save_checkpoint :83-103 (same dict layout), evaluate_loss :287-309, train_xe :312-385 (same step order:
forward -> zero_grad x2 -> NLL on shifted log-probs -> backward -> Adam x2 -> scalar all-reduce -> scheduler, and the
extra scheduler.step() before each epoch's loop, Q8), evaluate_metrics :144-230 (the beam-search model call; scoring
is delegated to a caller-supplied scorer because the reference's PTB tokenizer / METEOR are Java programs).
train_sc :388-492 (next-row N2): beam search with gradient + reward supplied by the caller (the reference's CIDEr /
PTB tokenizer are host-side Java).  Out of scope: log_epoch, inference_coco_test.

MI355X specifics: with grit_amd.amp.Bf16Compute the two Adams are FlatAdam (one kernel per parameter run over the flat
bf16-gradient / fp32-master state), otherwise torch's fused multi-tensor Adam; the model may be wrapped either in
torch DDP or in grit_amd.ddp.BucketedDataParallel (gradient buckets all-reduced by RCCL on its side stream while
backward is still running); tqdm / TensorBoard are optional.
"""
import time

import torch
import torch.distributed as dist
from torch.nn import NLLLoss

from grit_amd.data import PAD
from grit_amd.utils.misc import get_rank, get_world_size, is_dist_avail_and_initialized


def _unwrap(model):
    return getattr(model, 'module', model)


def _consolidate(model):
    """End of a training epoch, EVERY rank: with a sharded optimizer (grit_amd.amp.Bf16Compute(shard_optimizer=True)) the fp32
    masters and Adam moments of the other ranks' slices are gathered, so that the checkpoints rank 0 (and the evaluation ranks,
    reference train_caption.py:149-204) write afterwards hold the whole model.  No-op for every other wrapper."""
    fn = getattr(model, 'consolidate', None)
    if fn is not None:
        fn()


def build_optimizers(model, config, mode='xe'):
    # grit_amd.amp.Bf16Compute exposes fp32 masters under the module's parameter names; otherwise the module's own
    masters = getattr(model, 'named_master_parameters', None)
    wrapper = model
    model = _unwrap(model)
    all_named = masters() if masters is not None else list(model.named_parameters())
    no_decay = ['bias', 'gamma', 'beta']

    def groups(in_detector):
        named = [(n, p) for n, p in all_named if p.requires_grad and (('detector' in n) == in_detector)]
        return [
            {'params': [p for n, p in named if any(nd in n for nd in no_decay)], 'weight_decay_rate': 0.0},
            {'params': [p for n, p in named if not any(nd in n for nd in no_decay)],
             'weight_decay_rate': config.optimizer.weight_decay},
        ]

    betas = (config.optimizer.beta_1, config.optimizer.beta_2)
    fused = all(p.is_cuda for _, p in all_named)

    flat = getattr(wrapper, 'flat_adam', None) if getattr(wrapper, 'flat_optimizer', False) else None

    def adam(param_groups, lr):
        param_groups = [g for g in param_groups if len(g['params'])] or param_groups
        if flat is not None:  # grit_amd.amp.FlatAdam: same arithmetic and state layout, one kernel per parameter run
            return flat(param_groups, lr, betas)
        return torch.optim.Adam(param_groups, lr=lr, betas=betas, **({'fused': True} if fused else {}))

    return {
        'model': adam(groups(False), getattr(config.optimizer, f'{mode}_lr', config.optimizer.sc_lr)),
        'backbone': adam(groups(True), getattr(config.optimizer, f'{mode}_backbone_lr', config.optimizer.sc_backbone_lr)),
        'mode': mode,
    }


def gather_result(value):
    """Average a tensor over the ranks (sum all-reduce, then 1/world)."""
    if isinstance(value, torch.Tensor) and is_dist_avail_and_initialized():
        dist.all_reduce(value, async_op=False)
        value.mul_(1.0 / get_world_size())
    return value


def save_checkpoint(model, optimizers, epoch, scores, best_ciders, config=None, filename='checkpoint_last.pth',
                    scheduler=None):
    torch.save(
        {
            "state_dict": (model.master_state_dict() if hasattr(model, 'master_state_dict')
                           else _unwrap(model).state_dict()),
            "optim_model": optimizers['model'].state_dict(),
            "optim_backbone": optimizers['backbone'].state_dict(),
            "scores": scores,
            "best_ciders": best_ciders,
            "epoch": epoch,
            "exp_name": "" if config is None else config.exp.name,
            "scheduler": [] if scheduler is None else scheduler.state_dict(),
        }, filename)


def _progress(iterable, **kw):
    try:
        from tqdm import tqdm
        return tqdm(iterable, **kw)
    except Exception:  # pragma: no cover
        return iterable


def _pad_index(text_field):
    if text_field is None:
        return PAD
    return text_field.vocab.stoi['<pad>']


def xe_loss(model, batch, loss_fn):
    """NLL of token t+1 given tokens <= t (log-probs at positions 0..T-2 vs captions 1..T-1)."""
    out = model(batch['samples'], batch['captions'])
    target = batch['captions'][:, 1:].contiguous()
    out = out[:, :-1].contiguous()
    return loss_fn(out.view(-1, out.shape[-1]), target.view(-1))


def evaluate_loss(model, dataloader, loss_fn, text_field, epoch, writer):
    model.eval()
    running = .0
    with torch.no_grad():
        for it, batch in enumerate(_progress(dataloader, desc='Epoch %d - validation' % epoch, unit='it')):
            loss = gather_result(xe_loss(model, batch, loss_fn))
            running += loss.item()
    val_loss = running / max(1, len(dataloader))
    if writer is not None and get_rank() == 0:
        writer.add_scalar('val_loss', val_loss, epoch)
    return val_loss


def train_xe_step(model, batch, optimizers, loss_fn, scheduler=None, autocast_dtype=None, gather=True):
    """One optimisation step in the reference's order; returns the (rank-averaged) loss tensor, no host sync.
    gather=False leaves the rank average of the loss to the caller (the segmented capture of the step issues it between replays)."""
    dev = batch['captions'].device.type
    with torch.autocast(dev, dtype=autocast_dtype, enabled=autocast_dtype is not None):
        out = model(batch['samples'], batch['captions'])
    optimizers['model'].zero_grad(set_to_none=False)
    optimizers['backbone'].zero_grad(set_to_none=False)
    target = batch['captions'][:, 1:].contiguous()
    out = out[:, :-1].contiguous()
    loss = loss_fn(out.view(-1, out.shape[-1]).float(), target.view(-1))
    loss.backward()
    finalize = getattr(model, 'finish_gradient_sync', None)
    if finalize is not None:
        finalize()
    optimizers['model'].step()
    optimizers['backbone'].step()
    post = getattr(model, 'after_optimizer_step', None)
    if post is not None:
        post()
    loss = gather_result(loss.detach()) if gather else loss.detach()
    if scheduler is not None:
        lr = scheduler.step()
        assert optimizers['model'].param_groups[0]['lr'] == lr, "LR scheduler doesn't work properly."
    return loss


class _XEStepper(object):
    """train_xe's step: the step graph (GRIT_TRAIN_STEP_GRAPH, default 1 since round 5: what bench.py times is what training runs), i.e. -- one rank, Bf16Compute + FlatAdam, no autocast -- the step
    captured ONCE as a HIP graph (grit_amd/engine/graph_step.py) after two eager steps on the first batch shape and replayed for
    every batch of that shape (collators that pad to a fixed size make that every batch but an epoch's last); other batches, and
    everything once the wrapper's live parameter set changed, run eagerly.  The captured graph is kept on the wrapper across epochs
    (one capture per wrapper and process)."""

    def __init__(self, model, optimizers, loss_fn, scheduler, autocast_dtype):
        import os
        self.args = (model, optimizers, loss_fn, scheduler, autocast_dtype)
        self.want = os.environ.get("GRIT_TRAIN_STEP_GRAPH", "1") != "0" and autocast_dtype is None
        self.eager_seen = 0

    def __call__(self, batch):
        model, optimizers, loss_fn, scheduler, autocast_dtype = self.args
        if self.want and batch['captions'].is_cuda:
            from grit_amd.engine import graph_step
            g = getattr(model, '_grit_step_graph', None)
            if g is not None and g.graph is not None:
                if g.loss_fn_ignore == loss_fn.ignore_index and g.matches(batch, optimizers):
                    g.scheduler = scheduler
                    return g(batch)
                if not g.matches(batch, optimizers) and g.matches_shapes(batch):
                    # same batch shape, yet the graph no longer fits: the optimizers were rebuilt (XE -> SC -> XE) or their runs
                    # re-derived (load_state_dict, a changed live set).  The recorded launches are stale for good: drop the graph
                    # (one capture per wrapper and process) and stay on eager launches
                    g.release()
                    model._grit_step_graph = None  # (the graph's private pool -- a full step of activations -- goes back to the allocator)
                    self.want = False
                    import sys
                    sys.stderr.write("train_xe: the captured step no longer fits this wrapper (optimizers rebuilt, load_state_dict or a changed "
                                     "live parameter set); dropped -- eager launches for the rest of the process\n")
            if g is None and graph_step.supported(model, optimizers) and not getattr(model, '_grit_step_graph_taken', False):
                # (with collectives the capture needs the bucket wrapper in steady state -- live set agreed, no late gradient in the
                # last step: until then the steps stay eager and the capture is simply tried again at the next batch)
                if self.eager_seen >= 2 and (not model.ddp.collective or model.ddp.capture_ready()):
                    try:
                        g = graph_step.GraphedXEStep(model, optimizers, loss_fn, batch, scheduler=scheduler, eager_steps=0)
                        g.loss_fn_ignore = loss_fn.ignore_index
                        model._grit_step_graph = g
                        return g(batch)  # (a capture records, it does not run: the first replay is this batch's step)
                    except Exception as e:  # stay on eager launches; say so once
                        import sys
                        sys.stderr.write("train_xe: step graph not captured (%s: %s); eager launches\n" % (type(e).__name__, str(e)[:200]))
                        self.want = False  # (GraphedXEStep cleaned up after itself: graph_step.abandon_capture)
                self.eager_seen += 1
        return train_xe_step(model, batch, optimizers, loss_fn, scheduler, autocast_dtype)


def train_xe(model, dataloaders, optimizers, text_field, epoch, rank=0, config=None, scheduler=None, writer=None,
             autocast_dtype=None, evaluate=True, checkpoint=True):
    model.train()
    loss_fn = NLLLoss(ignore_index=_pad_index(text_field))
    if scheduler is not None:
        scheduler.step()
    running = .0
    n = len(dataloaders['train'])
    stepper = _XEStepper(model, optimizers, loss_fn, scheduler, autocast_dtype)
    for it, batch in enumerate(_progress(dataloaders['train'], desc=f'Epoch {epoch} - train', unit='it')):
        loss = stepper(batch)
        running += loss.item()  # the reference syncs to the host every step as well (:343)
        if rank == 0 and writer is not None:
            writer.add_scalar('backbone_lr', optimizers['backbone'].param_groups[0]['lr'], epoch * n + it)
            writer.add_scalar('model_lr', optimizers['model'].param_groups[0]['lr'], epoch * n + it)
    _consolidate(model)
    val_loss = evaluate_loss(model, dataloaders['valid'], loss_fn, text_field, epoch, writer) \
        if evaluate and 'valid' in dataloaders else 0.0
    if rank == 0 and checkpoint:
        save_checkpoint(model=model, optimizers=optimizers, epoch=epoch, scores=[], best_ciders=(0, 0), config=config,
                        filename='checkpoint_last.pth', scheduler=scheduler)
    if is_dist_avail_and_initialized():
        dist.barrier()
    return {'loss': running / max(1, n), 'reward': 0, 'reward_baseline': 0, 'val_loss': val_loss}


def evaluate_metrics(model, optimizers, dataloader, text_field, epoch=0, split='test', config=None, train_res=None,
                     writer=None, best_cider=None, which='ft_xe', scheduler=None, log_and_save=True, scorer=None):
    """Beam-search every batch (reference :165-183) and hand tokens to `scorer(gts, gen) -> dict` if given.
    Returns (token tensors per batch, seconds per batch) when no scorer is supplied."""
    model.eval()
    times, tokens, gen, gts = [], [], {}, {}
    for it, batch in enumerate(_progress(dataloader, desc=f'Epoch {epoch} - evaluation on {split}', unit='it')):
        t0 = time.time()
        with torch.no_grad():
            out, _ = model(batch['samples'], seq=None, use_beam_search=True, max_len=config.model.beam_len,
                           eos_idx=config.model.eos_idx, beam_size=config.model.beam_size, out_size=1,
                           return_probs=False)
        if out.is_cuda:
            torch.cuda.synchronize()
        times.append(time.time() - t0)
        tokens.append(out)
        if text_field is not None and scorer is not None:
            import itertools
            for i, (gts_i, gen_i) in enumerate(zip(batch['captions'], text_field.decode(out, join_words=False))):
                gen[f'{it}_{i}'] = [' '.join(k for k, _ in itertools.groupby(gen_i))]
                gts[f'{it}_{i}'] = gts_i
    avg = sum(times) / max(1, len(times))
    if scorer is None:
        return tokens, avg
    scores = scorer(gts, gen)
    if log_and_save and best_cider is not None and scores.get('CIDEr', 0) >= best_cider:
        best = (scores['CIDEr'], 0) if split == 'valid' else (0, scores['CIDEr'])
        save_checkpoint(model, optimizers=optimizers, epoch=epoch, scores=scores, best_ciders=best, config=config,
                        filename=f'checkpoint_best_{split}.pth', scheduler=scheduler)
        return scores['CIDEr']
    return scores


def sc_loss(log_probs, reward):
    """Self-critical loss (reference :440-443): -(mean token log-prob of each beam) x (reward - mean reward of the
    image's beams), averaged.  log_probs [B, beam, T] with grad, reward [B, beam]."""
    baseline = torch.mean(reward, -1, keepdim=True)
    return (-torch.mean(log_probs, -1) * (reward - baseline)).mean(), baseline


def train_sc_step(model, batch, optimizers, reward_fn, config):
    """One self-critical step in the reference's order (engine/caption_engine.py:421-449): zero_grad x2 -> beam search
    WITH gradient (out_size = beam_size) -> host reward -> loss -> backward -> barrier -> Adam x2.
    reward_fn(tokens [B, beam, T] int64, batch) -> float tensor [B, beam] on the model's device: in the reference it is
    CIDEr-D of the decoded, PTB-tokenised beams against the ground-truth captions (:433-438), a host-side (Java)
    component that the caller supplies.  Returns (loss, mean reward, mean baseline) as detached device tensors."""
    beam_size, seq_len = config.model.beam_size, config.model.beam_len
    optimizers['model'].zero_grad()
    optimizers['backbone'].zero_grad()
    outs, log_probs = model(batch['samples'], seq=None, use_beam_search=True, max_len=seq_len, eos_idx=config.model.eos_idx,
                            beam_size=beam_size, out_size=beam_size, return_probs=False)
    reward = reward_fn(outs.detach(), batch).to(log_probs.device, torch.float32).view(outs.shape[0], beam_size)
    loss, baseline = sc_loss(log_probs, reward)
    loss.backward()
    finalize = getattr(model, 'finish_gradient_sync', None)
    if finalize is not None:
        finalize()
    if is_dist_avail_and_initialized():
        dist.barrier()  # reference :443
    optimizers['model'].step()
    optimizers['backbone'].step()
    post = getattr(model, 'after_optimizer_step', None)
    if post is not None:
        post()
    return gather_result(loss.detach()), gather_result(reward.mean()), gather_result(baseline.mean())


def cider_reward_fn(cider, text_field, tokenizer_pool=None, tokenize=None):
    """The reference's reward (:433-438): text_field.decode -> PTB tokenisation of generated and ground-truth captions ->
    cider.compute_score(...)[1].  `tokenize` defaults to the native PTB-style tokenizer
    (grit_amd.datasets.caption.metrics.PTBTokenizer.tokenize; the reference's is a Java program); with a `tokenizer_pool` the
    two corpora are tokenised through pool.map as the reference does."""
    import itertools

    import numpy as np

    if tokenize is None:
        from grit_amd.datasets.caption.metrics import PTBTokenizer
        tokenize = PTBTokenizer.tokenize

    def reward_fn(tokens, batch):
        B, beam, T = tokens.shape
        caps_gen = text_field.decode(tokens.view(-1, T))
        caps_gt = list(itertools.chain(*([c] * beam for c in batch['captions'])))
        caps_gen, caps_gt = (tokenizer_pool.map(tokenize, [caps_gen, caps_gt]) if tokenizer_pool is not None
                             else (tokenize(caps_gen), tokenize(caps_gt)))
        reward = cider.compute_score(caps_gt, caps_gen)[1].astype(np.float32)
        return torch.from_numpy(reward).view(B, beam)

    return reward_fn


def train_sc(model, dataloaders, optimizers, cider, text_field, tokenizer_pool, device, epoch, config, rank=0, writer=None,
             tokenize=None, evaluate=True, checkpoint=True):
    """Self-critical epoch (reference engine/caption_engine.py:388-492), same signature plus `tokenize` (the reference
    hard-wires metrics.PTBTokenizer.tokenize, a Java program that is not part of this build)."""
    model.train()
    reward_fn = cider_reward_fn(cider, text_field, tokenizer_pool, tokenize)
    running_loss = running_reward = running_baseline = 0.0
    n = len(dataloaders['train_dict'])
    for it, batch in enumerate(_progress(dataloaders['train_dict'], desc=f'Epoch {epoch} - train', unit='it')):
        loss, reward, baseline = train_sc_step(model, batch, optimizers, reward_fn, config)
        running_loss += loss.item()
        running_reward += reward.item()
        running_baseline += baseline.item()
        if rank == 0 and writer is not None:
            writer.add_scalar('backbone_lr', optimizers['backbone'].param_groups[0]['lr'], epoch * n + it)
            writer.add_scalar('model_lr', optimizers['model'].param_groups[0]['lr'], epoch * n + it)
    loss_fn = NLLLoss(ignore_index=_pad_index(text_field))
    _consolidate(model)
    val_loss = evaluate_loss(model, dataloaders['valid'], loss_fn, text_field, epoch, writer) \
        if evaluate and 'valid' in dataloaders else 0.0
    if rank == 0 and checkpoint:
        save_checkpoint(model=model, optimizers=optimizers, epoch=epoch, scores=[], best_ciders=(0, 0), config=config,
                        filename='checkpoint_last.pth', scheduler=None)
    if is_dist_avail_and_initialized():
        dist.barrier()
    n = max(n, 1)
    return {'loss': running_loss / n, 'reward': running_reward / n, 'reward_baseline': running_baseline / n,
            'val_loss': val_loss}
