"""Entry point with the reference's shape (train_caption.py:24-209): `main(gpu, config)` per process, one process per
GPU, torch.distributed 'nccl' (= RCCL on ROCm) over xGMI.

    python train_caption.py [--config my.yaml] [--gpus N] [--synthetic-steps K] [key=value ...]
    torchrun --nproc-per-node N train_caption.py ...      (RANK/LOCAL_RANK/WORLD_SIZE from the environment)

hydra is not installed in this image; configuration is grit_amd.config (same keys as configs/caption/coco_config.yaml,
dotted key=value overrides).  The COCO reader is out of scope of this build: batches come from
grit_amd.data.SyntheticLoader unless the caller passes its own `dataloaders` dict to main().

Phases follow the reference's epoch schedule (train_caption.py:92-147): fr_xe -> fr_sc -> ft_xe -> ft_sc, `cached_features`
while the loader serves cached detector outputs, optimizers rebuilt when the mode flips between 'xe' and 'sc', and the best
validation checkpoint loaded into the (wrapped) model before every self-critical epoch.  The self-critical phases need
`dataloaders['train_dict']`, a `text_field` (decode) and the training captions for the CIDEr statistics; without them they are
skipped (synthetic runs).
"""
import argparse
import os
import random

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from grit_amd.config import default_config, load_yaml
from grit_amd.data import SyntheticLoader
from grit_amd.amp import Bf16Compute
from grit_amd.ddp import BucketedDataParallel
from grit_amd.tuning import load_tuned_gemms
from engine.caption_engine import *  # noqa: F401,F403  (reference does the same star import)
from engine.caption_engine import build_optimizers, save_checkpoint, train_sc, train_xe
from models.caption import Transformer
from models.caption.detector import build_detector
from utils.cap_scheduler import CosineLRScheduler


def phase_of(epoch, opt):
    """reference train_caption.py:92-107."""
    fr_xe = opt.freezing_xe_epochs
    fr_sc = fr_xe + opt.freezing_sc_epochs
    ft_xe = fr_sc + opt.finetune_xe_epochs
    ft_sc = ft_xe + opt.finetune_sc_epochs
    if epoch < fr_xe:
        return 'fr_xe'
    if epoch < fr_sc:
        return 'fr_sc'
    if epoch < ft_xe:
        return 'ft_xe'
    return 'ft_sc' if epoch < ft_sc else None


def main(gpu, config, dataloaders=None, text_field=None, cider=None, tokenizer_pool=None, finetune_dataloaders=None):
    """dataloaders: {'train': ...} (+ 'train_dict' for the self-critical phases, 'valid' for the validation loss); when the
    first phases run on cached detector features pass those loaders here and the image loaders as `finetune_dataloaders`
    (the reference rebuilds them at the switch, train_caption.py:105-107).  cider: a Cider built from the tokenised training
    captions (grit_amd.datasets.caption.metrics)."""
    rank = int(os.environ.get('RANK', config.exp.rank * config.exp.ngpus_per_node + gpu))
    world = int(os.environ.get('WORLD_SIZE', config.exp.world_size))
    use_cuda = torch.cuda.is_available()
    if not dist.is_initialized():
        dist.init_process_group('nccl' if use_cuda else 'gloo', 'env://', rank=rank, world_size=world)
    torch.manual_seed(config.exp.seed)
    np.random.seed(config.exp.seed)
    random.seed(config.exp.seed)
    device = torch.device(f"cuda:{gpu}" if use_cuda else "cpu")
    if use_cuda:
        torch.cuda.set_device(gpu)
        load_tuned_gemms()  # library GEMM solutions tuned for this model's shapes (grit_amd/tuning.py)

    detector = build_detector(config).to(device)
    model = Transformer(detector=detector, config=config).to(device)
    opt = config.optimizer
    start_epoch = int(getattr(config.exp, 'start_epoch', 0))
    if start_epoch < opt.freezing_xe_epochs:
        if getattr(opt, 'freeze_backbone', False):
            for n, p in model.named_parameters():
                if 'backbone' in n:
                    p.requires_grad = False
        if getattr(opt, 'freeze_detector', False):
            for n, p in model.named_parameters():
                if 'detector' in n:
                    p.requires_grad = False
    raw = model
    raw.cached_features = bool(getattr(config.exp, 'cached_features', False))
    # precision: bf16 compute copies + fp32 master weights on the GPU (grit_amd/amp.py; gradients are produced and
    # all-reduced in flat bf16 buckets), plain fp32 with exp.bf16=False or on CPU
    # exp.grad_sync = 'shard' (or GRIT_GRAD_SYNC=shard): reduce-scatter + sharded FlatAdam + all-gather instead of all-reduce
    grad_sync = getattr(config.exp, 'grad_sync', os.environ.get('GRIT_GRAD_SYNC', 'allreduce'))
    if getattr(config.exp, 'bf16', use_cuda):
        model = Bf16Compute(model, shard_optimizer=(grad_sync == 'shard' and use_cuda and world > 1))
    else:
        model = BucketedDataParallel(model)
    optimizers = build_optimizers(model, config, mode='xe')
    resume = getattr(config.exp, 'resume_from', '')
    if resume:  # reference-format checkpoint (engine/caption_engine.py save_checkpoint): weights (+ optimizer state when present)
        ckpt = torch.load(resume, map_location='cpu')
        raw.load_state_dict(ckpt['state_dict'], strict=False)  # reaches the fp32 masters through the wrapper's hook
        for k in ('model', 'backbone'):
            if ckpt.get('optim_' + k) and getattr(config.exp, 'resume_optimizers', True):
                try:
                    optimizers[k].load_state_dict(ckpt['optim_' + k])
                except (ValueError, KeyError) as e:  # a checkpoint of another phase / parameter grouping
                    print(f"optimizer state of '{k}' not restored: {e}")
        start_epoch = max(start_epoch, int(ckpt.get('epoch', -1)) + 1)

    if dataloaders is None:
        steps = getattr(config.exp, 'synthetic_steps', 20)
        h, w = getattr(config.exp, 'synthetic_size', [640, 640])
        dataloaders = {'train': SyntheticLoader(steps, opt.batch_size, h, w, device=device, rank=rank)}
    xe_epochs = opt.freezing_xe_epochs + opt.finetune_xe_epochs
    scheduler = CosineLRScheduler(optimizers['model'], num_epochs=xe_epochs, num_its_per_epoch=len(dataloaders['train']),
                                  init_lr=opt.xe_lr, min_lr=opt.min_lr, warmup_init_lr=opt.warmup_init_lr)
    total = xe_epochs + opt.freezing_sc_epochs + opt.finetune_sc_epochs
    can_sc = 'train_dict' in dataloaders and text_field is not None and cider is not None
    save = getattr(config.exp, 'save', False)
    results = []
    for epoch in range(start_epoch, min(total, getattr(config.exp, 'max_epochs', total))):
        phase = phase_of(epoch, opt)
        if phase in ('fr_sc', 'ft_sc') and not can_sc:
            continue  # synthetic run: no captions to score
        if phase in ('ft_xe', 'ft_sc') and raw.cached_features:
            raw.cached_features = False  # from here on the detector is part of the graph (the wrapper re-derives its live set)
            if finetune_dataloaders is not None:
                dataloaders = finetune_dataloaders
        if phase in ('fr_sc', 'ft_sc') and optimizers['mode'] == 'xe':
            optimizers = build_optimizers(model, config, mode='sc')
        if phase in ('fr_xe', 'ft_xe') and optimizers['mode'] == 'sc':
            optimizers = build_optimizers(model, config, mode='xe')
        print(f"Train: rank={rank}, epoch={epoch}, phase={phase}")
        if phase in ('fr_xe', 'ft_xe'):
            res = train_xe(model, dataloaders, optimizers=optimizers, text_field=text_field, epoch=epoch, rank=rank,
                           config=config, scheduler=scheduler, writer=None, checkpoint=save)
        else:
            best = getattr(config.exp, 'best_checkpoint', 'checkpoint_best_valid.pth')
            if os.path.exists(best):  # reference :131-132
                missing, unexpected = raw.load_state_dict(torch.load(best, map_location='cpu')['state_dict'], strict=False)
                print(f"Start self-critical optimization: missing={len(missing)}, unexpected={len(unexpected)}")
            res = train_sc(model, dataloaders, optimizers=optimizers, cider=cider, text_field=text_field,
                           tokenizer_pool=tokenizer_pool, device=device, epoch=epoch, config=config, rank=rank, writer=None,
                           evaluate='valid' in dataloaders, checkpoint=save)
        results.append(res)
        if rank == 0 and save:
            save_checkpoint(model, optimizers, epoch=epoch, scores=[], best_ciders=[0, 0], config=config,
                            filename=f'checkpoint_{phase}.pth', scheduler=scheduler)
        dist.barrier()
    if dist.is_initialized() and getattr(config.exp, 'destroy_group', True):
        dist.destroy_process_group()
    return results


def _parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default=None)
    ap.add_argument('--gpus', type=int, default=None)
    ap.add_argument('overrides', nargs='*')
    a = ap.parse_args()
    over = {}
    for kv in a.overrides:
        k, v = kv.split('=', 1)
        try:
            import ast
            v = ast.literal_eval(v)
        except Exception:
            pass
        over[k] = v
    cfg = load_yaml(a.config) if a.config else default_config(**over)
    return cfg, a.gpus


def run_main():
    config, gpus = _parse()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "6688")
    if 'LOCAL_RANK' in os.environ:  # launched by torchrun: one process already exists per GPU
        main(int(os.environ['LOCAL_RANK']), config)
        return
    n = gpus or max(1, torch.cuda.device_count())
    config.exp.ngpus_per_node = n
    config.exp.world_size = n
    mp.spawn(main, nprocs=n, args=(config,))


if __name__ == "__main__":
    run_main()
