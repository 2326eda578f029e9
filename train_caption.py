"""Entry point with the reference's shape (train_caption.py:24-209): `main(gpu, config)` per process, one process per
GPU, torch.distributed 'nccl' (= RCCL on ROCm) over xGMI.

    python train_caption.py [--config my.yaml] [--gpus N] [--synthetic-steps K] [key=value ...]
    torchrun --nproc-per-node N train_caption.py ...      (RANK/LOCAL_RANK/WORLD_SIZE from the environment)

hydra is not installed in this image; configuration is grit_amd.config (same keys as configs/caption/coco_config.yaml,
dotted key=value overrides).  The COCO reader is out of scope of this build: batches come from
grit_amd.data.SyntheticLoader unless the caller passes its own `dataloaders` dict to main().
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
from engine.caption_engine import *  # noqa: F401,F403  (reference does the same star import)
from engine.caption_engine import build_optimizers, save_checkpoint, train_xe
from models.caption import Transformer
from models.caption.detector import build_detector
from utils.cap_scheduler import CosineLRScheduler


def main(gpu, config, dataloaders=None):
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

    detector = build_detector(config).to(device)
    model = Transformer(detector=detector, config=config).to(device)
    if getattr(config.optimizer, 'freeze_backbone', False):
        for n, p in model.named_parameters():
            if 'backbone' in n:
                p.requires_grad = False
    if getattr(config.optimizer, 'freeze_detector', False):
        for n, p in model.named_parameters():
            if 'detector' in n:
                p.requires_grad = False
    model.cached_features = False
    # precision: bf16 compute copies + fp32 master weights on the GPU (grit_amd/amp.py; gradients are produced and
    # all-reduced in flat bf16 buckets), plain fp32 with exp.bf16=False or on CPU
    if getattr(config.exp, 'bf16', use_cuda):
        model = Bf16Compute(model)
    else:
        model = BucketedDataParallel(model)
    optimizers = build_optimizers(model, config, mode='xe')

    if dataloaders is None:
        steps = getattr(config.exp, 'synthetic_steps', 20)
        h, w = getattr(config.exp, 'synthetic_size', [640, 640])
        dataloaders = {'train': SyntheticLoader(steps, config.optimizer.batch_size, h, w, device=device, rank=rank)}
    epochs = config.optimizer.freezing_xe_epochs + config.optimizer.finetune_xe_epochs
    scheduler = CosineLRScheduler(optimizers['model'], num_epochs=epochs, num_its_per_epoch=len(dataloaders['train']),
                                  init_lr=config.optimizer.xe_lr, min_lr=config.optimizer.min_lr,
                                  warmup_init_lr=config.optimizer.warmup_init_lr)
    results = []
    for epoch in range(getattr(config.exp, 'max_epochs', epochs)):
        print(f"Train: rank={rank}, epoch={epoch}, phase=ft_xe")
        res = train_xe(model, dataloaders, optimizers=optimizers, text_field=None, epoch=epoch, rank=rank, config=config,
                       scheduler=scheduler, writer=None, checkpoint=getattr(config.exp, 'save', False))
        results.append(res)
        if rank == 0 and getattr(config.exp, 'save', False):
            save_checkpoint(model, optimizers, epoch=epoch, scores=[], best_ciders=[0, 0], config=config,
                            filename='checkpoint_ft_xe.pth', scheduler=scheduler)
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
