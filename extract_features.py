"""Cache the detector's outputs for the decoder-only training mode (reference tools/extract_features.py:48-155).

For every image: resize policy of the dataset config -> device batch padded to the fixed canvas (H, W) -> frozen
detector -> one row per dataset of the feature store (`gri_feat`, `gri_mask`, `reg_feat`, `reg_mask`, `image_ids`;
grit_amd/datasets/caption/feature_store.py).  Differences from the reference, none of which changes a stored value:

* the canvas comes from `pad_to=(H, W)` of the device collator, not from a dummy random image appended to each batch;
* ranks write their rows (`rank::world_size`, the order DistributedSampler(shuffle=False) deals) straight into the shared
  memory-mapped files -- no per-rank files, no merge on rank 0;
* the resize + normalise + pad of a batch is one device operation (grit_amd/ops/image_batch.py).

    python extract_features.py --img-root DIR --out DIR [--checkpoint ckpt.pth]      (one process per GPU under torchrun)
"""
import argparse
import os

import numpy as np
import torch
import torch.distributed as dist

from grit_amd.config import default_config
from grit_amd.datasets.caption.feature_store import FeatureStore
from grit_amd.datasets.caption.transforms import collate_images, get_transform


def canvas_size(transform_cfg):
    """tools/extract_features.py:57-62."""
    if transform_cfg.resize_name in ('normal', 'maxwh'):
        return int(transform_cfg.size[0]), int(transform_cfg.size[1])
    if transform_cfg.resize_name == 'minmax':
        return int(transform_cfg.size[1]), int(transform_cfg.size[1])
    raise ValueError(transform_cfg.resize_name)


def grid_tokens(H, W):
    """Tokens of the coarsest Swin map (gri_feat) for an H x W canvas: PatchEmbed pads to a multiple of 4, each of the four
    PatchMerging steps pads odd sizes (ceil), models/common/swin_model.py:324-349, 482-499 -- 1333 -> 21, not 1333 // 64 = 20."""
    def down(n):
        n = -(-n // 4)
        for _ in range(4):
            n = (n + 1) // 2
        return n
    return down(H) * down(W)


class ImageFolder(object):
    """Every image file under `root`, id = the number after the last underscore of the file name (COCO naming)."""

    def __init__(self, root):
        names = sorted(n for n in os.listdir(root) if n.lower().endswith(('.jpg', '.jpeg', '.png')))
        self.root, self.names = root, names
        self.img_ids = [int(n.split('_')[-1].split('.')[0]) for n in names]

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        from PIL import Image
        return np.asarray(Image.open(os.path.join(self.root, self.names[i])).convert('RGB'))


@torch.inference_mode()
def extract_vis_features(detector, images, image_ids, config, out_path, device='cuda', rank=0, world_size=1, batch_size=64):
    """images: indexable of decoded RGB uint8 arrays [h, w, 3]; image_ids: their ids.  Returns the FeatureStore."""
    detector = detector.eval()
    policy = get_transform(config.dataset.transform_cfg)['valid']
    H, W = canvas_size(config.dataset.transform_cfg)
    tokens = grid_tokens(H, W)
    synced = dist.is_available() and dist.is_initialized() and world_size > 1
    det = config.model.detector
    queries, d_model = (det.num_queries, det.d_model) if config.model.use_reg_feat else (None, None)
    if rank == 0:
        FeatureStore.create(out_path, image_ids, tokens, config.model.grid_feat_dim, queries, d_model)
    if synced:
        dist.barrier()
    store = FeatureStore.open(out_path, mode='r+')
    mine = list(range(rank, len(image_ids), world_size))
    for at in range(0, len(mine), batch_size):
        rows = mine[at:at + batch_size]
        samples = collate_images([policy(images[i]) for i in rows], device, pad_to=(H, W))
        out = detector(samples)
        for name in store.arrays:
            value = out[name] if out[name].dtype == torch.bool else out[name].float()
            store[name][rows] = value.cpu().numpy()
    store.flush()
    if synced:
        dist.barrier()
    return FeatureStore.open(out_path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--img-root', required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--checkpoint', default='')
    ap.add_argument('--batch-size', type=int, default=64)
    a = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    if world > 1:
        dist.init_process_group('nccl')
    config = default_config()
    config.dataset.transform_cfg.randaug = False
    from inference_caption import build_model
    model = build_model(config, 'cuda', a.checkpoint)
    data = ImageFolder(a.img_root)
    store = extract_vis_features(model.detector, data, data.img_ids, config, a.out, 'cuda', rank, world, a.batch_size)
    if rank == 0:
        print('wrote', {k: tuple(v.shape) for k, v in store.arrays.items()}, 'to', a.out)


if __name__ == '__main__':
    main()
