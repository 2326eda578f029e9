"""Batch contract of the captioning hot path + the synthetic generator used by bench/tests.

What `train_xe` feeds the model (reference datasets/caption/coco.py:56-81 PairedCollator, engine/utils.py:278-295):
    batch['samples']  = NestedTensor(tensors [B,3,H,W] float32 (ImageNet-normalised), mask [B,H,W] bool, True on pad)
    batch['captions'] = int64 [B,T] = [BOS=2] + tokens + [EOS=3] + PAD=1..., already on the device.
The COCO reader itself (tokenisation, RandAugment, hdf5 caches) is out of scope: throughput is defined on
synthetic batches of exactly this shape.
"""
import torch

from grit_amd.utils.misc import NestedTensor

PAD, BOS, EOS = 1, 2, 3


def synthetic_batch(batch_size, height=640, width=640, caption_len=20, vocab_size=10201, device='cpu', seed=0,
                    ragged=False):
    """images ~ N(0,1); captions = [BOS] + (T-2) tokens ~ U{4..V-1} + [EOS]; ragged=True varies image sizes and
    caption lengths so that masks / PAD tokens are exercised."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(batch_size, 3, height, width, generator=g)
    mask = torch.zeros(batch_size, height, width, dtype=torch.bool)
    caps = torch.randint(4, vocab_size, (batch_size, caption_len), generator=g)
    caps[:, 0] = BOS
    caps[:, -1] = EOS
    if ragged:
        for b in range(1, batch_size):
            h = height - 32 * (b % 3)
            w = width - 32 * ((b + 1) % 4)
            images[b, :, h:, :] = 0
            images[b, :, :, w:] = 0
            mask[b, h:, :] = True
            mask[b, :, w:] = True
            n = max(3, caption_len - (b % 5))
            caps[b, n - 1] = EOS
            caps[b, n:] = PAD
    return {'samples': NestedTensor(images, mask, any_padding=bool(ragged and batch_size > 1)).to(device),
            'captions': caps.to(device)}


class SyntheticLoader(object):
    """len()-able iterable of identical-shape synthetic batches, one seed per (rank, step)."""

    def __init__(self, steps, batch_size, height=640, width=640, caption_len=20, vocab_size=10201, device='cpu',
                 rank=0, pregenerate=True):
        self.steps, self.kw = steps, dict(batch_size=batch_size, height=height, width=width, caption_len=caption_len,
                                          vocab_size=vocab_size, device=device)
        self.rank = rank
        self.cache = [synthetic_batch(seed=1000 * rank + i, **self.kw) for i in range(min(steps, 4))] if pregenerate else None
        self.dataset = self

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.cache[i % len(self.cache)] if self.cache else synthetic_batch(seed=1000 * self.rank + i, **self.kw)
