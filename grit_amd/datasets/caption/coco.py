"""Collators of the captioning datasets (reference datasets/caption/coco.py:26-81).

Same batch dictionary as the reference: `samples` (NestedTensor on the device, or the cached-feature dictionary),
`captions`, `image_id`.  Images arrive as the `Deferred` records of `get_transform`; resize, ToTensor, Normalize, padding
and mask run as one device operation per batch (grit_amd/ops/image_batch.py).  The COCO readers themselves (pycocotools,
spacy tokeniser, hdf5) are out of scope."""
import torch

from .transforms import collate_images


class DictionaryCollator(object):

    def __init__(self, img_field=None, device='cuda'):
        self.img_field = img_field
        self.device = device

    def __call__(self, batch):
        imgs = [item[0] for item in batch]
        outputs = {'captions': [item[1] for item in batch], 'image_id': [item[2] for item in batch]}
        if getattr(self.img_field, 'use_hdf5_feat', False):  # cached detector features, coco.py:39-47
            samples = {}
            for feat, flag in (('gri', 'use_gri_feat'), ('reg', 'use_reg_feat')):
                if getattr(self.img_field, flag, False):
                    for key in (feat + '_feat', feat + '_mask'):
                        samples[key] = torch.stack([im[key] for im in imgs]).to(self.device, non_blocking=True)
            outputs['samples'] = samples
        else:
            outputs['samples'] = collate_images(imgs, self.device)
        return outputs


class PairedCollator(DictionaryCollator):
    """+ captions as one int64 [B, T] tensor: [bos] + tokens + [eos] + pad..., T = longest caption + 2."""

    def __init__(self, img_field=None, device='cuda', max_len=54, pad_idx=1, bos_idx=2, eos_idx=3):
        super().__init__(img_field, device)
        self.max_len, self.pad_idx, self.bos_idx, self.eos_idx = max_len, pad_idx, bos_idx, eos_idx

    def __call__(self, batch):
        b = super().__call__(batch)
        # the reference pads to the longest *untruncated* caption (coco.py:66-67); kept
        longest = max(len(c) for c in b['captions'])
        rows = torch.full((len(b['captions']), longest + 2), self.pad_idx, dtype=torch.int64)
        for i, c in enumerate(b['captions']):
            c = list(c[:self.max_len])
            rows[i, :len(c) + 2] = torch.tensor([self.bos_idx] + c + [self.eos_idx], dtype=torch.int64)
        b['captions'] = rows.pin_memory().to(self.device, non_blocking=True) if torch.device(self.device).type == 'cuda' \
            else rows
        return b
