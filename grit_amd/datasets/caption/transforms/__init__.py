"""`get_transform(cfg)` of the captioning datasets (reference datasets/caption/transforms/__init__.py:6-32).

train / valid transforms = resize policy only: ToTensor and Normalize are applied on the device together with the
resampling and the batch padding (`collate_images`), so a transform returns a `Deferred` image, not a tensor.
RandAugment (`cfg.randaug`) is host-side PIL augmentation and not part of this build (out of scope)."""
from grit_amd.ops.image_batch import MEAN, STD, image_batch
from grit_amd.utils.misc import NestedTensor

from .utils import Deferred, FixedResize, MaxWHResize, MinMaxResize

RESIZE = {'minmax': MinMaxResize, 'maxwh': MaxWHResize}


def get_transform(cfg):
    if getattr(cfg, 'randaug', False):
        raise NotImplementedError("RandAugment is host-side PIL augmentation; not provided by the device pipeline "
                                  "(grit_amd.config.default_config sets transform_cfg.randaug = False for that reason)")
    if cfg.resize_name == 'normal':
        # the reference's 'normal' policy is torchvision Resize = PIL BILINEAR; the device resampler implements Pillow's
        # BICUBIC taps only, and silently resampling with another filter would change the input pixels
        raise NotImplementedError("resize_name='normal' (bilinear torchvision Resize) is not provided; use 'maxwh' or 'minmax'")
    resize = RESIZE[cfg.resize_name](cfg.size)
    return {'train': resize, 'valid': resize}


def collate_images(items, device=None, pad_to=None):
    """List of `Deferred` (what the transforms return) -> NestedTensor on the device; the device counterpart of
    Compose([resize, ToTensor(), normalize()]) per image + nested_tensor_from_tensor_list(imgs).to(device)."""
    for it in items:
        if not isinstance(it, Deferred):
            raise TypeError("collate_images expects the Deferred images returned by get_transform()'s transforms")
    sizes = [it.size for it in items]
    tensors, mask = image_batch([it.pixels for it in items], sizes, MEAN, STD, device, pad_to)
    padded = len(set(sizes)) > 1 or (pad_to is not None and tuple(pad_to) != tuple(sizes[0]))
    return NestedTensor(tensors, mask, any_padding=padded)
