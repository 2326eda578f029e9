"""Batch contract + small helpers of the hot path.

Mirrors, by name and behaviour: NestedTensor / nested_tensor_from_tensor_list (reference
engine/utils.py:250-295, utils/misc.py:323-368) and inverse_sigmoid (utils/misc.py:516-520), plus the
torch.distributed helpers the engine needs.  The logging utilities of the reference file
(MetricLogger, SmoothedValue, ...) are out of scope.
"""
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class NestedTensor(object):
    """A batch of images zero-padded to a common size + a bool mask that is True on the padding."""

    def __init__(self, tensors, mask: Optional[Tensor], any_padding: Optional[bool] = None):
        self.tensors = tensors
        self.mask = mask
        # host-side knowledge about the mask: False = the builder knows no pixel is padding (all images the same
        # size), True = some are, None = unknown.  Lets the detector skip the no-op masked_fill of every value map
        # without a device->host sync; the result is identical either way.
        self.any_padding = any_padding

    def to(self, device, non_blocking=False):
        mask = None if self.mask is None else self.mask.to(device, non_blocking=non_blocking)
        return NestedTensor(self.tensors.to(device, non_blocking=non_blocking), mask, self.any_padding)

    def record_stream(self, *args, **kwargs):
        self.tensors.record_stream(*args, **kwargs)
        if self.mask is not None:
            self.mask.record_stream(*args, **kwargs)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]):
    """[C,H_i,W_i] images -> NestedTensor([B,C,maxH,maxW], mask[B,maxH,maxW]); top-left aligned."""
    if tensor_list[0].ndim != 3:
        raise ValueError('not supported')
    c = tensor_list[0].shape[0]
    h = max(int(t.shape[1]) for t in tensor_list)
    w = max(int(t.shape[2]) for t in tensor_list)
    ref = tensor_list[0]
    batch = torch.zeros((len(tensor_list), c, h, w), dtype=ref.dtype, device=ref.device)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=ref.device)
    for i, img in enumerate(tensor_list):
        batch[i, :img.shape[0], :img.shape[1], :img.shape[2]].copy_(img)
        mask[i, :img.shape[1], :img.shape[2]] = False
    return NestedTensor(batch, mask, any_padding=any(int(t.shape[1]) != h or int(t.shape[2]) != w for t in tensor_list))


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0
