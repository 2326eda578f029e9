"""Resize policies of the captioning datasets (reference datasets/caption/transforms/utils.py:4-45).

The reference resizes each PIL image on the host inside the dataset worker.  Here a policy only *decides* the target
size (`output_size`, same float arithmetic as the reference's `__call__`); calling it on a decoded image returns a
`Deferred` record and the resampling itself happens for the whole batch on the device (grit_amd/ops/image_batch.py)."""
import numpy as np


class Deferred(object):
    """A decoded RGB image ([h, w, 3] uint8) waiting for the device-side resize to `size` = (newh, neww)."""
    __slots__ = ('pixels', 'size')

    def __init__(self, pixels, size):
        self.pixels, self.size = pixels, size


def _pixels(x):
    if isinstance(x, Deferred):
        raise ValueError("image already carries a resize request")
    if hasattr(x, 'shape'):  # torch / numpy, [h, w, 3]
        return x
    return np.asarray(x)  # PIL image -> [h, w, 3] uint8 view of the decoded pixels


class _Policy(object):

    def __call__(self, x):
        pixels = _pixels(x)
        return Deferred(pixels, self.output_size(int(pixels.shape[0]), int(pixels.shape[1])))


class MaxWHResize(_Policy):
    """Largest size of the same aspect ratio inside (max_h, max_w)."""

    def __init__(self, size):
        self.size = size
        self.max_h, self.max_w = size[0], size[1]

    def output_size(self, h, w):
        scale = min(self.max_w / w, self.max_h / h)
        return int(h * scale), int(w * scale)


class MinMaxResize(_Policy):
    """Short side to `min` unless the long side would exceed `max`; both sides rounded down to multiples of 32."""

    def __init__(self, size):
        self.size = size
        self.min, self.max = size[0], size[1]

    def output_size(self, h, w):
        scale = self.min / min(w, h)
        newh, neww = (self.min, scale * w) if h < w else (scale * h, self.min)
        longest = max(newh, neww)
        if longest > self.max:
            shrink = self.max / longest
            newh, neww = newh * shrink, neww * shrink
        newh, neww = int(newh + 0.5), int(neww + 0.5)
        return newh // 32 * 32, neww // 32 * 32


class FixedResize(_Policy):
    """torchvision.transforms.Resize((h, w)) of the 'normal' entry: always the configured size."""

    def __init__(self, size):
        self.size = tuple(size)

    def output_size(self, h, w):
        return self.size
