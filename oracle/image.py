"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  ctypes front-end of oracle/image_oracle.c (numpy in / out).

The image side of the batch contract: resize (datasets/caption/transforms/utils.py:4-45) -> ToTensor -> Normalize
(datasets/caption/transforms/__init__.py:6-32) -> zero-padded batch + mask (engine/utils.py:278-295).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libimage_oracle.so")
_lib = None

MEAN = (0.485, 0.456, 0.406)  # transforms/__init__.py:6-7
STD = (0.229, 0.224, 0.225)


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "image_oracle.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-std=c99", "-o", _LIB, src, "-lm"])
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def maxwh_size(h, w, size):
    """transforms/utils.py:11-16: size = (max_h, max_w) -> (newh, neww)."""
    scale = min(size[1] / w, size[0] / h)
    return int(h * scale), int(w * scale)


def minmax_size(h, w, size):
    """transforms/utils.py:26-45: size = (min, max) -> (newh, neww), multiples of 32."""
    lo, hi = size
    scale = lo / min(w, h)
    if h < w:
        newh, neww = lo, scale * w
    else:
        newh, neww = scale * h, lo
    if max(newh, neww) > hi:
        scale = hi / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    newh, neww = int(newh + 0.5), int(neww + 0.5)
    return newh // 32 * 32, neww // 32 * 32


def resize_bicubic(img, oh, ow):
    """img [h, w, 3] uint8 -> [oh, ow, 3] uint8, Pillow's Image.resize((ow, oh), Image.BICUBIC)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    assert c == 3
    if (oh, ow) == (h, w):
        return img.copy()
    out = np.empty((oh, ow, 3), np.uint8)
    _load().oracle_resize_bicubic_rgb(_p(img), h, w, _p(out), oh, ow)
    return out


def image_batch(images, sizes):
    """images: list of [h, w, 3] uint8; sizes: list of (oh, ow).  -> (tensors [B,3,H,W] f32, mask [B,H,W] bool)."""
    H = max(s[0] for s in sizes)
    W = max(s[1] for s in sizes)
    out = np.empty((len(images), 3, H, W), np.float32)
    mask = np.empty((len(images), H, W), np.uint8)
    mean = np.asarray(MEAN, np.float32)
    std = np.asarray(STD, np.float32)
    for i, (img, (oh, ow)) in enumerate(zip(images, sizes)):
        small = resize_bicubic(img, oh, ow)
        _load().oracle_to_padded_slot(_p(small), oh, ow, _p(mean), _p(std), _p(out[i]), _p(mask[i]), H, W)
    return out, mask.astype(bool)
