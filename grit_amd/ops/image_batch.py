"""Decoded RGB images -> the model's input batch, on the device.

    [h_i, w_i, 3] uint8  --bicubic resize to (oh_i, ow_i), Pillow's 8-bit arithmetic-->  --/255, (x - mean) / std-->
    --zero padding to (max oh, max ow)-->  tensors [B, 3, H, W] f32,  mask [B, H, W] bool (True on padding)

Replaces the per-image host chain of the reference -- PIL `x.resize(..., Image.BICUBIC)` in MaxWHResize / MinMaxResize
(datasets/caption/transforms/utils.py:4-45), ToTensor + Normalize (transforms/__init__.py:6-32) and
nested_tensor_from_tensor_list (engine/utils.py:278-295) -- by one pinned upload and two launches per batch
(grit_image_batch_fwd).  The uint8 stage is bit-identical to Pillow; the float stage is a lookup in a 3 x 256 table that
is built here with the same torch ops ToTensor / Normalize apply (`v.float().div(255)`, `.sub(mean).div(std)`).

The tap tables depend only on (source size, target size) per axis; they are computed by the library's host function
grit_resample_taps_bicubic and cached."""
import ctypes
import functools

import numpy as np
import torch

from grit_amd import lib as _lib

MEAN = (0.485, 0.456, 0.406)  # transforms/__init__.py:6-7
STD = (0.229, 0.224, 0.225)
DESC_FIELDS = 12  # GRIT_IMAGE_DESC_FIELDS
SRC_PAD = 64  # GRIT_IMAGE_SRC_PAD


def tmp_pitch(dst_w):
    return (3 * dst_w + 3) & ~3  # GRIT_IMAGE_TMP_PITCH


@functools.lru_cache(maxsize=4096)
def axis_taps(in_size, out_size):
    """-> (ksize, bounds int32 [out_size, 2], taps int32 [out_size, ksize]) of one axis (host, cached)."""
    lib = _lib.load()
    ksize = lib.grit_resample_taps_bicubic(in_size, out_size, None, None, 0)
    if ksize <= 0:
        raise _lib.GritHipError("grit_resample_taps_bicubic(%d, %d) failed" % (in_size, out_size))
    bounds = np.empty((out_size, 2), np.int32)
    taps = np.empty((out_size, ksize), np.int32)
    got = lib.grit_resample_taps_bicubic(in_size, out_size, bounds.ctypes.data_as(ctypes.c_void_p),
                                         taps.ctypes.data_as(ctypes.c_void_p), taps.size)
    if got != ksize:
        raise _lib.GritHipError("grit_resample_taps_bicubic(%d, %d) failed" % (in_size, out_size))
    return ksize, bounds, taps


@functools.lru_cache(maxsize=16)
def _lut(mean, std, device):
    v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)  # ToTensor
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None]
    return v[None, :].sub(m).div(s).contiguous().to(device)  # Normalize


class _Staging(object):
    """Grow-only pinned buffers handed out round-robin; a slot is reused only after the upload that read it finished."""

    def __init__(self, slots=3):
        self.slots = [[None, None] for _ in range(slots)]  # (pinned uint8 buffer, event of its last upload)
        self.turn = 0

    def take(self, nbytes):
        slot = self.slots[self.turn]
        self.turn = (self.turn + 1) % len(self.slots)
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < nbytes:
            with torch.inference_mode(False):  # the buffer outlives the caller's inference_mode block
                slot[0] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
        return slot

    def upload(self, slot, nbytes, device):
        dev = slot[0][:nbytes].to(device, non_blocking=True)
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record()
        return dev


_pixels_staging, _tables_staging = _Staging(), _Staging()


def _as_u8(img):
    if isinstance(img, torch.Tensor):
        t = img
    else:
        arr = np.ascontiguousarray(np.asarray(img))  # numpy array or PIL image (decoded, RGB)
        t = torch.from_numpy(arr if arr.flags.writeable else arr.copy())  # PIL hands out read-only views
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("expected decoded RGB images as [h, w, 3] uint8, got %s %s" % (tuple(t.shape), t.dtype))
    return t.contiguous()


def plan(shapes, sizes):
    """Host side of one batch: -> (desc int64 [B, 12], tables int32 [n], tmp_bytes, src_bytes)."""
    desc = np.empty((len(shapes), DESC_FIELDS), np.int64)
    parts, n, src_off, tmp_off = [], 0, 0, 0
    placed = {}

    def place(key, arr):
        nonlocal n
        if key not in placed:
            placed[key] = n
            parts.append(arr.reshape(-1))
            n += arr.size
            if n % 2:  # the kernels read (first, count) pairs as one 8-byte load
                parts.append(np.zeros(1, np.int32))
                n += 1
        return placed[key]

    for i, ((h, w), (oh, ow)) in enumerate(zip(shapes, sizes)):
        if min(h, w, oh, ow) <= 0:
            raise ValueError("empty image or target size: %s -> %s" % ((h, w), (oh, ow)))
        kx, xb, xt = axis_taps(w, ow)
        ky, yb, yt = axis_taps(h, oh)
        desc[i] = (src_off, h, w, oh, ow, kx, ky, place(('b', w, ow), xb), place(('t', w, ow), xt),
                   place(('b', h, oh), yb), place(('t', h, oh), yt), tmp_off)
        src_off += h * w * 3
        tmp_off += h * tmp_pitch(ow)
    return desc, np.concatenate(parts), tmp_off, src_off


def image_batch(images, sizes, mean=MEAN, std=STD, device=None, pad_to=None):
    """images: list of [h, w, 3] uint8 (torch / numpy / PIL, host or device); sizes: list of (oh, ow).

    -> (tensors [B, 3, H, W] f32, mask [B, H, W] bool) on `device`, H = max oh, W = max ow -- or `pad_to` = (H, W), the
    fixed canvas the feature extractor needs (the reference gets it by appending a dummy H x W image to every batch,
    tools/extract_features.py:103)."""
    imgs = [_as_u8(im) for im in images]
    if len(imgs) == 0 or len(imgs) != len(sizes):
        raise ValueError("need one target size per image and at least one image")
    if device is None:
        device = imgs[0].device if imgs[0].is_cuda else torch.device('cuda', torch.cuda.current_device())
    device = torch.device(device)
    if device.type != 'cuda':
        raise _lib.GritHipError("Not implemented on the CPU: the image pipeline runs as HIP kernels")
    lib = _lib.load()
    desc, tables, tmp_bytes, src_bytes = plan([tuple(im.shape[:2]) for im in imgs], [tuple(s) for s in sizes])
    H, W = int(desc[:, 3].max()), int(desc[:, 4].max())
    max_dst_w = W
    if pad_to is not None:
        if pad_to[0] < H or pad_to[1] < W:
            raise ValueError("pad_to %s is smaller than the largest resized image (%d, %d)" % (tuple(pad_to), H, W))
        H, W = int(pad_to[0]), int(pad_to[1])
    B = len(imgs)
    with _lib.device_guard(device):
        if all(im.is_cuda for im in imgs):
            src = torch.cat([im.reshape(-1) for im in imgs] + [torch.zeros(SRC_PAD, dtype=torch.uint8, device=device)])
        elif all(im.is_cuda or im.is_pinned() for im in imgs):  # decoder wrote into pinned memory: no staging copy
            src = torch.empty(src_bytes + SRC_PAD, dtype=torch.uint8, device=device)
            for im, off in zip(imgs, desc[:, 0].tolist()):
                src[off:off + im.numel()].copy_(im.reshape(-1), non_blocking=True)
        else:  # one pinned staging blob, one asynchronous upload
            slot = _pixels_staging.take(src_bytes + SRC_PAD)
            for im, off in zip(imgs, desc[:, 0].tolist()):
                slot[0][off:off + im.numel()].copy_(im.reshape(-1))
            src = _pixels_staging.upload(slot, src_bytes + SRC_PAD, device)
        # descriptor (int64) and tables (int32) travel in one pinned blob
        nd, nt = 8 * desc.size, 4 * tables.size
        slot = _tables_staging.take(nd + nt)
        slot[0][:nd].copy_(torch.from_numpy(desc.reshape(-1)).view(torch.uint8))
        slot[0][nd:nd + nt].copy_(torch.from_numpy(tables).view(torch.uint8))
        dev = _tables_staging.upload(slot, nd + nt, device)
        d_desc, d_tables = dev[:nd], dev[nd:]
        tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=device)
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=device)
        mask = torch.empty(B, H, W, dtype=torch.bool, device=device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        st = lib.grit_image_batch_fwd(p(src), p(d_desc), p(d_tables), p(tmp), p(_lut(tuple(mean), tuple(std), device)),
                                      B, int(desc[:, 1].max()), max_dst_w, int(desc[:, 5].max()), H, W, p(out), p(mask), _lib.current_stream_ptr())
        _lib.check(st, "grit_image_batch_fwd")
    return out, mask
