"""Image side of the batch contract on the CPU: the C oracle (oracle/image_oracle.c) against fixture G11 (made from the
reference's resize classes, tests/golden/make_golden.py) and against Pillow itself; the library's host-side tap
tables; the resize policies and collators of grit_amd.datasets (host logic only -- no kernel is launched here)."""
import os

import numpy as np
import pytest
import torch

from grit_amd.datasets.caption.transforms.utils import Deferred, MaxWHResize, MinMaxResize
from grit_amd.ops import image_batch as ib
from oracle import image as oimg

POLICIES = {'maxwh': (MaxWHResize((48, 64)), lambda h, w: oimg.maxwh_size(h, w, (48, 64))),
            'minmax': (MinMaxResize((64, 128)), lambda h, w: oimg.minmax_size(h, w, (64, 128)))}


@pytest.fixture(scope="module")
def g11(golden_dir):
    return np.load(os.path.join(golden_dir, "image_g11.npz"))


@pytest.mark.parametrize("name", ["maxwh", "minmax"])
def test_oracle_reproduces_reference_batch_bit_exact(g11, name):
    images = [g11['%s_in%d' % (name, i)] for i in range(5)]
    sizes = [tuple(int(v) for v in s) for s in g11[name + '_sizes']]
    for i, img in enumerate(images):  # the uint8 stage alone
        np.testing.assert_array_equal(oimg.resize_bicubic(img, *sizes[i]), g11['%s_u8_%d' % (name, i)])
    tensors, mask = oimg.image_batch(images, sizes)
    np.testing.assert_array_equal(tensors, g11[name + '_tensors'])
    np.testing.assert_array_equal(mask, g11[name + '_mask'])


@pytest.mark.parametrize("name", ["maxwh", "minmax"])
def test_resize_policies_choose_the_reference_sizes(g11, name):
    policy, oracle_size = POLICIES[name]
    for i in range(5):
        img = g11['%s_in%d' % (name, i)]
        want = tuple(int(v) for v in g11[name + '_sizes'][i])
        assert policy.output_size(*img.shape[:2]) == want == oracle_size(*img.shape[:2])
        d = policy(img)
        assert isinstance(d, Deferred) and d.size == want and d.pixels is img
    # COCO-like sizes through the float arithmetic of the reference (int() truncation, //32 rounding)
    assert MaxWHResize((384, 640)).output_size(480, 640) == (384, 512)
    assert MaxWHResize((384, 640)).output_size(427, 640) == (384, 575)
    assert MinMaxResize((384, 640)).output_size(480, 640) == (384, 512)
    assert MinMaxResize((384, 640)).output_size(333, 500) == (384, 576)


def test_oracle_equals_pillow_on_seeded_images():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for (h, w, oh, ow) in [(480, 640, 384, 512), (100, 80, 224, 179), (37, 53, 37, 90), (64, 64, 64, 64), (600, 800, 96, 128),
                           (33, 47, 200, 31), (5, 7, 20, 3), (1, 9, 4, 4)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img, 'RGB').resize((ow, oh), resample=Image.BICUBIC))
        np.testing.assert_array_equal(oimg.resize_bicubic(img, oh, ow), want)


def _resize_with_tables(img, oh, ow):
    """numpy evaluation of the two passes with the LIBRARY's tap tables (host function of libgrit_hip.so)."""
    h, w, _ = img.shape
    _, xb, xt = ib.axis_taps(w, ow)
    _, yb, yt = ib.axis_taps(h, oh)
    tmp = np.empty((h, ow, 3), np.uint8)
    for xx in range(ow):
        f, c = xb[xx]
        s = (img[:, f:f + c].astype(np.int64) * xt[xx, :c][None, :, None]).sum(1) + (1 << 21)
        tmp[:, xx] = np.clip(s >> 22, 0, 255)
    out = np.empty((oh, ow, 3), np.uint8)
    for yy in range(oh):
        f, c = yb[yy]
        s = (tmp[f:f + c].astype(np.int64) * yt[yy, :c][:, None, None]).sum(0) + (1 << 21)
        out[yy] = np.clip(s >> 22, 0, 255)
    return out


def test_library_tap_tables_reproduce_the_oracle():
    rng = np.random.default_rng(6)
    for (h, w, oh, ow) in [(60, 80, 48, 64), (75, 50, 96, 64), (31, 97, 32, 128), (240, 320, 17, 23), (9, 9, 9, 9)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        np.testing.assert_array_equal(_resize_with_tables(img, oh, ow), oimg.resize_bicubic(img, oh, ow))
    ksize, bounds, taps = ib.axis_taps(640, 512)  # shrink by 0.8: support 2.5 -> 7 taps
    assert ksize == 7 and bounds.shape == (512, 2) and taps.shape == (512, 7)
    assert (bounds[:, 0] >= 0).all() and (bounds.sum(1) <= 640).all()
    assert np.abs(taps.sum(1) - (1 << 22)).max() <= 4  # normalised rows
    k1, b1, t1 = ib.axis_taps(64, 64)  # identity
    assert (t1[np.arange(64), np.arange(64) - b1[:, 0]] == 1 << 22).all() and (t1.sum(1) == 1 << 22).all()


def test_plan_layout_and_bad_inputs():
    desc, tables, tmp_bytes, src_bytes = ib.plan([(60, 80), (75, 50), (60, 80)], [(48, 64), (48, 32), (48, 64)])
    assert desc.shape == (3, ib.DESC_FIELDS) and src_bytes == 3 * (60 * 80 + 75 * 50 + 60 * 80)
    assert tmp_bytes == 3 * (60 * 64 + 75 * 32 + 60 * 64)
    assert (desc[0, 7:11] == desc[2, 7:11]).all()  # equal (size -> size) pairs share one table
    assert (desc[:, 7:11] % 2 == 0).all() and desc[:, 7:11].max() < tables.size
    with pytest.raises(ValueError):
        ib.plan([(0, 5)], [(4, 4)])
    with pytest.raises(ValueError):
        ib.image_batch([np.zeros((4, 4), np.uint8)], [(2, 2)])
    with pytest.raises(Exception, match="Not implemented on the CPU"):
        ib.image_batch([np.zeros((4, 4, 3), np.uint8)], [(2, 2)], device='cpu')


def test_paired_collator_captions_follow_the_reference_padding():
    from grit_amd.datasets.caption.coco import PairedCollator

    class Field(object):
        use_hdf5_feat, use_gri_feat, use_reg_feat = True, True, False

    feat = lambda: {'gri_feat': torch.zeros(4, 8), 'gri_mask': torch.zeros(1, 1, 4, dtype=torch.bool)}
    batch = [(feat(), [5, 6, 7], 11), (feat(), [8], 12), (feat(), [9, 9, 9, 9, 9], 13)]
    out = PairedCollator(Field(), device='cpu', max_len=4)(batch)
    assert out['captions'].tolist() == [[2, 5, 6, 7, 3, 1, 1], [2, 8, 3, 1, 1, 1, 1], [2, 9, 9, 9, 9, 3, 1]]
    assert out['samples']['gri_feat'].shape == (3, 4, 8) and out['image_id'] == [11, 12, 13]


def test_image_collator_fails_loudly_without_a_device():
    """No CPU fallback: the deferred resize can only be carried out by the HIP kernels."""
    from grit_amd.datasets.caption.coco import DictionaryCollator
    from grit_amd.datasets.caption.transforms import collate_images, get_transform
    from grit_amd.lib import GritHipError

    class Cfg(object):
        size, resize_name, randaug = (48, 64), 'maxwh', False

    policy = get_transform(Cfg())['valid']
    item = policy(np.zeros((60, 80, 3), np.uint8))
    with pytest.raises(GritHipError, match="Not implemented on the CPU"):
        collate_images([item], device='cpu')
    with pytest.raises(GritHipError, match="Not implemented on the CPU"):
        DictionaryCollator(device='cpu')([(item, [4, 5], 7)])
    with pytest.raises(TypeError):
        collate_images([np.zeros((60, 80, 3), np.uint8)], device='cpu')  # not a Deferred record
    Cfg.randaug = True
    with pytest.raises(NotImplementedError):
        get_transform(Cfg())
