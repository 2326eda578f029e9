"""Device image pipeline (grit_image_batch_fwd through grit_amd.ops.image_batch) against the C oracle, fixture G11 (made
from the reference's resize classes) and size-independent properties.  Integer stage and float stage: bit-exact."""
import os

import numpy as np
import pytest
import torch

from grit_amd.datasets.caption.coco import DictionaryCollator
from grit_amd.datasets.caption.transforms import MaxWHResize, MinMaxResize, collate_images
from grit_amd.ops.image_batch import MEAN, STD, image_batch
from oracle import image as oimg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["maxwh", "minmax"])
def test_fixture_g11_bit_exact(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "image_g11.npz"))
    policy = MaxWHResize((48, 64)) if name == 'maxwh' else MinMaxResize((64, 128))
    nt = collate_images([policy(g['%s_in%d' % (name, i)]) for i in range(5)], device='cuda')
    assert nt.tensors.dtype == torch.float32 and nt.mask.dtype == torch.bool and nt.any_padding
    np.testing.assert_array_equal(nt.tensors.cpu().numpy(), g[name + '_tensors'])
    np.testing.assert_array_equal(nt.mask.cpu().numpy(), g[name + '_mask'])


@pytest.mark.parametrize("on_device", [False, True])
def test_coco_like_ragged_batch_equals_oracle(on_device):
    rng = np.random.default_rng(21)
    shapes = [(480, 640), (427, 640), (640, 480), (333, 500), (375, 500), (500, 375), (96, 128), (1200, 1600)]
    images = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in shapes]
    policy = MaxWHResize((384, 640))
    sizes = [policy.output_size(*s) for s in shapes]
    want_t, want_m = oimg.image_batch(images, sizes)
    feed = [torch.from_numpy(im).cuda() for im in images] if on_device else images
    got_t, got_m = image_batch(feed, sizes, device='cuda')
    assert got_t.shape == (8, 3, 384, 576)
    np.testing.assert_array_equal(got_m.cpu().numpy(), want_m)
    np.testing.assert_array_equal(got_t.cpu().numpy(), want_t)


def test_upscale_downscale_extremes_equal_oracle():
    rng = np.random.default_rng(22)
    cases = [((5, 7), (20, 3)), ((1, 9), (4, 4)), ((33, 47), (200, 31)), ((900, 1300), (37, 53)), ((64, 64), (64, 64))]
    images = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s, _ in cases]
    # saturated edges: bicubic overshoot must clamp like Pillow
    images[2][:, ::2] = 255
    images[2][:, 1::2] = 0
    sizes = [t for _, t in cases]
    want_t, want_m = oimg.image_batch(images, sizes)
    got_t, got_m = image_batch(images, sizes, device='cuda')
    np.testing.assert_array_equal(got_t.cpu().numpy(), want_t)
    np.testing.assert_array_equal(got_m.cpu().numpy(), want_m)


@pytest.mark.parametrize("shape,target,ksize", [((60, 80), (48, 64), 7), ((100, 150), (60, 90), 9), ((100, 150), (40, 60), 11),
                                                 ((90, 130), (30, 45), 13), ((64, 64), (3, 2), 0), ((3, 2), (64, 64), 5),
                                                 ((2, 1), (1, 1), 0)])
def test_each_tap_count_variant_equals_oracle(shape, target, ksize):
    """One batch per horizontal-pass variant (<= 7, 9, 13 taps, generic) plus images narrower than a dword."""
    from grit_amd.ops.image_batch import axis_taps
    if ksize:
        assert axis_taps(shape[1], target[1])[0] == ksize
    rng = np.random.default_rng(sum(shape) + sum(target))
    images = [rng.integers(0, 256, shape + (3,), dtype=np.uint8) for _ in range(3)]
    want_t, want_m = oimg.image_batch(images, [target] * 3)
    got_t, got_m = image_batch(images, [target] * 3, device='cuda')
    np.testing.assert_array_equal(got_t.cpu().numpy(), want_t)
    np.testing.assert_array_equal(got_m.cpu().numpy(), want_m)


def test_pad_to_canvas():
    rng = np.random.default_rng(31)
    images = [rng.integers(0, 256, (50, 70, 3), dtype=np.uint8), rng.integers(0, 256, (80, 40, 3), dtype=np.uint8)]
    sizes = [(40, 56), (64, 32)]
    want_t, want_m = oimg.image_batch(images, sizes)
    got_t, got_m = image_batch(images, sizes, device='cuda', pad_to=(96, 101))  # odd width: scalar-store path
    assert got_t.shape == (2, 3, 96, 101) and got_m.shape == (2, 96, 101)
    np.testing.assert_array_equal(got_t[:, :, :64, :56].cpu().numpy(), want_t)
    np.testing.assert_array_equal(got_m[:, :64, :56].cpu().numpy(), want_m)
    assert got_m[:, 64:].all() and got_m[:, :, 56:].all() and (got_t[:, :, 64:] == 0).all() and (got_t[:, :, :, 56:] == 0).all()
    with pytest.raises(ValueError):
        image_batch(images, sizes, device='cuda', pad_to=(32, 32))


def test_full_size_properties_without_oracle():
    """BASELINE-size batch (32 x 640 x 640): resizing to the source size is the identity, so the output is exactly the
    ToTensor + Normalize table applied to the pixels; a constant image stays constant under any resize."""
    g = torch.Generator().manual_seed(3)
    imgs = [torch.randint(0, 256, (640, 640, 3), dtype=torch.uint8, generator=g) for _ in range(32)]
    t, m = image_batch(imgs, [(640, 640)] * 32, device='cuda')
    assert not m.any()
    mean = torch.tensor(MEAN)[:, None, None]
    std = torch.tensor(STD)[:, None, None]
    for i in (0, 17, 31):
        want = imgs[i].permute(2, 0, 1).to(torch.float32).div(255).sub(mean).div(std)
        assert torch.equal(t[i].cpu(), want)
    flat = [torch.full((480, 640, 3), v, dtype=torch.uint8) for v in (0, 37, 255)]
    t, m = image_batch(flat, [(384, 512), (200, 300), (640, 853)], device='cuda')
    for i, (v, (oh, ow)) in enumerate(zip((0, 37, 255), [(384, 512), (200, 300), (640, 853)])):
        want = ((torch.tensor(float(v)) / 255) - torch.tensor(MEAN)) / torch.tensor(STD)
        assert torch.equal(t[i, :, :oh, :ow].cpu(), want[:, None, None].expand(3, oh, ow))
        assert not m[i, :oh, :ow].any() and m[i, oh:].all() and m[i, :, ow:].all()
        assert (t[i, :, oh:] == 0).all() and (t[i, :, :, ow:] == 0).all()


def test_collator_feeds_the_detector_contract():
    rng = np.random.default_rng(23)
    policy = MinMaxResize((384, 640))
    batch = [(policy(rng.integers(0, 256, s + (3,), dtype=np.uint8)), [4, 5], i) for i, s in enumerate([(480, 640), (640, 427)])]
    out = DictionaryCollator(device='cuda')(batch)
    nt = out['samples']
    assert nt.tensors.is_cuda and nt.tensors.shape == (2, 3, 576, 512) and nt.mask.shape == (2, 576, 512)
    assert out['image_id'] == [0, 1] and nt.any_padding
    assert not nt.mask[0, :384, :512].any() and nt.mask[0, 384:].all() and not nt.mask[1, :576, :384].any()
