"""Cached-feature path (SURVEY next-row N3): the feature store with the reference's dataset names / shapes / dtypes
(tools/extract_features.py:76-86), ImageField.preprocess in cached mode (datasets/caption/field.py:47-63), the collator's
cached branch (datasets/caption/coco.py:39-47) -- CPU -- and the extraction itself on the HIP path (GPU)."""
import numpy as np
import pytest
import torch

from grit_amd.datasets.caption.coco import PairedCollator
from grit_amd.datasets.caption.feature_store import FeatureStore
from grit_amd.datasets.caption.field import ImageField


def test_store_layout_roundtrip_and_field_preprocess(tmp_path):
    ids = [391895, 522418, 184613, 318219]
    store = FeatureStore.create(str(tmp_path / 'feats'), ids, grid_tokens=60, grid_dim=1024, queries=150, d_model=512)
    assert {k: (v.shape, v.dtype.name) for k, v in store.arrays.items()} == {
        'gri_feat': ((4, 60, 1024), 'float32'), 'gri_mask': ((4, 1, 1, 60), 'bool'),
        'reg_feat': ((4, 150, 512), 'float32'), 'reg_mask': ((4, 1, 1, 150), 'bool')}
    rng = np.random.default_rng(0)
    want = {k: (rng.random(v.shape) > 0.5) if v.dtype == bool else rng.standard_normal(v.shape).astype(np.float32)
            for k, v in store.arrays.items()}
    for r in (0, 1):  # two "ranks" write disjoint rows of the same files
        w = FeatureStore.open(str(tmp_path / 'feats'), mode='r+')
        for k in want:
            w[k][r::2] = want[k][r::2]
        w.flush()
    field = ImageField(hdf5_path=str(tmp_path / 'feats'), use_gri_feat=True, use_reg_feat=True, use_hdf5_feat=True)
    item = field.preprocess('val2014/COCO_val2014_000000184613.jpg')  # id parsed from the COCO file name
    for k in want:
        assert item[k].dtype == (torch.bool if want[k].dtype == bool else torch.float32)
        np.testing.assert_array_equal(item[k].numpy(), want[k][2])
    assert set(field.preprocess('x', image_id=522418)) == set(want)
    grid_only = ImageField(hdf5_path=str(tmp_path / 'feats'), use_gri_feat=True, use_hdf5_feat=True)
    assert set(grid_only.preprocess('x', image_id=391895)) == {'gri_feat', 'gri_mask'}
    batch = [(field.preprocess('x', image_id=i), [4, 5, 6], i) for i in ids[:3]]
    out = PairedCollator(field, device='cpu')(batch)
    assert out['samples']['reg_feat'].shape == (3, 150, 512) and out['samples']['gri_mask'].shape == (3, 1, 1, 60)
    assert out['captions'].tolist() == [[2, 4, 5, 6, 3]] * 3
    with pytest.raises(KeyError):
        field.preprocess('x', image_id=1)


@pytest.mark.gpu
def test_extraction_writes_what_the_detector_computes(tmp_path):
    """extract_vis_features on the HIP path: rows land at their image index for every rank's share, equal the detector's
    own outputs for the same padded batch, and feed the cached-feature training mode."""
    from extract_features import canvas_size, extract_vis_features
    from grit_amd.datasets.caption.transforms import collate_images, get_transform
    from grit_amd.engine.caption_engine import build_optimizers, train_xe_step
    from tests.helpers import build_model
    model, cfg = build_model(2)
    model = model.to('cuda').eval()
    cfg.dataset.transform_cfg.size = [128, 192]
    cfg.dataset.transform_cfg.randaug = False
    assert canvas_size(cfg.dataset.transform_cfg) == (128, 192)
    rng = np.random.default_rng(4)
    images = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in [(240, 320), (300, 200), (128, 192), (90, 400), (333, 500)]]
    ids = [11, 22, 33, 44, 55]
    out = str(tmp_path / 'feats')
    for rank in (0, 1):  # two ranks' shares, one after the other
        store = extract_vis_features(model.detector, images, ids, cfg, out, 'cuda', rank=rank, world_size=2, batch_size=2)
    assert store['gri_feat'].shape == (5, 2 * 3, 1024) and store['reg_feat'].shape == (5, 150, 512)
    assert store['image_ids'].tolist() == ids
    policy = get_transform(cfg.dataset.transform_cfg)['valid']
    with torch.inference_mode():
        for rows in ([0, 2], [4], [1, 3]):  # the batches the two ranks formed
            want = model.detector(collate_images([policy(images[i]) for i in rows], 'cuda', pad_to=(128, 192)))
            for k in ('gri_feat', 'gri_mask', 'reg_feat', 'reg_mask'):
                np.testing.assert_array_equal(store[k][rows], want[k].cpu().numpy())
    assert store['gri_mask'][1].any() and not store['gri_mask'][2].any()  # 300x200 leaves the right of the canvas padded
    # decoder-only training consumes the rows
    field = ImageField(hdf5_path=out, use_gri_feat=True, use_reg_feat=True, use_hdf5_feat=True)
    batch = PairedCollator(field, device='cuda')([(field.preprocess('x', image_id=i), [4, 5, 6, 7], i) for i in ids[:4]])
    model.cached_features = True
    model.train()
    loss = train_xe_step(model, batch, build_optimizers(model, cfg, mode='xe'), torch.nn.NLLLoss(ignore_index=1))
    assert torch.isfinite(loss)
