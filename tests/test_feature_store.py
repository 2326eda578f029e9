"""Cached-feature path (SURVEY next-row N3): the feature store with the reference's dataset names / shapes / dtypes
(tools/extract_features.py:76-86), ImageField.preprocess in cached mode (datasets/caption/field.py:47-63), the collator's
cached branch (datasets/caption/coco.py:39-47) -- CPU -- and the extraction itself on the HIP path (GPU)."""
import os
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


# ---------------------------------------------------------------------------------------------------------------------
# HDF5 container (the reference's format, tools/extract_features.py:66-155) without h5py
def _expected_ref():
    import numpy as np
    i = np.arange(30)
    return {"image_ids": np.array([139, 285, 632, 724, 776], np.int64),
            "gri_feat": ((np.arange(5 * 6 * 16) % 97) * 0.25 - 3.0).astype(np.float32).reshape(5, 6, 16),
            "gri_mask": ((i % 7 == 3) | (i % 5 == 0)).reshape(5, 1, 1, 6),
            "reg_feat": ((np.arange(160) ** 2 % 31) - 15.5).astype(np.float32).reshape(5, 4, 8),
            "reg_mask": np.zeros((5, 1, 1, 4), bool)}


def test_hdf5_reader_on_a_file_written_by_the_hdf5_library():
    """tests/golden/features_ref.h5 was written by libhdf5 with the calls h5py's create_dataset makes (make_golden.py h5)."""
    import numpy as np
    from grit_amd.datasets.caption.feature_store import FeatureStore
    from grit_amd.datasets.caption.hdf5_min import H5File
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "features_ref.h5")
    exp = _expected_ref()
    h5 = H5File(path)
    assert sorted(h5.keys()) == sorted(exp)
    for k, v in exp.items():
        got = h5[k]
        assert got.dtype == v.dtype and got.shape == v.shape, k
        np.testing.assert_array_equal(np.asarray(got), v)
    store = FeatureStore.open(path)  # the interface ImageField uses
    np.testing.assert_array_equal(store["image_ids"], exp["image_ids"])
    np.testing.assert_array_equal(np.asarray(store["gri_feat"][3]), exp["gri_feat"][3])
    assert store["gri_mask"][2].dtype == np.bool_


def test_hdf5_writer_round_trip_and_layout(tmp_path):
    """A store created as .h5: rows written through the memory maps come back through a fresh reader; the file starts with the
    same superblock / group structure bytes as the library's (signature, version 0, 8-byte offsets, K values)."""
    import numpy as np
    from grit_amd.datasets.caption.feature_store import FeatureStore
    from grit_amd.datasets.caption.field import ImageField
    path = str(tmp_path / "feats.h5")
    exp = _expected_ref()
    store = FeatureStore.create(path, exp["image_ids"], 6, 16, queries=4, d_model=8)
    for k in ("gri_feat", "gri_mask", "reg_feat", "reg_mask"):
        for row in range(5):  # row-wise writes, as extract_features does per rank
            store[k][row] = exp[k][row]
    store.flush()
    again = FeatureStore.open(path)
    for k, v in exp.items():
        np.testing.assert_array_equal(np.asarray(again[k]), v)
    ref = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "features_ref.h5"), "rb").read(24)
    assert open(path, "rb").read(24) == ref
    field = ImageField(hdf5_path=path, use_gri_feat=True, use_reg_feat=True, use_hdf5_feat=True)
    out = field.preprocess("COCO_val2014_000000000632.jpg")
    np.testing.assert_array_equal(out["gri_feat"].numpy(), exp["gri_feat"][2])
    assert out["gri_mask"].dtype == torch.bool and out["reg_feat"].shape == (4, 8)


def test_hdf5_rejects_what_it_does_not_implement(tmp_path):
    from grit_amd.datasets.caption.hdf5_min import H5File, H5FormatError, create
    import numpy as np
    import pytest
    bad = tmp_path / "x.h5"
    bad.write_bytes(b"not an hdf5 file at all" * 8)
    with pytest.raises(H5FormatError):
        H5File(str(bad))
    with pytest.raises(H5FormatError):
        create(str(tmp_path / "y.h5"), {"n%d" % i: ((2,), np.float32) for i in range(9)})  # more than one symbol-table node
    with pytest.raises(H5FormatError):
        create(str(tmp_path / "z.h5"), {"c": ((2,), np.complex64)})
