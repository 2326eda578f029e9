"""Cached detector features for the decoder-only ("freezing") training mode -- SURVEY next-row N3.

The reference keeps them in one HDF5 file (tools/extract_features.py:48-155) with the datasets

    image_ids [N] int64,  gri_feat [N, fh*fw, C] f32,  gri_mask [N, 1, 1, fh*fw] bool,
    reg_feat [N, Q, D] f32,  reg_mask [N, 1, 1, Q] bool

and reads one row per image in ImageField.preprocess (datasets/caption/field.py:47-63).  Two containers behind one
interface (`FeatureStore.open(path)[name][idx]` is what `h5py.File(path)[name][idx]` is in the reference):

  * an HDF5 file -- any path that is not a directory / ends in .h5 / .hdf5: read and written through
    grit_amd.datasets.caption.hdf5_min (pure numpy: h5py is not installed here), the on-disk shape h5py produces for these
    datasets, so caches made by the reference's tools/extract_features.py load as they are and files written here open in h5py;
  * a directory holding one `.npy` per dataset -- same names, shapes and dtypes.

Both are opened memory-mapped and every rank writes its own rows of the shared file(s) directly, which removes the reference's
per-rank temporary files and the rank-0 merge pass."""
import json
import os

import numpy as np

from . import hdf5_min

DATASETS = ('gri_feat', 'gri_mask', 'reg_feat', 'reg_mask')


class FeatureStore(object):

    def __init__(self, path, arrays, image_ids):
        self.path, self.arrays, self.image_ids = path, arrays, image_ids

    @staticmethod
    def layout(n, grid_tokens, grid_dim, queries=None, d_model=None):
        spec = {'gri_feat': ((n, grid_tokens, grid_dim), 'float32'), 'gri_mask': ((n, 1, 1, grid_tokens), 'bool')}
        if queries:
            spec.update({'reg_feat': ((n, queries, d_model), 'float32'), 'reg_mask': ((n, 1, 1, queries), 'bool')})
        return spec

    @staticmethod
    def is_hdf5(path):
        if os.path.isdir(path):
            return False
        return path.lower().endswith(('.h5', '.hdf5', '.hdf')) or os.path.isfile(path)

    @classmethod
    def create(cls, path, image_ids, grid_tokens, grid_dim, queries=None, d_model=None):
        """Allocate the files (rank 0, before the barrier)."""
        image_ids = np.asarray(image_ids, np.int64)
        if cls.is_hdf5(path):
            spec = cls.layout(len(image_ids), grid_tokens, grid_dim, queries, d_model)
            spec['image_ids'] = ((len(image_ids),), 'int64')
            h5 = hdf5_min.create(path, {k: (v[0], np.dtype(v[1])) for k, v in spec.items()})
            h5['image_ids'][:] = image_ids
            h5.close()
            return cls.open(path, mode='r+')
        os.makedirs(path, exist_ok=True)
        np.save(os.path.join(path, 'image_ids.npy'), image_ids)
        spec = cls.layout(len(image_ids), grid_tokens, grid_dim, queries, d_model)
        for name, (shape, dtype) in spec.items():
            np.lib.format.open_memmap(os.path.join(path, name + '.npy'), mode='w+', dtype=dtype, shape=shape).flush()
        with open(os.path.join(path, 'layout.json'), 'w') as f:
            json.dump({k: [list(v[0]), v[1]] for k, v in spec.items()}, f)
        return cls.open(path, mode='r+')

    @classmethod
    def open(cls, path, mode='r'):
        if cls.is_hdf5(path):
            h5 = hdf5_min.H5File(path, mode=mode)
            if 'gri_feat' not in h5 and 'reg_feat' not in h5:
                raise FileNotFoundError("no gri_feat / reg_feat dataset in %s" % path)
            ids = np.array(h5['image_ids'][:len(h5['image_ids'])])
            store = cls(path, {name: h5[name] for name in DATASETS if name in h5}, ids)
            store._h5 = h5
            return store
        ids = np.load(os.path.join(path, 'image_ids.npy'))
        arrays = {}
        for name in DATASETS:
            f = os.path.join(path, name + '.npy')
            if os.path.exists(f):
                arrays[name] = np.load(f, mmap_mode=mode)
        if 'gri_feat' not in arrays:
            raise FileNotFoundError("no gri_feat.npy under %s" % path)
        return cls(path, arrays, ids)

    def __getitem__(self, name):
        return self.image_ids if name == 'image_ids' else self.arrays[name]

    def __contains__(self, name):
        return name == 'image_ids' or name in self.arrays

    def flush(self):
        for a in self.arrays.values():
            if hasattr(a, 'flush'):
                a.flush()
