"""Cached detector features for the decoder-only ("freezing") training mode -- SURVEY next-row N3.

The reference keeps them in one HDF5 file (tools/extract_features.py:48-155) with the datasets

    image_ids [N] int64,  gri_feat [N, fh*fw, C] f32,  gri_mask [N, 1, 1, fh*fw] bool,
    reg_feat [N, Q, D] f32,  reg_mask [N, 1, 1, Q] bool

and reads one row per image in ImageField.preprocess (datasets/caption/field.py:47-63).  h5py is not available in this
environment, so the container here is a directory holding one `.npy` per dataset -- same names, shapes and dtypes,
opened memory-mapped; every rank writes its own rows of the shared files, which removes the reference's per-rank
temporary files and the rank-0 merge pass.  `FeatureStore.open(path)[name][idx]` is what `h5py.File(path)[name][idx]` is
in the reference."""
import json
import os

import numpy as np

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

    @classmethod
    def create(cls, path, image_ids, grid_tokens, grid_dim, queries=None, d_model=None):
        """Allocate the files (rank 0, before the barrier)."""
        os.makedirs(path, exist_ok=True)
        image_ids = np.asarray(image_ids, np.int64)
        np.save(os.path.join(path, 'image_ids.npy'), image_ids)
        spec = cls.layout(len(image_ids), grid_tokens, grid_dim, queries, d_model)
        for name, (shape, dtype) in spec.items():
            np.lib.format.open_memmap(os.path.join(path, name + '.npy'), mode='w+', dtype=dtype, shape=shape).flush()
        with open(os.path.join(path, 'layout.json'), 'w') as f:
            json.dump({k: [list(v[0]), v[1]] for k, v in spec.items()}, f)
        return cls.open(path, mode='r+')

    @classmethod
    def open(cls, path, mode='r'):
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
