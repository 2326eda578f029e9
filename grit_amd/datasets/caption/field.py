"""ImageField of the captioning datasets (reference datasets/caption/field.py:23-66): one image -> what the collator
stacks.  Two modes, as in the reference: cached detector features (a row of the feature store per image id) or the
decoded image behind the resize policy (a `Deferred`; resampling happens per batch on the device).  TextField / Vocab
(spacy tokeniser, vocabulary building) stay out of scope; decoding token ids to strings is `inference_caption.decode`."""
import numpy as np
import torch

from .feature_store import FeatureStore


class ImageField(object):

    def __init__(self, hdf5_path=None, transform=None, use_reg_feat=False, use_gri_feat=False, use_hdf5_feat=False, **kwargs):
        self.hdf5_path = hdf5_path
        self.use_hdf5_feat = use_hdf5_feat
        self.use_reg_feat = use_reg_feat
        self.use_gri_feat = use_gri_feat
        self.transform = transform
        self._store = None

    def init_hdf5_feat(self):
        self.use_hdf5_feat = True
        self._store = FeatureStore.open(self.hdf5_path)
        self.image_ids = self._store['image_ids']
        self.img_id2idx = {int(img_id): idx for idx, img_id in enumerate(self.image_ids)}

    def preprocess(self, path, image_id=None):
        if self.use_hdf5_feat:
            if self._store is None:
                self.init_hdf5_feat()
            if image_id is None:  # COCO file names end in _<12-digit id>.jpg (field.py:50-51)
                image_id = int(path.split('_')[-1].split('.')[0])
            row = self.img_id2idx[int(image_id)]
            out = {}
            for feat, wanted in (('gri', self.use_gri_feat), ('reg', self.use_reg_feat)):
                if wanted:
                    for key in (feat + '_feat', feat + '_mask'):
                        out[key] = torch.from_numpy(np.array(self._store[key][row]))
            return out
        from PIL import Image  # decoding is host work (out of scope); everything after it runs on the device
        img = np.asarray(Image.open(path).convert('RGB'))
        return self.transform(img) if self.transform is not None else img
