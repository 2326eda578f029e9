"""Configuration as a plain attribute namespace.

The reference drives everything through hydra/OmegaConf (configs/caption/coco_config.yaml, train_caption.py:207);
neither is installed here, and the model code only ever does attribute access (`config.model.d_model`,
`getattr(cfg, 'aux_loss', False)`), so a nested namespace with the same keys is a drop-in.  `default_config()`
returns the values of coco_config.yaml:1-93; `load_yaml()` reads a user yaml with the same layout (${...}
interpolations resolved for the two forms the reference uses: ${oc.env:NAME} and ${a.b}).
"""
import os
import re
from types import SimpleNamespace


class Config(SimpleNamespace):
    """Namespace that also supports `'key' in cfg` and dict-style get (OmegaConf habits used by the scripts)."""

    def __contains__(self, key):
        return key in self.__dict__

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Config) else v) for k, v in self.__dict__.items()}


def to_config(obj):
    if isinstance(obj, dict):
        return Config(**{k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_config(v) for v in obj]
    return obj


_DEFAULTS = {
    'exp': dict(seed=42, name='eval', rank=0, ngpus_per_node=8, world_size=8, checkpoint='', eval=False, resume=False),
    'dataset': dict(overfit=False, ann_root='', img_root='', hdf5_path='', vocab_path='',
                    use_gri_feat=True, use_reg_feat=True,
                    transform_cfg=dict(size=[384, 640], resize_name='maxwh', randaug=False)),  # reference yaml: randaug true (host-side PIL, not provided)
    'model': dict(
        use_gri_feat=True, use_reg_feat=True, grid_feat_dim=1024, frozen_stages=2, beam_size=5, beam_len=20,
        dropout=0.2, attn_dropout=0.2, vocab_size=10201, max_len=54, pad_idx=1, bos_idx=2, eos_idx=3, d_model=512,
        n_heads=8, grid_net=dict(n_memories=1, n_layers=3), cap_generator=dict(decoder_name='parallel', n_layers=3),
        detector=dict(checkpoint='', d_model=512, dim_feedforward=1024, num_heads=8, num_layers=6, num_levels=4,
                      num_points=4, num_queries=150, num_classes=1849, dropout=0.1, activation='relu',
                      return_intermediate=True, with_box_refine=True)),
    'optimizer': dict(warmup_init_lr=1e-5, min_lr=1e-4, xe_lr=1e-4, sc_lr=5e-6, xe_backbone_lr=1e-5,
                      sc_backbone_lr=5e-6, weight_decay=0.01, beta_1=0.9, beta_2=0.99, batch_size=16, num_workers=2,
                      freezing_xe_epochs=0, freezing_sc_epochs=0, finetune_xe_epochs=10, finetune_sc_epochs=10,
                      freeze_detector=False, freeze_backbone=False),
}


def default_config(**overrides):
    """coco_config.yaml as a namespace.  Overrides use dotted keys: default_config(**{'model.dropout': 0.0})."""
    import copy
    raw = copy.deepcopy(_DEFAULTS)
    data_root = os.environ.get('DATA_ROOT', '')
    raw['dataset'].update(ann_root=os.path.join(data_root, 'annotations'), img_root=data_root,
                          hdf5_path=os.path.join(data_root, 'all_splits.h5'),
                          vocab_path=os.path.join(data_root, 'annotations', 'vocab.json'))
    for dotted, value in overrides.items():
        node = raw
        *path, leaf = dotted.split('.')
        for p in path:
            node = node.setdefault(p, {})
        node[leaf] = value
    return to_config(raw)


def load_yaml(path):
    import yaml
    with open(path) as f:
        raw = yaml.safe_load(f)

    def lookup(dotted):
        node = raw
        for p in dotted.split('.'):
            node = node[p]
        return node

    def resolve(v):
        if isinstance(v, dict):
            return {k: resolve(x) for k, x in v.items()}
        if isinstance(v, list):
            return [resolve(x) for x in v]
        if isinstance(v, str):
            whole = re.fullmatch(r"\$\{([\w.]+)\}", v)
            if whole:
                return resolve(lookup(whole.group(1)))
            return re.sub(r"\$\{oc\.env:(\w+)\}", lambda m: os.environ.get(m.group(1), ''), v)
        return v

    raw.pop('hydra', None)
    return to_config(resolve(raw))
