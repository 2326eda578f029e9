"""Top-level `models` package of the reference, served by grit_amd.models (same dotted paths:
models.caption, models.caption.detector, models.detection.det_module, models.ops.modules, models.common.*)."""
import importlib
import sys

_PREFIX = 'grit_amd.models'
for _sub in ('', '.common.swin_model', '.common.attention', '.common.pos_embed', '.ops.functions', '.ops.modules',
             '.ops.functions.ms_deform_attn_func', '.ops.modules.ms_deform_attn', '.detection.det_module',
             '.caption.containers', '.caption.base', '.caption.grid_net', '.caption.cap_generator',
             '.caption.transformer', '.caption.detector'):
    importlib.import_module(_PREFIX + _sub)
for _name, _mod in list(sys.modules.items()):
    if _name == _PREFIX or _name.startswith(_PREFIX + '.'):
        sys.modules['models' + _name[len(_PREFIX):]] = _mod
