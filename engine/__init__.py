"""Top-level `engine` package of the reference, served by grit_amd.engine (engine.caption_engine, engine.utils)."""
import importlib
import sys

_PREFIX = 'grit_amd.engine'
for _sub in ('', '.utils', '.caption_engine'):
    importlib.import_module(_PREFIX + _sub)
for _name, _mod in list(sys.modules.items()):
    if _name == _PREFIX or _name.startswith(_PREFIX + '.'):
        sys.modules['engine' + _name[len(_PREFIX):]] = _mod
