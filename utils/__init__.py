"""Top-level `utils` package of the reference, served by grit_amd.utils (utils.misc, utils.cap_scheduler, utils.typing)."""
import importlib
import sys

_PREFIX = 'grit_amd.utils'
for _sub in ('', '.misc', '.cap_scheduler', '.typing'):
    importlib.import_module(_PREFIX + _sub)
for _name, _mod in list(sys.modules.items()):
    if _name == _PREFIX or _name.startswith(_PREFIX + '.'):
        sys.modules['utils' + _name[len(_PREFIX):]] = _mod
