from .transformer import *  # noqa: F401,F403
from .grid_net import *  # noqa: F401,F403
from .cap_generator import *  # noqa: F401,F403
