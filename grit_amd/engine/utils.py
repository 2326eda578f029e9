"""`engine.utils` names used on the captioning path (reference engine/utils.py:250-295)."""
from grit_amd.utils.misc import (NestedTensor, get_rank, get_world_size, inverse_sigmoid,  # noqa: F401
                                 is_dist_avail_and_initialized, is_main_process, nested_tensor_from_tensor_list)
