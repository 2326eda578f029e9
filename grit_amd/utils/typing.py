"""Type aliases used by the stateful containers (reference: utils/typing.py)."""
from typing import Sequence, Union

import torch

TensorOrSequence = Union[Sequence[torch.Tensor], torch.Tensor]
TensorOrNone = Union[torch.Tensor, None]
