"""Type aliases used by the stateful containers (reference: utils/typing.py)."""
from typing import Optional, Sequence, Union

from torch import Tensor

TensorOrNone = Optional[Tensor]
TensorOrSequence = Union[Tensor, Sequence[Tensor]]
