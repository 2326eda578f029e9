from .caption import Transformer  # noqa: F401
from .caption.base import BaseCaptioner  # noqa: F401
