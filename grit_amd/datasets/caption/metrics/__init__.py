from .cider import Cider
from .tokenizer import PTBTokenizer

__all__ = ['Cider', 'PTBTokenizer']
